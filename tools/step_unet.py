#!/usr/bin/env python3
"""One parametric training step of BASELINE configs[1] (UNet 2->1 on 512 x 512, fused FEM energy loss, backward), timed
end to end; `--profile` runs few steps for rocprofv3 --kernel-trace --stats."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffnet_amd import DiffNet2DFEM
from diffnet_amd.networks.unets import UNet

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--eval", action="store_true", help="eval mode (no dropout)")
ap.add_argument("--benchmark", action="store_true", help="torch.backends.cudnn.benchmark = True (MIOpen exhaustive find)")
ap.add_argument("--channels-last", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
if a.benchmark:
    torch.backends.cudnn.benchmark = True
net = UNet(2, 1).to(dev)
if a.channels_last:
    net = net.to(memory_format=torch.channels_last)
if a.eval:
    net.eval()
fem = DiffNet2DFEM(net, domain_size=a.size, ngp_1d=3).to(dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
B, n = a.batch, a.size
nu = torch.rand(B, 1, n, n, device=dev) + 0.5
bc = torch.zeros(B, 1, n, n, device=dev, dtype=torch.uint8)
bc[..., 0] = 1; bc[..., -1] = 1
f = torch.rand(B, 1, n, n, device=dev)
x = torch.cat([nu, bc.float()], 1)


def step():
    opt.zero_grad(set_to_none=True)
    u = net(x)
    loss = fem.energy_loss(u, nu, f, dirichlet=[(bc, 0.0)], c=0.5)
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print(f"UNet(2->1) {n}x{n} batch {B}: {dt * 1e3:.2f} ms per training step ({B / dt:.0f} samples/s), peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
