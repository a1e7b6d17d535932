#!/usr/bin/env python3
"""One parametric 3-D training step (GoodGenerator 1->1 on n^3, fused FEM energy loss, backward, Adam), timed end to end."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffnet_amd import DiffNet3DFEM
from diffnet_amd.networks.wgan3d import GoodGenerator

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--size", type=int, default=128)
ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = GoodGenerator(1, 1).to(dev)
fem = DiffNet3DFEM(net, domain_size=a.size, ngp_1d=2, nsd=3).to(dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
B, n = a.batch, a.size
nu = torch.rand(B, 1, n, n, n, device=dev) + 0.5
bc = torch.zeros(B, 1, n, n, n, device=dev, dtype=torch.uint8)
bc[..., 0] = 1; bc[..., -1] = 1


def step():
    opt.zero_grad(set_to_none=True)
    u = net(nu)
    loss = fem.energy_loss(u, nu, None, dirichlet=[(bc, 0.0)], c=0.5)
    loss.backward()
    opt.step()
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print(f"GoodGenerator(1->1) {n}^3 batch {B}: {dt * 1e3:.2f} ms per training step, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
