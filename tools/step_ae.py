#!/usr/bin/env python3
"""One parametric training step with the AE(1,1,n_downsample=2) generator of IBN_2D.py:186 + fused FEM energy loss."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffnet_amd import DiffNet2DFEM
from diffnet_amd.networks.autoencoders import AE

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = AE(1, 1, n_downsample=2).to(dev)
fem = DiffNet2DFEM(net, domain_size=a.size, ngp_1d=3).to(dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
B, n = a.batch, a.size
nu = torch.rand(B, 1, n, n, device=dev) + 0.5
bc = torch.zeros(B, 1, n, n, device=dev, dtype=torch.uint8)
bc[..., 0] = 1; bc[..., -1] = 1
f = torch.rand(B, 1, n, n, device=dev)


def step():
    opt.zero_grad(set_to_none=True)
    u = net(nu)
    loss = fem.energy_loss(u, nu, f, dirichlet=[(bc, 0.0)], c=0.5)
    loss.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print(f"AE(1,1,nd=2) {n}x{n} batch {B}: {dt * 1e3:.2f} ms per training step, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
