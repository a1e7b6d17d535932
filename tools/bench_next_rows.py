#!/usr/bin/env python3
"""Device time of the SURVEY 8(f) kernels at BASELINE-scale sizes: winding-number field (512^2 nodes x 1000 boundary points x B = 16)
and the FDM stencils (512^2, B = 64), HIP events; GB/s on the algorithmic bytes, GFLOP/s for the winding field."""
import os
import sys
import math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd.ops import compute_winding_nodes   # noqa: E402
from diffnet_amd.fdm import DiffNetFDM   # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    return ts[len(ts) // 2]


B, npts, n = 16, 1000, 512
g = torch.Generator().manual_seed(0)
th = torch.sort(torch.rand((B, npts), generator=g) * 2 * math.pi, dim=1).values
pts = torch.stack([0.5 + 0.3 * torch.cos(th), 0.5 + 0.3 * torch.sin(th)], -1).to(dev).unsqueeze(1)
nrm = torch.stack([torch.cos(th), torch.sin(th)], -1).to(dev).unsqueeze(1)
area = torch.zeros(B, 1, npts, 1, device=dev)
xx, yy = torch.meshgrid(torch.linspace(0, 1, n), torch.linspace(0, 1, n), indexing="xy")
nodes = torch.stack((xx, yy), 0).to(dev)
t = timeit(lambda: compute_winding_nodes(pts, nrm, area, nodes))
pairs = B * n * n * npts
print(f"dn_winding_nodes {n}^2 nodes x {npts} points x B={B}: {t:.1f} us  {pairs / t / 1e3:.1f} G point-node pairs/s  (~{pairs * 12 / t / 1e6:.1f} TFLOP/s at 12 flop per pair; "
      f"output {B * n * n * 4 / t / 1e3:.1f} GB/s)", flush=True)
m = DiffNetFDM(None, domain_size=n).to(dev)
Bf = 64
u = torch.rand(Bf, 1, n, n, device=dev)
up = m.pad(u)
for name, fn in (("derivative_x (padded input)", lambda: m.derivative_x(up)), ("derivative_xx (padded input)", lambda: m.derivative_xx(up)),
                 ("pad + derivative_x (torch pad, then kernel)", lambda: m.derivative_x(m.pad(u))), ("dx fused pad", lambda: m.dx(u)),
                 ("dyy fused pad", lambda: m.dyy(u))):
    t = timeit(fn)
    byt = 8 * Bf * n * n
    print(f"FDM {name:45s} {n}^2 B={Bf}: {t:7.1f} us  {byt / t / 1e3:7.1f} GB/s on 8 B/node ({byt / t / 8e6:.3f} of the HBM peak)", flush=True)
ug = u.clone().requires_grad_(True)
d = m.dx(ug)
cot = torch.rand_like(d)
t = timeit(lambda: torch.autograd.grad(d, ug, cot, retain_graph=True))
print(f"FDM dx fused backward: {t:.1f} us  {8 * Bf * n * n / t / 1e3:.1f} GB/s", flush=True)
