#!/usr/bin/env python3
"""The FSDT loss + gradient pair (ops.FsdtPlan: residual launch with deferred sums, VJP launch) repeated, for rocprofv3: run_fsdt.py [B] [reps] [n]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import DiffNet2DFEM, ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1025
dev = torch.device("cuda:0")
m = DiffNet2DFEM(None, domain_size=n, fem_basis_deg=2, ngp_1d=3).to(dev)
shape = (B, 1, n, n)
g = torch.Generator().manual_seed(2)
f = [torch.rand(shape, generator=g).to(dev) for _ in range(3)]
bc = torch.zeros(shape, device=dev); bc[..., 0] = 1; bc[..., -1] = 1; bc[..., 0, :] = 1; bc[..., -1, :] = 1
plan = ops.FsdtPlan(m.geom, *f, bc, (0.0, 0.0, 0.0), q=1.0, wscale=(0.5 * m.h) ** 2)
for _ in range(reps):
    plan.launch()
torch.cuda.synchronize()
print("done")
