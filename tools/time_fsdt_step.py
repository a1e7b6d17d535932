#!/usr/bin/env python3
"""fsdt_loss forward + backward device time by batch (HIP events around each phase)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import DiffNet2DFEM   # noqa: E402
from diffnet_amd.elasticity import fsdt_loss   # noqa: E402

dev = torch.device("cuda:0")
n, deg = int(sys.argv[1]), int(sys.argv[2])
m = DiffNet2DFEM(None, domain_size=n, fem_basis_deg=deg, ngp_1d=3 if deg > 1 else 2).to(dev)
for B in (1, 8):
    shape = (B, 1, n, n)
    g = torch.Generator().manual_seed(2)
    fields = [torch.rand(shape, generator=g).to(dev).requires_grad_(True) for _ in range(3)]
    bc = torch.zeros(shape, device=dev)
    bc[..., 0] = 1; bc[..., -1] = 1; bc[..., 0, :] = 1; bc[..., -1, :] = 1
    for it in range(6):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        loss = sum(fsdt_loss(m, *fields, bc))
        e[1].record()
        torch.autograd.grad(loss, fields)
        e[2].record()
        torch.cuda.synchronize()
        if it >= 3:
            print(f"B={B} fwd {e[0].elapsed_time(e[1])*1e3:.0f} us  bwd {e[1].elapsed_time(e[2])*1e3:.0f} us", flush=True)
