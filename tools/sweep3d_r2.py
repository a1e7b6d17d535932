#!/usr/bin/env python3
"""Launch-geometry sweep of the 3-D kernel through dn_config_set("PLAN3D", "TX,TY,E,R"), one process, HIP events."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import DiffNet3DFEM, _lib, ops   # noqa: E402

dev = torch.device("cuda:0")


def run(n, B, plans, reps=40):
    m = DiffNet3DFEM(None, domain_size=n, nsd=3).to(dev)
    shape = (B, 1, n, n, n)
    g = torch.Generator().manual_seed(1)
    u, nu, f = (torch.rand(shape, generator=g).to(dev) for _ in range(3))
    nu += 0.5
    bc = torch.zeros(shape, dtype=torch.uint8, device=dev)
    bc[..., 0] = 1; bc[..., -1] = 1; bc[..., 0, :] = 1; bc[..., -1, :] = 1; bc[:, :, 0] = 1; bc[:, :, -1] = 1
    fn = lambda: m.energy_loss_and_grad(u, nu, f, dirichlet=[(bc, 0.0)], c=1.0)
    for plan in plans:
        _lib.config_set("PLAN3D", plan)
        ops._POISSON_WS_BYTES.clear()          # the workspace size depends on the launch plan
        for _ in range(4):
            fn()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in evs:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
        print(f"n={n} B={B} plan={plan or 'default'} median_us={ts[len(ts)//2]:.1f} min_us={ts[0]:.1f}", flush=True)
    _lib.config_set("PLAN3D", "")


if __name__ == "__main__":
    n, B = int(sys.argv[1]), int(sys.argv[2])
    run(n, B, [""] + sys.argv[3:])
