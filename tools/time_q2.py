#!/usr/bin/env python3
"""Fused Poisson loss + gradient on Q2 / Q3 meshes by launch plan (dn_config_set("PLAN2D", "T,E,R")), steady state.
usage: time_q2.py n degree B [plan ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import DiffNet2DFEM, _lib, ops
dev = torch.device("cuda:0")
n, deg, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
m = DiffNet2DFEM(None, domain_size=n, fem_basis_deg=deg, ngp_1d=3 if deg < 3 else 4).to(dev)
shape = (B, 1, n, n)
g = torch.Generator().manual_seed(3)
u, nu, f = (torch.rand(shape, generator=g).to(dev) for _ in range(3))
nu += 0.5
bc = torch.zeros(shape, dtype=torch.uint8, device=dev)
bc[..., 0] = 1; bc[..., -1] = 1
scale = 1.0 / (B * m.geom.nelem_total)
for plan in [""] + sys.argv[4:]:
    _lib.config_set("PLAN2D", plan)
    ops._POISSON_WS_BYTES.clear()
    pl = ops.PoissonPlan(m.geom, u, nu, f, None, [(bc, 0.0)], alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=scale, want_out=True, want_sums=True, loss_scale=scale)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.04:
        for _ in range(10):
            pl.launch()
        torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100):
            pl.launch()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 10)
    print(f"n={n} Q{deg} B={B} plan={plan or 'default':10s} us per launch: median {sorted(ts)[1]:.1f}  {[round(t, 1) for t in ts]}  loss {float(pl.result[2]):.6g}", flush=True)
_lib.config_set("PLAN2D", "")
