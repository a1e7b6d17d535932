#!/usr/bin/env python3
"""3-D Q1 launch at 256^3 (and 128^3) with and without the coefficient / forcing fields: how does the time follow the instruction count?
(nu + f: ~255 VALU instructions per element layer, nu only: ~207, neither: ~180)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import DiffNet3DFEM, ops
dev = torch.device("cuda:0")
for n in (256, 128):
    m = DiffNet3DFEM(None, domain_size=n, nsd=3).to(dev)
    shape = (1, 1, n, n, n)
    g = torch.Generator().manual_seed(1)
    sets = [[torch.rand(shape, generator=g).to(dev) for _ in range(3)] for _ in range(4)]
    bc = torch.zeros(shape, dtype=torch.uint8, device=dev); bc[..., 0] = 1; bc[..., -1] = 1
    scale = 1.0 / m.geom.nelem_total
    kw = dict(alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=scale, want_out=True, want_sums=True, loss_scale=scale)
    for name, sel in (("nu + f", (1, 1)), ("nu only", (1, 0)), ("f only", (0, 1)), ("neither", (0, 0))):
        plans = [ops.PoissonPlan(m.geom, s[0], s[1] if sel[0] else None, s[2] if sel[1] else None, None, [(bc, 0.0)], **kw) for s in sets]
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.05:
            for i in range(16):
                plans[i % 4].launch()
            torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(200):
                plans[i % 4].launch()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 5)
        print(f"{n}^3 {name:8s}: {sorted(ts)[1]:7.2f} us per launch (4 field sets in rotation)", flush=True)
