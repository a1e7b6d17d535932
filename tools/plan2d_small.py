#!/usr/bin/env python3
"""2-D Q1 512^2 3x3 at small batches (cfg2 B = 1, 4, 16: launches that do not fill the chip at 16-row strips): the library's plan against chained
strips (PLAN2D "T,E,R,W") and other strip heights; 4 sets of arrays in rotation, steady state."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import DiffNet2DFEM, PackedMask, _lib, ops
dev = torch.device("cuda:0")
m = DiffNet2DFEM(None, domain_size=512, ngp_1d=3).to(dev)
plans_to_try = ["", "128,4,1,4", "128,4,2,4", "128,4,3,4", "128,4,4,4", "128,4,6,4"]
for B in (1, 2, 4, 8, 16, 24):
    shape = (B, 1, 512, 512)
    g = torch.Generator().manual_seed(1)
    sets = [[torch.rand(shape, generator=g).to(dev) for _ in range(3)] for _ in range(4)]
    bc = torch.zeros(shape, dtype=torch.uint8, device=dev); bc[..., 0] = 1; bc[..., -1] = 1; bc[..., 0, :] = 1; bc[..., -1, :] = 1
    scale = 1.0 / (B * m.geom.nelem_total)
    kw = dict(alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=scale, want_out=True, want_sums=True, loss_scale=scale)
    for plan in plans_to_try:
        _lib.config_set("PLAN2D", plan)
        pls = [ops.PoissonPlan(m.geom, s[0], s[1], s[2], None, [(PackedMask.pack(bc), 0.0)], **kw) for s in sets]
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.03:
            for i in range(16):
                pls[i % 4].launch()
            torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(400):
                pls[i % 4].launch()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 2.5)
        print(f"B={B:2d} plan {plan or 'default':12s}: {sorted(ts)[1]:7.2f} us", flush=True)
    _lib.config_set("PLAN2D", "")
