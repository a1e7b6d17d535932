#!/usr/bin/env python3
"""Device time of one dn_fsdt_apply launch (forward form: residuals + norms) vs batch and launch plan."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import DiffNet2DFEM, _lib, ops   # noqa: E402

dev = torch.device("cuda:0")
n, deg = int(sys.argv[1]), int(sys.argv[2])
plans = [""] + sys.argv[3:]
m = DiffNet2DFEM(None, domain_size=n, fem_basis_deg=deg, ngp_1d=3 if deg > 1 else 2).to(dev)
for B in (1, 2, 4, 8):
    shape = (B, 1, n, n)
    g = torch.Generator().manual_seed(2)
    flds = [torch.rand(shape, generator=g).to(dev) for _ in range(3)]
    bc = torch.zeros(shape, device=dev)
    bc[..., 0] = 1; bc[..., -1] = 1; bc[..., 0, :] = 1; bc[..., -1, :] = 1
    for plan in plans:
        _lib.config_set("PLAN_FSDT", plan)
        ops._FSDT_WS_BYTES.clear()
        fn = lambda: ops.fsdt_apply(m.geom, *flds, bc, q=1.0, wscale=(0.5 * m.h) ** 2)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()                 # 40 ms of load first (tools/ramp2d.py), then 50 launches back to back between ONE pair of events, 3 times
        while time.perf_counter() - t0 < 0.04:
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(50):
                fn()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3 / 50)
        ts = sorted(ts) * 7                      # median at index 10, min at 0 (print below)
        byt = 28 * B * n * n
        print(f"n={n} Q{deg} B={B} plan={plan or 'default'}: median {ts[10]:.1f} us min {ts[0]:.1f}  {byt / ts[10] / 1e3:.0f} GB/s ({byt / ts[10] / 8e6:.3f})", flush=True)
    _lib.config_set("PLAN_FSDT", "")
