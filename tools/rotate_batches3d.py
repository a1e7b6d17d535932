#!/usr/bin/env python3
"""3-D 256^3 launch (268 MB of arrays, like the 2-D bench batch) over 1 / 4 different meshes' worth of fields in rotation: is the 3-D
steady-state rate Infinity-Cache assisted as the 2-D one is (tools/rotate_batches.py)?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import DiffNet3DFEM, ops
dev = torch.device("cuda:0")
n = 256
m = DiffNet3DFEM(None, domain_size=n, nsd=3).to(dev)
shape = (1, 1, n, n, n)
g = torch.Generator().manual_seed(1)
bc = torch.zeros(shape, dtype=torch.uint8, device=dev)
bc[..., 0] = 1; bc[..., -1] = 1
scale = 1.0 / m.geom.nelem_total
kw = dict(alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=scale, want_out=True, want_sums=True, loss_scale=scale)
sets = []
for k in range(4):
    u, nu, f = (torch.rand(shape, generator=g).to(dev) for _ in range(3))
    nu += 0.5
    sets.append((u, nu, f))
for nb in (1, 4, 1, 4):
    plans = [ops.PoissonPlan(m.geom, *sets[k], None, [(bc, 0.0)], **kw) for k in range(nb)]
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.04:
        for i in range(12):
            plans[i % nb].launch()
        torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(200):
            plans[i % nb].launch()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 5)
    print(f"3-D 256^3: {nb} field set(s) in rotation: us per launch, back to back: median {sorted(ts)[1]:.1f}  {[round(t, 1) for t in ts]}", flush=True)
