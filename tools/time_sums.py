#!/usr/bin/env python3
"""What the in-kernel final reduction costs the 2-D bench launch: the same launch with and without the sums, steady state (40 ms of load,
then 200 prepared launches back to back between one pair of events, interleaved over 3 rounds)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import BoxFaces, DiffNet2DFEM, ops
dev = torch.device("cuda:0")
m = DiffNet2DFEM(None, domain_size=512, ngp_1d=3).to(dev)
shape = (64, 1, 512, 512)
g = torch.Generator().manual_seed(1)
u, nu, f = (torch.rand(shape, generator=g).to(dev) for _ in range(3))
nu += 0.5
scale = 1.0 / (64 * m.geom.nelem_total)
kw = dict(alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=scale, want_out=True)
plans = {"loss + gradient (sums in the launch)": ops.PoissonPlan(m.geom, u, nu, f, None, [(BoxFaces(), 0.0)], want_sums=True, loss_scale=scale, **kw),
         "gradient only (no sums)": ops.PoissonPlan(m.geom, u, nu, f, None, [(BoxFaces(), 0.0)], want_sums=False, **kw)}
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.04:
    for pl in plans.values():
        pl.launch()
    torch.cuda.synchronize()
res = {k: [] for k in plans}
for rnd in range(3):
    for k, pl in plans.items():
        for _ in range(20):
            pl.launch()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200):
            pl.launch()
        b.record()
        torch.cuda.synchronize()
        res[k].append(a.elapsed_time(b) * 5)
for k, v in res.items():
    print(f"{k:40s} us per launch, back to back: median {sorted(v)[1]:.2f}  rounds {[round(x, 2) for x in v]}", flush=True)
