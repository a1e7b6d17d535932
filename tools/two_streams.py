#!/usr/bin/env python3
"""Independent evaluations issued alternately on two streams (each with its own prepared launch and workspace): does the next launch's
streaming cover the previous launch's reduction tail?  Every loss is checked against the single-stream result."""
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
kw = dict(alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=scale, want_out=True, want_sums=True, loss_scale=scale)
SHARED = len(sys.argv) > 1 and sys.argv[1] == "shared"
for nstreams in (1, 2, 3, 1, 2):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    plans = []
    for k, s in enumerate(streams):
        # every stream evaluates ITS OWN batch (separate input arrays): concurrent launches reading the same arrays would share cache lines
        uk, nuk, fk = (u, nu, f) if (k == 0 or SHARED) else (u.clone(), nu.clone(), f.clone())
        with torch.cuda.stream(s):
            plans.append(ops.PoissonPlan(m.geom, uk, nuk, fk, None, [(BoxFaces(), 0.0)], **kw))
    torch.cuda.synchronize()

    def burst(n):
        for i in range(n):
            with torch.cuda.stream(streams[i % nstreams]):
                plans[i % nstreams].launch()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.04:
        burst(20)
        torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        burst(600)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 600 * 1e6)
    ref = plans[0].result
    ok = all(torch.equal(p.result[2], ref[2]) and torch.equal(p.result[0], ref[0]) for p in plans)
    print(f"{nstreams} stream(s): us per launch (wall, 600 launches): median {sorted(ts)[1]:.2f}  {[round(t, 2) for t in ts]}  results equal: {ok}  inputs {'shared' if SHARED else 'separate per stream'}", flush=True)
