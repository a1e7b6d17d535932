#!/usr/bin/env python3
"""Launch-plan sweep (dn_config_set("PLAN2D", "T,E,R")) of the 2-D headline launch for the box / u8 Dirichlet forms, in steady state:
40 ms of load first (tools/ramp2d.py), then 200 prepared launches back to back between ONE pair of events, plans interleaved over rounds.
usage: plan2d_bc.py [plan ...]   (DN_LIB_PATH selects a variant build)"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import BoxFaces, DiffNet2DFEM, _lib, ops   # noqa: E402

dev = torch.device("cuda:0")
n, B = 512, 64
m = DiffNet2DFEM(None, domain_size=n, ngp_1d=3).to(dev)
shape = (B, 1, n, n)
g = torch.Generator().manual_seed(1)
u, nu, f = (torch.rand(shape, generator=g).to(dev) for _ in range(3))
nu += 0.5
bc = torch.zeros(shape, dtype=torch.uint8, device=dev)
bc[..., 0] = 1; bc[..., -1] = 1; bc[..., 0, :] = 1; bc[..., -1, :] = 1
forms = {"box": [(BoxFaces(), 0.0)], "u8": [(bc, 0.0)]}
scale = 1.0 / (B * m.geom.nelem_total)
plans = sys.argv[1:] or ["", "128,4,8", "128,4,12", "128,4,16", "128,4,24", "128,4,32", "128,4,64", "64,4,16", "256,4,16", "128,2,16", "256,2,32", "256,2,16"]
prepared = {}
ref = {}
for plan in plans:
    _lib.config_set("PLAN2D", "" if plan == "default" else plan)
    ops._POISSON_WS_BYTES.clear()
    for name, d in forms.items():
        pl = ops.PoissonPlan(m.geom, u, nu, f, None, d, alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=scale, want_out=True, want_sums=True, loss_scale=scale)
        grad, sums, loss = pl.launch()
        if name not in ref:
            ref[name] = (grad.clone(), float(loss))
        err = float((grad - ref[name][0]).abs().max() / ref[name][0].abs().max())
        assert err < 1e-5 and abs(float(loss) - ref[name][1]) < 1e-5 * abs(ref[name][1]), (plan, name, err)
        prepared[plan, name] = pl
_lib.config_set("PLAN2D", "")
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.04:
    for (plan, _), pl in prepared.items():
        _lib.config_set("PLAN2D", "" if plan == "default" else plan)       # read by dn_poisson_apply at every launch
        pl.launch()
    torch.cuda.synchronize()
res = {k: [] for k in prepared}
for rnd in range(3):
    for k, pl in prepared.items():
        _lib.config_set("PLAN2D", "" if k[0] == "default" else k[0])
        for _ in range(20):
            pl.launch()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200):
            pl.launch()
        b.record()
        torch.cuda.synchronize()
        res[k].append(a.elapsed_time(b) * 1e3 / 200)
for (plan, name), v in res.items():
    print(f"plan={plan or 'default':10s} {name:4s} us per launch, back to back: median {sorted(v)[1]:.2f}  rounds {[round(x, 2) for x in v]}", flush=True)
