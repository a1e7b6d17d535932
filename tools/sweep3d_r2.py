#!/usr/bin/env python3
"""Launch-geometry sweep of the 3-D kernel through dn_config_set("PLAN3D", "TX,TY,E,R"), in steady state: 40 ms of load first
(tools/ramp2d.py), then 100 prepared launches back to back between ONE pair of events, plans interleaved over 3 rounds.
usage: sweep3d_r2.py n B [plan ...]"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import DiffNet3DFEM, _lib, ops   # noqa: E402

dev = torch.device("cuda:0")


def run(n, B, plans, reps=100):
    m = DiffNet3DFEM(None, domain_size=n, nsd=3).to(dev)
    shape = (B, 1, n, n, n)
    g = torch.Generator().manual_seed(1)
    u, nu, f = (torch.rand(shape, generator=g).to(dev) for _ in range(3))
    nu += 0.5
    bc = torch.zeros(shape, dtype=torch.uint8, device=dev)
    bc[..., 0] = 1; bc[..., -1] = 1; bc[..., 0, :] = 1; bc[..., -1, :] = 1; bc[:, :, 0] = 1; bc[:, :, -1] = 1
    scale = 1.0 / (B * m.geom.nelem_total)
    prepared, ref = {}, None
    for plan in plans:
        _lib.config_set("PLAN3D", plan)            # read by dn_poisson_apply at every launch; the workspace size depends on it
        ops._POISSON_WS_BYTES.clear()
        pl = ops.PoissonPlan(m.geom, u, nu, f, None, [(bc, 0.0)], alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=scale, want_out=True,
                             want_sums=True, loss_scale=scale)
        grad, _, loss = pl.launch()
        if ref is None:
            ref = (grad.clone(), float(loss))
        err = float((grad - ref[0]).abs().max() / ref[0].abs().max())
        assert err < 1e-5 and abs(float(loss) - ref[1]) < 1e-5 * abs(ref[1]), (plan, err)
        prepared[plan] = pl
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.04:
        for plan, pl in prepared.items():
            _lib.config_set("PLAN3D", plan)
            pl.launch()
        torch.cuda.synchronize()
    res = {k: [] for k in prepared}
    for rnd in range(3):
        for plan, pl in prepared.items():
            _lib.config_set("PLAN3D", plan)
            for _ in range(10):
                pl.launch()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                pl.launch()
            b.record()
            torch.cuda.synchronize()
            res[plan].append(a.elapsed_time(b) * 1e3 / reps)
    _lib.config_set("PLAN3D", "")
    for plan, v in res.items():
        print(f"n={n} B={B} plan={plan or 'default':12s} us per launch, back to back: median {sorted(v)[1]:.1f}  rounds {[round(x, 1) for x in v]}", flush=True)


if __name__ == "__main__":
    n, B = int(sys.argv[1]), int(sys.argv[2])
    run(n, B, [""] + sys.argv[3:])
