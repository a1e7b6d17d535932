#!/usr/bin/env python3
"""Wall time of the driver-sized timed region (20 prepared launches, synchronised on both sides) by the way the host waits for the GPU."""
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
pl = ops.PoissonPlan(m.geom, u, nu, f, None, [(BoxFaces(), 0.0)], alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=scale, want_out=True, want_sums=True, loss_scale=scale)
for _ in range(1000):
    pl.launch()
torch.cuda.synchronize()
K = 20


def run(kind):
    torch.cuda.synchronize()
    ev = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    for _ in range(K):
        pl.launch()
    if kind == "event.synchronize() then torch.cuda.synchronize()":
        ev.record(); ev.synchronize()
    elif kind == "stream.synchronize() then torch.cuda.synchronize()":
        torch.cuda.current_stream().synchronize()
    elif kind == "loss.item() then torch.cuda.synchronize()":
        pl.result[2].item()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e6


kinds = ["torch.cuda.synchronize()", "event.synchronize() then torch.cuda.synchronize()", "stream.synchronize() then torch.cuda.synchronize()",
         "loss.item() then torch.cuda.synchronize()"]
res = {k: [] for k in kinds}
for rnd in range(40):
    for k in kinds:
        res[k].append(run(k))
for k, v in res.items():
    v.sort()
    print(f"{k:52s} us per step over {K} steps: median {v[len(v) // 2]:.2f}  min {v[0]:.2f}  max {v[-1]:.2f}", flush=True)
