#!/usr/bin/env python3
"""One stream, prepared launches over 1 / 2 / 4 / 8 DIFFERENT batches in rotation (own input and output arrays each): how much of the
steady-state rate of re-evaluating one batch comes from lines that survive in the 256 MB Infinity Cache from one launch to the next?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import BoxFaces, DiffNet2DFEM, PackedMask, _lib, ops
dev = torch.device("cuda:0")
m = DiffNet2DFEM(None, domain_size=512, ngp_1d=3).to(dev)
shape = (64, 1, 512, 512)
g = torch.Generator().manual_seed(1)
scale = 1.0 / (64 * m.geom.nelem_total)
kw = dict(alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=scale, want_out=True, want_sums=True, loss_scale=scale)
sets = []
for k in range(int(os.environ.get('NSETS', '8'))):
    u, nu, f = (torch.rand(shape, generator=g).to(dev) for _ in range(3))
    nu += 0.5
    sets.append((u, nu, f))
# usage: rotate_batches.py [PLAN2D override | default] [box | bits | u8]   (DN_LIB_PATH selects a variant build)
if len(sys.argv) > 1 and sys.argv[1] != "default":
    _lib.config_set("PLAN2D", sys.argv[1])
form = sys.argv[2] if len(sys.argv) > 2 else "box"
bc = torch.zeros(shape, dtype=torch.uint8, device=dev)
bc[..., 0] = 1; bc[..., -1] = 1; bc[..., 0, :] = 1; bc[..., -1, :] = 1
cond = {"box": lambda: [(BoxFaces(), 0.0)], "bits": lambda: [(PackedMask.pack(bc.clone()), 0.0)], "u8": lambda: [(bc.clone(), 0.0)],
        "f32": lambda: [(bc.float(), 0.0)]}[form]
for nb in ([int(v) for v in os.environ['NBS'].split(',')] if 'NBS' in os.environ else ((1, 2, 4, 8, 1) if len(sys.argv) == 1 else (4, 1))):
    plans = [ops.PoissonPlan(m.geom, *sets[k], None, cond(), **kw) for k in range(nb)]
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.04:
        for i in range(24):
            plans[i % nb].launch()
        torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(400):
            plans[i % nb].launch()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 2.5)
    print(f"lib={os.path.basename(os.environ.get('DN_LIB_PATH', 'default'))} plan {sys.argv[1] if len(sys.argv) > 1 else 'default'} bc={form}: {nb} batch(es) in rotation: us per launch, back to back: median {sorted(ts)[1]:.2f}  {[round(t, 2) for t in ts]}", flush=True)
