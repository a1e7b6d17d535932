#!/usr/bin/env python3
"""Independent evaluations (4 different batches in rotation) issued alternately on S streams: the next launch's ramp-up overlaps the previous
one's tail (last strips, in-kernel reduction, launch gap).  usage: two_streams.py [bits|box] [PLAN2D]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import BoxFaces, DiffNet2DFEM, PackedMask, _lib, ops
dev = torch.device("cuda:0")
m = DiffNet2DFEM(None, domain_size=512, ngp_1d=3).to(dev)
shape = (64, 1, 512, 512)
g = torch.Generator().manual_seed(1)
scale = 1.0 / (64 * m.geom.nelem_total)
kw = dict(alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=scale, want_out=True, want_sums=True, loss_scale=scale)
form = sys.argv[1] if len(sys.argv) > 1 else "bits"
if len(sys.argv) > 2:
    _lib.config_set("PLAN2D", sys.argv[2])
NB = int(os.environ.get('NB', '4'))
sets = []
for k in range(NB):
    u, nu, f = (torch.rand(shape, generator=g).to(dev) for _ in range(3))
    nu += 0.5
    sets.append((u, nu, f))
bc = torch.zeros(shape, dtype=torch.uint8, device=dev)
bc[..., 0] = 1; bc[..., -1] = 1; bc[..., 0, :] = 1; bc[..., -1, :] = 1
cond = {"box": lambda: [(BoxFaces(), 0.0)], "bits": lambda: [(PackedMask.pack(bc.clone()), 0.0)]}[form]
ref = None
for S in [int(v) for v in os.environ.get('STREAMS', '1,2,3,4,1').split(',')]:
    streams = [torch.cuda.Stream() for _ in range(S)]
    plans = []
    for k in range(NB):
        with torch.cuda.stream(streams[k % S]):
            plans.append(ops.PoissonPlan(m.geom, *sets[k], None, cond(), **kw))
    torch.cuda.synchronize()
    def burst(n):
        for i in range(n):
            with torch.cuda.stream(streams[(i % NB) % S]):
                plans[i % NB].launch()
    burst(48)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for s in streams:
            s.wait_stream(torch.cuda.current_stream())
        burst(400)
        for s in streams:
            torch.cuda.current_stream().wait_stream(s)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 2.5)
    res = [(float(p.result[2]), float(p.result[0].double().abs().sum())) for p in plans]
    if ref is None:
        ref = res
    same = all(abs(a[0] - b[0]) <= 1e-6 * abs(b[0]) and abs(a[1] - b[1]) <= 1e-6 * abs(b[1]) for a, b in zip(res, ref))
    print(f"{form} {sys.argv[2] if len(sys.argv) > 2 else 'default plan'}: {S} stream(s): us per evaluation {sorted(ts)[1]:.2f}  {[round(t, 2) for t in ts]}  results {'equal' if same else 'DIFFER'}", flush=True)
