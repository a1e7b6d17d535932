#!/usr/bin/env python3
"""Launch duration of the bench kernel as a function of how long the GPU has been under load (cold process -> steady state).
The driver's bench run is 5 + 20 launches (about 1.3 ms of GPU time), so the first milliseconds are what it sees."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import BoxFaces, DiffNet2DFEM, DiffNet3DFEM, ops
dev = torch.device("cuda:0")
# usage: ramp2d.py [nsd n ngp B]   (default: the bench workload, 2 512 3 64)
nsd, n, ngp, B = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (2, 512, 3, 64)
m = (DiffNet3DFEM if nsd == 3 else DiffNet2DFEM)(None, domain_size=n, ngp_1d=ngp, nsd=nsd).to(dev)
shape = (B, 1, *m.geom.node_shape)
g = torch.Generator().manual_seed(1)
u, nu, f = (torch.rand(shape, generator=g).to(dev) for _ in range(3))
nu += 0.5
scale = 1.0 / (B * m.geom.nelem_total)
if nsd == 2:
    cond = [(BoxFaces(), 0.0)]
else:
    bc = torch.zeros(shape, dtype=torch.uint8, device=dev)
    bc[..., 0] = 1; bc[..., -1] = 1
    cond = [(bc, 0.0)]
pl = ops.PoissonPlan(m.geom, u, nu, f, None, cond, alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=scale, want_out=True, want_sums=True, loss_scale=scale)
torch.cuda.synchronize()


def series(n, label):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    evs[0].record()
    for i in range(n):
        pl.launch()
        evs[i + 1].record()
    torch.cuda.synchronize()
    d = [evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(n)]
    marks = [0, 1, 2, 5, 10, 20, 25, 50, 100, 200, 400, 800, 1600, 3000]
    row = []
    for a, b in zip(marks, marks[1:]):
        if b <= n:
            row.append(f"[{a},{b}) {sum(d[a:b]) / (b - a):.1f}")
    print(f"{label}: us per launch by launch index: " + "  ".join(row), flush=True)


series(3000 if nsd == 2 else 1600, "cold process")
series(400, "right after     ")
for idle in (0.001, 0.01, 0.1, 1.0):
    time.sleep(idle)
    series(400, f"after {idle * 1e3:6.0f} ms idle")
# does unrelated GPU work (a plain copy) bring the launch to its steady rate?
time.sleep(1.0)
a = torch.empty(64 << 20, device=dev); b = torch.empty_like(a)
for _ in range(100):
    b.copy_(a)
series(100, "after 1 s idle + 100 copies of 256 MiB")
