#!/usr/bin/env python3
"""2-D headline launch (512^2, B = 64, 3x3) with the Dirichlet condition held in each format, interleaved rounds, HIP events."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import BoxFaces, DiffNet2DFEM, PackedMask, ops   # noqa: E402

dev = torch.device("cuda:0")
n, B = (int(sys.argv[1]) if len(sys.argv) > 1 else 512), (int(sys.argv[2]) if len(sys.argv) > 2 else 64)
m = DiffNet2DFEM(None, domain_size=n, ngp_1d=3).to(dev)
shape = (B, 1, n, n)
g = torch.Generator().manual_seed(1)
u, nu, f = (torch.rand(shape, generator=g).to(dev) for _ in range(3))
nu += 0.5
bc = torch.zeros(shape, dtype=torch.uint8, device=dev)
bc[..., 0] = 1; bc[..., -1] = 1; bc[..., 0, :] = 1; bc[..., -1, :] = 1
forms = {"none": [], "u8": [(bc, 0.0)], "bits": [(PackedMask.pack(bc), 0.0)], "box": [(BoxFaces(), 0.0)], "f32": [(bc.float(), 0.0)],
         "bits+box": [(PackedMask.pack(bc), 1.0), (BoxFaces(), 0.0)], "u8+u8": [(bc, 1.0), (bc[:1].contiguous(), 0.0)]}
scale = 1.0 / (B * m.geom.nelem_total)
res = {k: [] for k in forms}
for rnd in range(4):
    for name, d in forms.items():
        fn = lambda: ops.poisson_apply(m.geom, u, nu, f, None, d, alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=scale)
        for _ in range(5):
            fn()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
        for a, b in evs:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
        res[name].append((ts[25], ts[0]))
for name, rows in res.items():
    med = sorted(r[0] for r in rows)
    print(f"n={n} B={B} {name:9s} median-of-rounds {med[len(med)//2]:.1f} us  rounds {[round(r[0],1) for r in rows]}  min {min(r[1] for r in rows):.1f}"
          f"  -> {16*B*n*n/med[len(med)//2]/1e3/8000:.3f} of HBM peak", flush=True)
