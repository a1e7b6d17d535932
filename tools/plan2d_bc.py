#!/usr/bin/env python3
"""Launch-plan sweep (dn_config_set("PLAN2D", "T,E,R")) of the 2-D headline launch for the box / u8 Dirichlet forms."""
import os
import sys
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
plans = ["", "128,4,8", "128,4,12", "128,4,16", "128,4,20", "128,4,24", "128,4,32", "128,4,64", "64,4,16", "256,4,16", "128,2,16"]
for rnd in range(2):
    for plan in plans:
        _lib.config_set("PLAN2D", plan)
        ops._POISSON_WS_BYTES.clear()
        for name, d in forms.items():
            fn = lambda: ops.poisson_apply(m.geom, u, nu, f, None, d, alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=scale)
            for _ in range(5):
                fn()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
            for a, b in evs:
                a.record(); fn(); b.record()
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
            print(f"round {rnd} plan={plan or 'default':10s} {name:4s} median {ts[25]:.1f} us  min {ts[0]:.1f}", flush=True)
_lib.config_set("PLAN2D", "")
