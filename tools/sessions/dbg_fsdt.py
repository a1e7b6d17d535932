import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_gpu_parity import boundary_mask, dev, module, seeded
from diffnet_amd import _lib, ops
m = module(dict(domain_size=513, fem_basis_deg=2, ngp_1d=3))
shape = (2, 1, 513, 513)
a3 = [seeded(shape, 100 + i).to(dev()) for i in range(3)]
bc = boundary_mask(shape).to(dev())
kw = dict(D11=1.0, D12=0.3, D22=1.0, D66=0.35, A44=40.0, A55=40.0, q=0.0, wscale=(0.5 * m.h) ** 2)
res = {}
for plan in ("64,7", "", "64,2", "64,2,4", "64,3,4", "64,2,10", "192,4"):
    _lib.config_set("PLAN_FSDT", plan)
    res[plan] = ops.fsdt_apply(m.geom, *a3, bc, **kw)[0]
ref = res["64,7"]
for plan, r in res.items():
    for k in range(3):
        d = (r[k] - ref[k]).abs()
        bad = (d > 0).nonzero()
        print(repr(plan), k, "max diff", float(d.max()), "count", bad.shape[0], bad[:6].tolist())
