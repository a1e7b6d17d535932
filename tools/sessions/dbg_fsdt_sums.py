import os, sys, torch, numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from test_gpu_parity import boundary_mask, cu, dev, module, seeded
from diffnet_amd import _lib, ops
deg, ngp, sizes, B = 2, 4, (257, 33), 1
kw = dict(nsd=2, domain_sizes=sizes, domain_lengths=(1.0, 0.9), domain_size=sizes[0], fem_basis_deg=deg, ngp_1d=ngp)
m = module(kw)
shape = (B, 1, sizes[1], sizes[0])
flds = [cu(seeded(shape, 90 + i)) for i in range(3)]
bcf = boundary_mask(shape).to(dev())
bcf[0, 0, sizes[1] // 2, 3:9] = 1.0
consts = dict(D11=1.3, D12=0.4, D22=1.1, D66=0.6, A44=0.8, A55=0.9, q=1.2, wscale=0.3)
for mask in (None, bcf, bcf.to(torch.uint8)):
    for plan in ["192,3", "64,1,2", "64,2,3", "64,1,12", "64,3,5", "64,5,4", "64,2,9", "64,4,7", "", "192,3"]:
        _lib.config_set("PLAN_FSDT", plan)
        got, sums, norms = ops.fsdt_apply(m.geom, *flds, mask, (0.1, -0.2, 0.3), want_norms=True, **consts)
        direct = [float((g.double() ** 2).sum()) for g in got]
        print(None if mask is None else mask.dtype, repr(plan), [f"{float(s):.6e}" for s in sums], [f"{d:.6e}" for d in direct])
_lib.config_set("PLAN_FSDT", "")
