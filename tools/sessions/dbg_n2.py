import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_gpu_parity import cu, dev, module, seeded
from diffnet_amd import _lib, ops
for sizes in [(34, 17, 3), (34, 16, 3), (32, 17, 3), (34, 17, 4), (34, 17, 2), (62, 32, 6)]:
    kw = dict(nsd=3, domain_sizes=sizes, domain_lengths=(1.0, 0.7, 1.4), domain_size=sizes[0], ngp_1d=2)
    m = module(kw)
    for B in (1, 3):
        shape = (B, 1, sizes[2], sizes[1], sizes[0])
        u, nu, f = cu(seeded(shape, 502)), cu(seeded(shape, 602) + 0.5), cu(seeded(shape, 702))
        bc = (seeded(shape, 802) < 0.15).to(torch.uint8).to(dev()); bc[..., 0] = 1
        for plan in ("", "16,16,2,1", "16,16,2,2"):
            for name, d in (("none", []), ("u8", [(bc, 0.3)]), ("u8 val0", [(bc, 0.0)])):
                res = {}
                for e1 in ("", "1"):
                    for esum in ("", "1"):
                        _lib.config_set("PLAN3D", plan if not e1 else ""); _lib.config_set("Q1_3D_E1", e1); _lib.config_set("Q1_3D_E1SUM", esum)
                        out, sums = ops.poisson_apply(m.geom, u, nu, f, None, d, alpha=1.0, beta=1.0, c=0.5)
                        res[(e1, esum)] = (float(sums[0]), float(sums[1]))
                _lib.config_set("PLAN3D", ""); _lib.config_set("Q1_3D_E1", ""); _lib.config_set("Q1_3D_E1SUM", "")
                ref = res[("1", "1")]
                bad = [k for k, v in res.items() if abs(v[0] - ref[0]) > 1e-4 * abs(ref[0]) + 1e-6 or abs(v[1] - ref[1]) > 1e-4 * abs(ref[1]) + 1e-9]
                if bad:
                    print(sizes, "B", B, "plan", repr(plan), name, {k: res[k] for k in res})
print("done")
