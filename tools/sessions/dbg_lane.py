import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_gpu_parity import boundary_mask, cu, dev, module, seeded
from diffnet_amd import _lib, ops
sizes = (516, 37)
kw = dict(nsd=2, domain_sizes=sizes, domain_lengths=(1.0, 0.8), domain_size=sizes[0], ngp_1d=3)
m = module(kw)
shape = (1, 1, sizes[1], sizes[0])
u, nu, f = cu(seeded(shape, 31)), cu(seeded(shape, 32) + 0.5), cu(seeded(shape, 33))
bc = boundary_mask((1,) + shape[1:]).to(torch.uint8).to(dev())
src = (seeded(shape, 34) < 0.05).to(torch.uint8).to(dev())
for name, d in (("u8 bc", [(bc, 0.0)]), ("u8 src", [(src, 1.0)]), ("u8 x2", [(src, 1.0), (bc, 0.0)])):
    _lib.config_set("Q1_RULE_KERNEL", "1")
    l0, g0 = m.energy_loss_and_grad(u, nu, f, dirichlet=d, c=0.7)
    _lib.config_set("Q1_RULE_KERNEL", "")
    for plan in ("128,4,16,1", "128,4,2,1", "64,2,16,1"):
        _lib.config_set("PLAN2D", plan)
        l1, g1 = m.energy_loss_and_grad(u, nu, f, dirichlet=d, c=0.7)
        diff = (g1 - g0).abs()[0, 0]
        bad = (diff > 1e-5 * float(g0.abs().max())).nonzero()
        print(name, plan, float(l0), float(l1), "bad nodes:", bad.shape[0], bad[:12].tolist())
    _lib.config_set("PLAN2D", "")
