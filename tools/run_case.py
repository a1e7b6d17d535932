#!/usr/bin/env python3
"""One fused energy loss + gradient case launched N times (for rocprofv3): python tools/run_case.py nsd n B ngp [PLAN] [reps]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import BoxFaces, DiffNet2DFEM, DiffNet3DFEM, PackedMask, _lib   # noqa: E402

nsd, n, B, ngp = (int(v) for v in sys.argv[1:5])
plan = sys.argv[5] if len(sys.argv) > 5 else ""
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
dev = torch.device("cuda:0")
m = (DiffNet3DFEM if nsd == 3 else DiffNet2DFEM)(None, domain_size=n, nsd=nsd, ngp_1d=ngp).to(dev)
shape = (B, 1, *m.geom.node_shape)
g = torch.Generator().manual_seed(1)
u, nu, f = (torch.rand(shape, generator=g).to(dev) for _ in range(3))
nu += 0.5
bc = torch.zeros(shape, dtype=torch.uint8, device=dev)
for d in range(2, len(shape)):
    idx = [slice(None)] * len(shape); idx[d] = 0; bc[tuple(idx)] = 1; idx[d] = -1; bc[tuple(idx)] = 1
if plan:
    _lib.config_set("PLAN3D" if nsd == 3 else "PLAN2D", plan)
form = os.environ.get("DN_BC_FORM", "u8")          # how the Dirichlet condition is held: u8 | bits | box | f32
dirichlet = {"u8": [(bc, 0.0)], "f32": [(bc.float(), 0.0)], "bits": [(PackedMask.pack(bc), 0.0)], "box": [(BoxFaces(), 0.0)]}[form]
if os.environ.get("DN_LOAD") == "1":                # round 4: the forcing as its assembled load vector (3-D two-element kernel)
    from diffnet_amd import LoadVector
    f = LoadVector.assemble(m.geom, f)
if os.environ.get("DN_SUMS") == "fold":             # round 4: prepared launches, every launch forms the scalars of the one before it
    from diffnet_amd import ops
    scale = 1.0 / (B * m.geom.nelem_total)
    plans = [ops.PoissonPlan(m.geom, u, nu, f, None, dirichlet, alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=scale, want_out=True,
                             want_sums=True, loss_scale=scale, pipelined_sums=True) for _ in range(2)]
    plans[0].fold(plans[1]); plans[1].fold(plans[0])
    for i in range(reps):
        plans[i & 1].launch()
    plans[(reps - 1) & 1].finish_sums()
else:
    for _ in range(reps):
        m.energy_loss_and_grad(u, nu, f, dirichlet=dirichlet, c=1.0)
torch.cuda.synchronize()
print("done")
