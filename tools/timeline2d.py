#!/usr/bin/env python3
"""Per-workgroup timeline of the 2-D bench launch in the bench regime (prepared launches over 4 batches in rotation, back to back; diagnostic
build: tools/variant_build.sh stamp2d "-DDN_STAMP2D"; DN_LIB_PATH=variants/libdn_stamp2d.so python tools/timeline2d.py [bits|box] [PLAN2D])."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import BoxFaces, DiffNet2DFEM, PackedMask, _lib, ops
dev = torch.device("cuda:0")
n, B = 512, 64
m = DiffNet2DFEM(None, domain_size=n, ngp_1d=3).to(dev)
shape = (B, 1, n, n)
g = torch.Generator().manual_seed(1)
form = sys.argv[1] if len(sys.argv) > 1 else "bits"
if len(sys.argv) > 2:
    _lib.config_set("PLAN2D", sys.argv[2])
scale = 1.0 / (B * m.geom.nelem_total)
kw = dict(alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=scale, want_out=True, want_sums=True, loss_scale=scale)
bc = torch.zeros(shape, dtype=torch.uint8, device=dev)
bc[..., 0] = 1; bc[..., -1] = 1; bc[..., 0, :] = 1; bc[..., -1, :] = 1
cond = {"box": lambda: [(BoxFaces(), 0.0)], "bits": lambda: [(PackedMask.pack(bc.clone()), 0.0)]}[form]
plans = []
for k in range(4):
    u, nu, f = (torch.rand(shape, generator=g).to(dev) for _ in range(3))
    plans.append(ops.PoissonPlan(m.geom, u, nu + 0.5, f, None, cond(), **kw))
h = _lib.lib()
h.dn_debug_stamps2d.argtypes = [C.c_void_p, C.c_size_t]
pc = lambda a: " ".join(f"{np.percentile(a, p):6.1f}" for p in (0, 10, 50, 90, 100))
for rep in range(3):
    for i in range(400):
        plans[i % 4].launch()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(200):
        plans[i % 4].launch()
    b.record()
    torch.cuda.synchronize()
    buf = np.zeros(8192 * 4, dtype=np.uint64)
    assert h.dn_debug_stamps2d(buf.ctypes.data, buf.nbytes) == 0
    q = buf.reshape(-1, 4)
    slot = np.nonzero(q[:, 1] > 0)[0]
    q = q[slot]
    t0 = q[:, 1].min()
    st, en, fin = ((q[:, k] - t0).astype(np.float64) * 0.01 for k in (1, 2, 3))
    nstr = len(q) // B
    strip, sample = slot % nstr, slot // nstr
    print(f"{form} {sys.argv[2] if len(sys.argv) > 2 else 'default'} rep {rep}: {a.elapsed_time(b) * 5:.2f} us per launch back to back; {len(q)} workgroups ({nstr} per sample)")
    print(f"   start {pc(st)}   march end {pc(en)}   kernel end {pc(fin)}   (us after the launch's first workgroup started: min p10 p50 p90 max)")
    print("   march end by sample group of 8 (dispatch order):  " + " ".join(f"{en[(sample // 8) == k].mean():5.1f}" for k in range(B // 8)))
    print("   march end by strip % 8:                           " + " ".join(f"{en[(strip % 8) == k].mean():5.1f}" for k in range(8)))
    print("   march end by strip index:                         " + " ".join(f"{en[strip == k].mean():5.1f}" for k in range(nstr)))
    print("   march duration by sample group / strip%8 spread inside a group (std): %.2f" % np.mean([en[(sample // 8 == a_) & (strip % 8 == b_)].std() for a_ in range(B // 8) for b_ in range(8)]), flush=True)
