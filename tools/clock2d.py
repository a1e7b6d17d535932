#!/usr/bin/env python3
"""Shader clock the 2-D bench kernel really runs at (diagnostic build: tools/variant_build.sh stamp2d poisson2d_q1_cf.hip -DDN_STAMP2D;
DN_LIB_PATH=variants/libdn_stamp2d.so python tools/clock2d.py [B])."""
import ctypes as C
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import BoxFaces, DiffNet2DFEM, _lib   # noqa: E402

dev = torch.device("cuda:0")
n = 512
for B in [int(v) for v in (sys.argv[1:] or ["64", "16", "4"])]:
    m = DiffNet2DFEM(None, domain_size=n, ngp_1d=3).to(dev)
    shape = (B, 1, n, n)
    g = torch.Generator().manual_seed(1)
    u, nu, f = (torch.rand(shape, generator=g).to(dev) for _ in range(3))
    nu += 0.5
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for _ in range(10):
        m.energy_loss_and_grad(u, nu, f, dirichlet=[(BoxFaces(), 0.0)], c=1.0)
    for a, b in evs:
        a.record(); m.energy_loss_and_grad(u, nu, f, dirichlet=[(BoxFaces(), 0.0)], c=1.0); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    buf = np.zeros(8192 * 4, dtype=np.uint64)
    h = _lib.lib()
    h.dn_debug_stamps2d.argtypes = [C.c_void_p, C.c_size_t]
    assert h.dn_debug_stamps2d(buf.ctypes.data, buf.nbytes) == 0
    q = buf.reshape(-1, 4)
    q = q[q[:, 1] > 0]
    t0 = q[:, 1].min()
    st, en, fin = ((q[:, k] - t0).astype(np.float64) * 0.01 for k in (1, 2, 3))        # us after the first workgroup started
    life = en - st
    mhz = q[:, 0].astype(np.float64) / (life * 1.0)
    pc = lambda a: " ".join(f"{np.percentile(a, p):6.1f}" for p in (0, 10, 50, 90, 100))
    print(f"2-D 512^2 B={B}: kernel median {ts[15]:.1f} us (events around one eager call); {len(q)} workgroups, march {life.mean():.1f} us on average; SHADER CLOCK mean {mhz.mean():.0f} MHz  min {mhz.min():.0f}  max {mhz.max():.0f}")
    print(f"   timeline of the last launch, us after its first workgroup started (min p10 p50 p90 max):  start {pc(st)}   march end {pc(en)}   kernel end {pc(fin)}", flush=True)
    if os.environ.get("DETAIL"):
        slot = np.nonzero(buf.reshape(-1, 4)[:, 1] > 0)[0]
        xcd = slot % 8
        strips = 32
        strip = slot % strips if True else 0
        sample = slot // strips
        print("   march end by XCD (slot % 8):      " + " ".join(f"{en[xcd == x].mean():5.1f}/{en[xcd == x].max():5.1f}" for x in range(8)))
        print("   march end by strip index (mean):  " + " ".join(f"{en[strip == k].mean():5.1f}" for k in range(strips)))
        print("   march end by sample (mean, 0..63):" + " ".join(f"{en[sample == k].mean():5.1f}" for k in range(0, B, max(B // 16, 1))))
        print("   start by sample (mean):           " + " ".join(f"{st[sample == k].mean():5.1f}" for k in range(0, B, max(B // 16, 1))))
        order = np.argsort(en)
        print("   slowest 12 workgroups (slot, xcd, strip, sample, end):", [(int(slot[i]), int(xcd[i]), int(strip[i]), int(sample[i]), round(float(en[i]), 1)) for i in order[-12:]])
        print("   fastest 6:", [(int(slot[i]), int(xcd[i]), int(strip[i]), int(sample[i]), round(float(en[i]), 1)) for i in order[:6]])
