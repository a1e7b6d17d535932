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
    buf = np.zeros(8192 * 2, dtype=np.uint64)
    h = _lib.lib()
    h.dn_debug_stamps2d.argtypes = [C.c_void_p, C.c_size_t]
    assert h.dn_debug_stamps2d(buf.ctypes.data, buf.nbytes) == 0
    r = buf.reshape(-1, 2).astype(np.float64)
    r = r[r[:, 1] > 0]
    mhz = r[:, 0] / (r[:, 1] * 10e-3)
    print(f"2-D 512^2 B={B}: kernel median {ts[15]:.1f} us; {len(r)} workgroups, lifetime {r[:,1].mean()*0.01:.1f} us on average; SHADER CLOCK mean {mhz.mean():.0f} MHz  min {mhz.min():.0f}  max {mhz.max():.0f}", flush=True)
