#!/usr/bin/env python3
"""Generic operators (the drop-in path every non-fused example body uses): gauss_pt_eval forward, its adjoint and the element->node
assembly.  DEVICE time of the launch behind each operator (ops._gpe_fwd / _gpe_bwd / _assemble_raw: the ctypes call on preallocated shapes,
30 launches between one pair of events) and achieved HBM bandwidth (bytes = input read + output written); next to it the HOST-bound time of the
same operator behind autograd (forward + backward of the registered operator, what rounds 2-3 reported as "bwd": the dispatcher's ~30 us per call
hid the kernels, profiles/r2_ops_b.txt)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffnet_amd import DiffNet2DFEM, DiffNet3DFEM, ops

dev = torch.device("cuda", 0)


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(20_000_000)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def wall(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


cases = [(2, 512, 2, 1, 16), (2, 512, 3, 1, 16), (2, 512, 4, 1, 16), (2, 513, 3, 2, 16), (2, 511, 4, 3, 16), (3, 128, 2, 1, 2), (3, 65, 3, 2, 2), (3, 129, 3, 2, 1),
         (3, 64, 4, 3, 1)]
for nsd, n, ngp, deg, B in cases:
    cls = DiffNet3DFEM if nsd == 3 else DiffNet2DFEM
    m = cls(None, domain_size=n, ngp_1d=ngp, nsd=nsd, fem_basis_deg=deg).to(dev)
    u = torch.rand((B, 1, *m.geom.node_shape), device=dev, requires_grad=True)
    y = m.gauss_pt_evaluation_der_x(u)
    g = torch.rand_like(y)
    nin, nout = u.numel() * 4, y.numel() * 4
    nbf, stride = deg + 1, deg
    tables = m._stacked("dN_x_gp", dev)
    ud = u.detach()
    t_f = timed(lambda: ops._gpe_fwd(ud, tables, nsd, nbf, stride))
    t_b = timed(lambda: ops._gpe_bwd(g, tables, tuple(u.shape), nsd, nbf, stride))
    r = torch.rand((B, m.nbf_total, *m.geom.elem_shape), device=dev)
    t_a = timed(lambda: ops._assemble_raw(r, nsd, nbf, None))
    h_fb = wall(lambda: torch.autograd.grad(m.gauss_pt_evaluation_der_x(u), u, g))
    print(f"{nsd}-D n={n} Q{deg} ngp={ngp} B={B}: eval fwd {t_f:7.1f} us ({(nin + nout) / t_f / 1e3:6.0f} GB/s)   adjoint {t_b:7.1f} us "
          f"({(nin + nout) / t_b / 1e3:6.0f} GB/s)   assemble {t_a:7.1f} us ({(r.numel() * 4 + nin) / t_a / 1e3:6.0f} GB/s)   "
          f"[autograd fwd + bwd, wall: {h_fb:7.1f} us]", flush=True)
