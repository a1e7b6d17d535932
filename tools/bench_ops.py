#!/usr/bin/env python3
"""Generic operators (the drop-in path every non-fused example body uses): gauss_pt_eval forward / backward and element->node
assembly, device time and achieved HBM bandwidth (bytes = input read + output written)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffnet_amd import DiffNet2DFEM, DiffNet3DFEM

dev = torch.device("cuda", 0)


def timed(fn, n=20):
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


for nsd, n, ngp, deg, B in [(2, 512, 2, 1, 16), (2, 512, 3, 1, 16), (2, 513, 3, 2, 16), (3, 128, 2, 1, 2), (3, 65, 3, 2, 2)]:
    cls = DiffNet3DFEM if nsd == 3 else DiffNet2DFEM
    m = cls(None, domain_size=n, ngp_1d=ngp, nsd=nsd, fem_basis_deg=deg).to(dev)
    u = torch.rand((B, 1, *m.geom.node_shape), device=dev, requires_grad=True)
    y = m.gauss_pt_evaluation_der_x(u)
    g = torch.rand_like(y)
    nin, nout = u.numel() * 4, y.numel() * 4
    t_f = timed(lambda: m.gauss_pt_evaluation_der_x(u.detach()))
    t_b = timed(lambda: torch.autograd.grad(m.gauss_pt_evaluation_der_x(u), u, g)) - t_f
    r = torch.rand((B, m.nbf_total, *m.geom.elem_shape), device=dev)
    t_a = timed(lambda: m.assemble(r))
    print(f"{nsd}-D n={n} Q{deg} ngp={ngp} B={B}: eval fwd {t_f:7.1f} us ({(nin + nout) / t_f / 1e3:6.0f} GB/s)   bwd {t_b:7.1f} us "
          f"({(nin + nout) / max(t_b, 1e-3) / 1e3:6.0f} GB/s)   assemble {t_a:7.1f} us ({(r.numel() * 4 + nin) / t_a / 1e3:6.0f} GB/s)", flush=True)
