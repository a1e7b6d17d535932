#!/usr/bin/env python3
"""3-D Q2 / Q3 fused Poisson loss + gradient (poisson3d_gen.hip: three launches) against the same loss composed from the single-launch operators
behind autograd (what the reference's loss body does): usage time_q2_3d.py n degree [B]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import DiffNet3DFEM, ops
dev = torch.device("cuda:0")
n, deg = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
m = DiffNet3DFEM(None, domain_size=n, nsd=3, fem_basis_deg=deg, ngp_1d=3).to(dev)
shape = (B, 1, n, n, n)
g = torch.Generator().manual_seed(3)
u, nu, f = (torch.rand(shape, generator=g).to(dev) for _ in range(3))
nu += 0.5
bc = torch.zeros(shape, device=dev)
bc[..., 0] = 1; bc[..., -1] = 1


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def composed():
    ur = u.clone().requires_grad_(True)
    uu = torch.where(bc > 0.5, torch.zeros_like(ur), ur)
    ux, uy, uz = m.gauss_pt_evaluation_der_x(uu), m.gauss_pt_evaluation_der_y(uu), m.gauss_pt_evaluation_der_z(uu)
    ug, ng, fg = m.gauss_pt_evaluation(uu), m.gauss_pt_evaluation(nu), m.gauss_pt_evaluation(f)
    w = m.gpw.to(dev).reshape(1, -1, 1, 1, 1) * jac
    loss = torch.mean(torch.sum(w * (0.5 * ng * (ux * ux + uy * uy + uz * uz) - ug * fg), 1))
    loss.backward()
    return loss


jac = (0.5 * m.hx) * (0.5 * m.hy) * (0.5 * m.hz)
fused = lambda: m.energy_loss_and_grad(u, nu, f, dirichlet=[(bc, 0.0)], c=0.5, jac=jac)
lf, _ = fused()
lc = composed()
print(f"3-D Q{deg} {n}^3 nodes B={B}: fused {timeit(fused):.1f} us  (loss {float(lf):.6g});  composed from operators + autograd {timeit(composed, 10):.1f} us (loss {float(lc.detach()):.6g})")
