#!/usr/bin/env python3
"""Host-side cost of the public fused calls (eager): wall time per call with the device kept busy, call-cache statistics, and a cProfile
of the FSDT loss + backward."""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import DiffNet2DFEM, DiffNet3DFEM, ops
from diffnet_amd.elasticity import _constants, fsdt_loss, fsdt_loss_and_grad, fsdt_total_loss
dev = torch.device("cuda:0")


def wall(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return t


m = DiffNet2DFEM(None, domain_size=64, ngp_1d=2).to(dev)
shape = (1, 1, 64, 64)
u, nu, f = (torch.rand(shape, device=dev) for _ in range(3))
bc = torch.zeros(shape, dtype=torch.uint8, device=dev); bc[..., 0] = 1
out = torch.empty_like(u)
print("energy_loss_and_grad 64^2            : %.1f us/call" % wall(lambda: m.energy_loss_and_grad(u, nu, f, dirichlet=[(bc, 0.0)], c=0.5)))
print("energy_loss_and_grad 64^2, out=      : %.1f us/call" % wall(lambda: m.energy_loss_and_grad(u, nu, f, dirichlet=[(bc, 0.0)], c=0.5, out=out)))
ur = u.clone().requires_grad_(True)


def lossbwd():
    loss = m.energy_loss(ur, nu, f, dirichlet=[(bc, 0.0)], c=0.5)
    torch.autograd.grad(loss, ur)


print("energy_loss + autograd.grad 64^2     : %.1f us/call" % wall(lossbwd))
print("residual_loss 64^2                   : %.1f us/call" % wall(lambda: m.residual_loss(u, nu, f, dirichlet=[(bc, 0.0)])))
n = 513
mq = DiffNet2DFEM(None, domain_size=n, fem_basis_deg=2, ngp_1d=3).to(dev)
flds = [torch.rand((1, 1, n, n), device=dev).requires_grad_(True) for _ in range(3)]
bcf = torch.zeros((1, 1, n, n), device=dev); bcf[..., 0] = 1


def fs():
    loss = sum(fsdt_loss(mq, *flds, bcf))
    torch.autograd.grad(loss, flds)


print("fsdt_loss + autograd.grad 513^2 Q2   : %.1f us/call" % wall(fs))
print("fsdt_total_loss + autograd.grad       : %.1f us/call" % wall(lambda: torch.autograd.grad(fsdt_total_loss(mq, *flds, bcf), flds)))
print("fsdt_loss_and_grad (no graph)         : %.1f us/call" % wall(lambda: fsdt_loss_and_grad(mq, *flds, bcf)))
bufs = [t.detach().clone() for t in flds]
plan = ops.FsdtPlan(mq.geom, *bufs, bcf, q=1.0, wscale=(0.5 * mq.h) * (0.5 * mq.h), **_constants(1.0, 0.25, 0.1, 1.0))
print("FsdtPlan.launch (prepared)            : %.1f us/call" % wall(plan.launch))
print("call cache:", ops._CALL_STATS)
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    fs()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
