import torch, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import DiffNet2DFEM, ops
from diffnet_amd.elasticity import fsdt_loss_and_grad
dev = torch.device("cuda:0")
m = DiffNet2DFEM(None, domain_size=1025, fem_basis_deg=2, ngp_1d=3).to(dev)
def timed(fn):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:
        for _ in range(10): fn()
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / 50
for B in (1, 8):
    shape = (B, 1, 1025, 1025)
    g = torch.Generator().manual_seed(2)
    f = [torch.rand(shape, generator=g).to(dev) for _ in range(3)]
    bc = torch.zeros(shape, device=dev); bc[..., 0] = 1; bc[..., -1] = 1; bc[..., 0, :] = 1; bc[..., -1, :] = 1
    for mask in (bc, bc.to(torch.uint8)):
        t = timed(lambda: fsdt_loss_and_grad(m, *f, mask))
        plan = ops.FsdtPlan(m.geom, *f, mask, (0.0, 0.0, 0.0), q=1.0, wscale=(0.5 * m.h) ** 2)
        tp = timed(plan.launch)
        print(f"B={B} mask {mask.dtype}: fsdt_loss_and_grad {t:.1f} us   FsdtPlan.launch {tp:.1f} us", flush=True)
