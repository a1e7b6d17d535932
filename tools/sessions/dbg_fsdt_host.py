import os, sys, time, torch, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffnet_amd import DiffNet2DFEM, ops
from diffnet_amd.elasticity import fsdt_loss
dev = torch.device("cuda:0")
for n in (513, 1025):
    m = DiffNet2DFEM(None, domain_size=n, fem_basis_deg=2, ngp_1d=3).to(dev)
    shape = (1, 1, n, n)
    g = torch.Generator().manual_seed(2)
    fields = [torch.rand(shape, generator=g).to(dev).requires_grad_(True) for _ in range(3)]
    bc = torch.zeros(shape, device=dev); bc[..., 0] = 1
    def fn():
        loss = sum(fsdt_loss(m, *fields, bc))
        torch.autograd.grad(loss, fields)
    for _ in range(20): fn()
    torch.cuda.synchronize()
    ops._CALL_STATS.update(hit=0, miss=0, uncached=0)
    t0 = time.perf_counter()
    for _ in range(40): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(n, "host us/call", (t1 - t0) / 40 * 1e6, "incl. drain", (t2 - t0) / 40 * 1e6, ops._CALL_STATS, "cache entries", len(ops._FSDT_CACHE))
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); t1 = time.perf_counter()
        print("   single call after sync: %.1f us" % ((t1 - t0) * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    fn()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
