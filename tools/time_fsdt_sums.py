import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import DiffNet2DFEM, _lib, ops
dev = torch.device("cuda:0")
n, deg = 1025, 2
m = DiffNet2DFEM(None, domain_size=n, fem_basis_deg=deg, ngp_1d=3).to(dev)
def timed(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.04:
        for _ in range(10): fn()
        torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / 50)
    return sorted(ts)[1]
for B in (1, 8):
    shape = (B, 1, n, n)
    g = torch.Generator().manual_seed(2)
    flds = [torch.rand(shape, generator=g).to(dev) for _ in range(3)]
    bc = torch.zeros(shape, device=dev); bc[..., 0] = 1; bc[..., -1] = 1; bc[..., 0, :] = 1; bc[..., -1, :] = 1
    bc8 = bc.to(torch.uint8)
    for label, mask in (("f32 mask", bc), ("u8 mask", bc8), ("no mask", None)):
        for sums in (True, False):
            t = timed(lambda: ops.fsdt_apply(m.geom, *flds, mask, q=1.0, wscale=(0.5 * m.h) ** 2, want_sums=sums))
            print(f"B={B} {label} sums={sums}: {t:.1f} us", flush=True)
