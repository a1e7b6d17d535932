#!/usr/bin/env python3
"""One rank's share of BASELINE configs[3] on ONE GPU (no collective): rank 3 of 8 of the 256^3 mesh = a slab of 33 node planes.  Device time
of its evaluation as the slab path launches it -- the strips next to the two faces first (after which the interface exchange would start),
then the interior -- against the single-launch evaluation of the same slab; and of rank 7's slab (32 planes, one face).  Run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel durations."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import ops
from diffnet_amd.slab import SlabPoisson
dev = torch.device("cuda:0")
n, world = 256, 8


def timed(fn, reps=200):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.04:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / reps * 1e3)
    return sorted(ts)[1]


for rank in (3, 7):
    sp = SlabPoisson(3, (n, n, n), (1.0, 1.0, 1.0), rank, world, ngp_1d=2, device=dev)
    dec = sp.dec
    nzl = dec.n1 - dec.n0 + 1
    shape = (1, 1, nzl, n, n)
    g = torch.Generator().manual_seed(rank)
    u, nu, f = (torch.rand(shape, generator=g).to(dev) for _ in range(3))
    nu += 0.5
    bc = torch.zeros(shape, dtype=torch.uint8, device=dev)
    bc[..., 0] = 1; bc[..., -1] = 1; bc[..., 0, :] = 1; bc[..., -1, :] = 1
    if rank == world - 1:
        bc[:, :, -1] = 1
    scale = 1.0 / dec.nel_global
    first, rest, local = sp._plans(u, nu, f, [(bc, 0.0)], 1.0, 1.0, scale)
    whole = ops.PoissonPlan(sp.fem.geom, u, nu, f, None, local, alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=scale, want_out=True, want_sums=True, loss_scale=scale)
    t_first, t_rest = timed(first.launch), timed(rest.launch)
    t_both = timed(lambda: (first.launch(), rest.launch()))
    t_whole = timed(whole.launch)
    print(f"rank {rank} of {world}: slab of {nzl} node planes ({nzl - 1} element layers): face strips {t_first:.1f} us, interior {t_rest:.1f} us, "
          f"both back to back {t_both:.1f} us; the slab in one launch {t_whole:.1f} us; interface layer = {4 * n * n / 1024:.0f} KiB per face", flush=True)
