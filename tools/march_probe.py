#!/usr/bin/env python3
"""Probe kernels (dn_probe_stream / dn_probe_march) on four sets of 64 x 512 x 512 fp32 arrays in rotation (nothing is found in the Infinity
Cache): what the memory system delivers for the plain 3-read / 1-write stream in each form, and for the marching access pattern of the fused
2-D kernel without arithmetic as a function of the strip height R (waves per SIMD) and the rows requested ahead per wave."""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import _lib
dev = torch.device("cuda:0")
B, n = 64, 512
g = torch.Generator().manual_seed(1)
sets = [[torch.rand((B, n, n), generator=g).to(dev) for _ in range(3)] + [torch.empty((B, n, n), device=dev)] for _ in range(4)]
L = _lib.lib()
stream = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
alg = 16 * B * n * n


def timeit(fn, reps=300):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.03:
        for i in range(24):
            fn(i % 4)
        torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(reps):
            fn(i % 4)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / reps * 1e3)
    return sorted(ts)[1]


for mode in (0, 2, 3):
    def fn(k, mode=mode):
        a, b, c, o = sets[k]
        rc = L.dn_probe_stream(a.data_ptr(), b.data_ptr(), c.data_ptr(), o.data_ptr(), a.numel(), mode, stream())
        assert rc == 0, rc
    us = timeit(fn)
    print(f"stream form {mode >> 2} nt_loads {(mode >> 1) & 1} nt_stores {mode & 1}: {us:7.2f} us  {alg / us / 1e6:6.2f} TB/s algorithmic", flush=True)

for flags in (7,):
    for R in (16,):
        for D in (1,):
            def fn(k, R=R, D=D, flags=flags):
                a, b, c, o = sets[k]
                rc = L.dn_probe_march(a.data_ptr(), b.data_ptr(), c.data_ptr(), o.data_ptr(), B, n, R, D, flags, stream())
                assert rc == 0, rc
            us = timeit(fn)
            print(f"march {'PAIRED rows ' if flags & 32 else ''}halo {flags & 1} shared-node {'lane' if flags & 16 else (flags >> 1) & 1} nt_stores {(flags >> 2) & 1} nt_loads {(flags >> 3) & 1}  R {R:3d} ({64 * (n // R) * 2 / 1024:4.1f} waves/SIMD) rows ahead {D}: "
                  f"{us:7.2f} us  {alg / us / 1e6:6.2f} TB/s algorithmic", flush=True)

for flags in (0, 8, 64, 72):
    for threads, TR in ((256, 4), (256, 8), (512, 8), (256, 16), (512, 16)):
        def fn(k, threads=threads, TR=TR, flags=flags):
            a, b, c, o = sets[k]
            rc = L.dn_probe_tile(a.data_ptr(), b.data_ptr(), c.data_ptr(), o.data_ptr(), B, n, TR, threads, flags, stream())
            assert rc == 0, rc
        us = timeit(fn)
        print(f"tile  rows {TR:2d} (+2 halo) threads {threads} nt_loads {(flags >> 3) & 1} xcd-order {(flags >> 6) & 1}: {us:7.2f} us  {alg / us / 1e6:6.2f} TB/s algorithmic", flush=True)
