#!/usr/bin/env python3
"""What a pair of HIP events around each launch adds to it, by event flavour: torch.cuda.Event (hipEventDefault: system-scope release at the
record), hipEventReleaseToDevice, hipEventDisableSystemFence -- against ONE pair of events around the same number of launches."""
import ctypes
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import BoxFaces, DiffNet2DFEM, ops   # noqa: E402

dev = torch.device("cuda:0")
m = DiffNet2DFEM(None, domain_size=512, ngp_1d=3).to(dev)
shape = (64, 1, 512, 512)
g = torch.Generator().manual_seed(1)
u, nu, f = (torch.rand(shape, generator=g).to(dev) for _ in range(3))
nu += 0.5
scale = 1.0 / (64 * m.geom.nelem_total)
pl = ops.PoissonPlan(m.geom, u, nu, f, None, [(BoxFaces(), 0.0)], alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=scale, want_out=True, want_sums=True, loss_scale=scale)
hip = ctypes.CDLL("libamdhip64.so")
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def hip_pairs(flags, n):
    evs = []
    for _ in range(2 * n):
        e = ctypes.c_void_p()
        assert hip.hipEventCreateWithFlags(ctypes.byref(e), ctypes.c_uint(flags)) == 0
        evs.append(e)
    for i in range(n):
        hip.hipEventRecord(evs[2 * i], stream)
        pl.launch()
        hip.hipEventRecord(evs[2 * i + 1], stream)
    torch.cuda.synchronize()
    out = []
    for i in range(n):
        ms = ctypes.c_float()
        assert hip.hipEventElapsedTime(ctypes.byref(ms), evs[2 * i], evs[2 * i + 1]) == 0
        out.append(ms.value * 1e3)
    for e in evs:
        hip.hipEventDestroy(e)
    return out


def torch_pairs(n):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record(); pl.launch(); b.record()
    torch.cuda.synchronize()
    return [a.elapsed_time(b) * 1e3 for a, b in evs]


def region(n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        pl.launch()
    b.record()
    torch.cuda.synchronize()
    return [a.elapsed_time(b) * 1e3 / n]


t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.05:
    for _ in range(20):
        pl.launch()
    torch.cuda.synchronize()
kinds = {"one pair around 100 launches": lambda: region(100), "torch.cuda.Event pairs (hipEventDefault)": lambda: torch_pairs(100),
         "hipEventReleaseToDevice pairs": lambda: hip_pairs(0x40000000, 100), "hipEventDisableSystemFence pairs": lambda: hip_pairs(0x20000000, 100),
         "hipEventDefault pairs (ctypes)": lambda: hip_pairs(0, 100)}
for rnd in range(3):
    for name, fn in kinds.items():
        for _ in range(20):
            pl.launch()
        v = sorted(fn())
        print(f"round {rnd} {name:44s} mean {sum(v) / len(v):6.2f} us  median {v[len(v) // 2]:6.2f}  min {v[0]:6.2f}", flush=True)
