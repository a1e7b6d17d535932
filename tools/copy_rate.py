#!/usr/bin/env python3
"""Steady-state rate of plain device copies and fills on this GPU (40 ms of load first, 100 back to back between one pair of events):
the achievable HBM rate the streaming kernels are compared with (the 8 TB/s of the roofline is the specification peak)."""
import time
import torch
dev = torch.device("cuda:0")
for mib in (64, 256, 1024):
    n = mib << 18
    a = torch.rand(n, device=dev); b = torch.empty_like(a); c = torch.rand(n, device=dev)
    ops = {"copy  b = a          (1 read + 1 write)": (lambda: b.copy_(a), 2),
           "add   b = a + c      (2 reads + 1 write)": (lambda: torch.add(a, c, out=b), 3),
           "fill  b = 0          (1 write)": (lambda: b.zero_(), 1),
           "sum   a.sum()        (1 read)": (lambda: a.sum(), 1)}
    for name, (fn, k) in ops.items():
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.04:
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 10
        print(f"{mib:5d} MiB  {name}: {us:8.1f} us  {k * n * 4 / us / 1e6:6.2f} TB/s", flush=True)
