#!/usr/bin/env python3
"""Fused 2-D Q1 loss + gradient by mesh width: rows that are a multiple of four nodes wide take the 16-byte vector path (E = 4), others the
two-element path.  usage: time_2d_sizes.py [B]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import DiffNet2DFEM, PackedMask, ops
dev = torch.device("cuda:0")
sizes = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else [512, 513, 514, 510, 257, 256]
for n in sizes:
    B = max(1, min(64, (64 * 512 * 512) // (n * n)))
    m = DiffNet2DFEM(None, domain_size=n, ngp_1d=3).to(dev)
    shape = (B, 1, n, n)
    g = torch.Generator().manual_seed(1)
    plans = []
    for k in range(4):
        u, nu, f = (torch.rand(shape, generator=g).to(dev) for _ in range(3))
        bc = torch.zeros(shape, dtype=torch.uint8, device=dev); bc[..., 0] = 1; bc[..., -1] = 1
        scale = 1.0 / (B * m.geom.nelem_total)
        plans.append(ops.PoissonPlan(m.geom, u, nu + 0.5, f, None, [(PackedMask.pack(bc), 0.0)], alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=scale,
                                     want_out=True, want_sums=True, loss_scale=scale))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.04:
        for i in range(12):
            plans[i % 4].launch()
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(200):
        plans[i % 4].launch()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 5
    print(f"2-D Q1 {n}^2 B={B} (4 batches in rotation): {us:.1f} us per launch = {16 * B * n * n / us / 1e6:.2f} TB/s algorithmic ({16 * B * n * n / us / 8e6:.3f} of 8 TB/s)", flush=True)
