#!/usr/bin/env python3
"""Fused InstanceNorm + LeakyReLU (dn_instnorm_act_*) vs torch's InstanceNorm2d + LeakyReLU on the UNet's activations."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
from diffnet_amd.networks.fused import InstanceNormAct

dev = torch.device("cuda", 0)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for shape in [(64, 32, 256, 256), (64, 64, 128, 128), (64, 128, 64, 64), (64, 256, 32, 32), (64, 512, 4, 4), (2, 16, 64, 64, 64)]:
    x = torch.randn(shape, device=dev, requires_grad=True)
    cot = torch.randn(shape, device=dev)
    IN = nn.InstanceNorm3d if len(shape) == 5 else nn.InstanceNorm2d
    ref = nn.Sequential(IN(shape[1]), nn.LeakyReLU(0.2))
    fus = InstanceNormAct(shape[1], slope=0.2)
    def run(mod):
        y = mod(x)
        y.backward(cot)
        x.grad = None
    gb = x.numel() * 4 / 1e9
    tr, tf = timeit(lambda: run(ref)), timeit(lambda: run(fus))
    print(f"{str(shape):26s} torch {tr:8.1f} us   fused {tf:8.1f} us   x{tr / tf:.2f}   fused eff. {5 * gb / (tf * 1e-6):.0f} GB/s (5 passes)", flush=True)
