#!/usr/bin/env python3
"""Per-layer forward / backward time of the GoodGenerator's 3-D convolutions at 128^3 (MIOpen), to see which layer the
slow weight-gradient kernels belong to."""
import torch, time
from torch import nn
dev = torch.device("cuda", 0)


def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


layers = [("down1 Conv3d 1->16 k4s2", nn.Conv3d(1, 16, 4, 2, 1, bias=False), (1, 1, 128, 128, 128)),
          ("down2 Conv3d 16->32 k4s2", nn.Conv3d(16, 32, 4, 2, 1, bias=False), (1, 16, 64, 64, 64)),
          ("down3 Conv3d 32->64 k4s2", nn.Conv3d(32, 64, 4, 2, 1, bias=False), (1, 32, 32, 32, 32)),
          ("down4 Conv3d 64->128 k4s2", nn.Conv3d(64, 128, 4, 2, 1, bias=False), (1, 64, 16, 16, 16)),
          ("down5 Conv3d 128->128 k4s2", nn.Conv3d(128, 128, 4, 2, 1, bias=False), (1, 128, 8, 8, 8)),
          ("up3 ConvT3d 128->128", nn.ConvTranspose3d(128, 128, 4, 2, 1, bias=False), (1, 128, 4, 4, 4)),
          ("up4 ConvT3d 256->64", nn.ConvTranspose3d(256, 64, 4, 2, 1, bias=False), (1, 256, 8, 8, 8)),
          ("up5 ConvT3d 128->32", nn.ConvTranspose3d(128, 32, 4, 2, 1, bias=False), (1, 128, 16, 16, 16)),
          ("up6 ConvT3d 64->16", nn.ConvTranspose3d(64, 16, 4, 2, 1, bias=False), (1, 64, 32, 32, 32)),
          ("final Conv3d 32->1 k3 (on upsampled 128^3)", nn.Conv3d(32, 1, 3, padding=1), (1, 32, 128, 128, 128))]
for name, m, shape in layers:
    m = m.to(dev)
    x = torch.randn(shape, device=dev, requires_grad=True)
    y = m(x)
    g = torch.randn_like(y)
    fwd = t(lambda: m(x))
    bwd_x = t(lambda: torch.autograd.grad(m(x), x, g))
    bwd_w = t(lambda: torch.autograd.grad(m(x), m.weight, g))
    print(f"{name:46s} fwd {fwd:8.2f} ms   fwd+bwd_data {bwd_x:8.2f} ms   fwd+bwd_weight {bwd_w:8.2f} ms", flush=True)
