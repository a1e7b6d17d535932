#!/usr/bin/env python3
"""Per-layer device time and TFLOP/s of the 2-D k4s2 conv family (dn_conv2d_k4s2_down / _up / _wrw) on the U-Net layer shapes
(512 x 512 input, batch 16), next to torch's (MIOpen) convolution of the same layer."""
import os
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd.networks import fused   # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
miopen = "--miopen" in sys.argv


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


# (name, C fine channels, M coarse channels, coarse H = W)
layers = [("down1/conv 2->32", 2, 32, 256), ("down2 32->64", 32, 64, 128), ("down3 64->128", 64, 128, 64), ("down4 128->256", 128, 256, 32),
          ("down5 256->256", 256, 256, 16), ("up1 convT 256->256", 256, 256, 16), ("up2 convT 512->128", 128, 512, 32),
          ("up3 convT 256->64", 64, 256, 64), ("up4 convT 128->32", 32, 128, 128)]
tot = {"down": 0.0, "up": 0.0, "wrw": 0.0}
for name, C, M, H in layers:
    fine = torch.randn(B, C, 2 * H, 2 * H, device=dev)
    coarse = torch.randn(B, M, H, H, device=dev)
    w = torch.randn(M, C, 4, 4, device=dev) * 0.05
    fl = 2.0 * B * H * H * M * C * 16
    td = timeit(lambda: fused._c2_down(fine, w))
    tu = timeit(lambda: fused._c2_up(coarse, w))
    tw = timeit(lambda: fused._c2_wrw(fine, coarse))
    tot["down"] += td; tot["up"] += tu; tot["wrw"] += tw
    line = f"{name:22s} C={C:3d} M={M:3d} {H:3d}^2  down {td*1e3:7.1f} us {fl/td/1e9:6.1f} TF/s | up {tu*1e3:7.1f} us {fl/tu/1e9:6.1f} | wrw {tw*1e3:7.1f} us {fl/tw/1e9:6.1f}"
    if miopen:
        tm = timeit(lambda: F.conv2d(fine, w, None, 2, 1))
        tmt = timeit(lambda: F.conv_transpose2d(coarse, w, None, 2, 1))
        line += f" | torch conv {tm*1e3:7.1f} us {fl/tm/1e9:6.1f}  convT {tmt*1e3:7.1f} us {fl/tmt/1e9:6.1f}"
    print(line, flush=True)
print("sum over layers: down %.3f ms  up %.3f ms  wrw %.3f ms" % (tot["down"], tot["up"], tot["wrw"]))
