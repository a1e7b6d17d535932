#!/usr/bin/env python3
"""One conv2d_k4s2 contraction repeated (for rocprofv3): python tools/run_conv.py down|up|wrw C M H [B] [reps]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd.networks import fused   # noqa: E402
kind, C, M, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 16
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 10
dev = torch.device("cuda:0")
fine = torch.randn(B, C, 2 * H, 2 * H, device=dev)
coarse = torch.randn(B, M, H, H, device=dev)
w = torch.randn(M, C, 4, 4, device=dev) * 0.05
fn = {"down": lambda: fused._c2_down(fine, w), "up": lambda: fused._c2_up(coarse, w), "wrw": lambda: fused._c2_wrw(fine, coarse)}[kind]
for _ in range(reps):
    fn()
torch.cuda.synchronize()
print("done")
