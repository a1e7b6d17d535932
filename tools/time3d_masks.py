#!/usr/bin/env python3
"""3-D fused loss + gradient with the Dirichlet masks held as uint8 or as the reference's fp32 images (node-owner kernel), and
through the generic form of the T16 kernel (dn_config_set("Q1_3D_T16")) that fp32 masks used before."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import DiffNet3DFEM, _lib   # noqa: E402

dev = torch.device("cuda:0")
for n, B in [(256, 1), (128, 4)]:
    m = DiffNet3DFEM(None, domain_size=n, nsd=3).to(dev)
    shape = (B, 1, n, n, n)
    g = torch.Generator().manual_seed(1)
    u, nu, f = (torch.rand(shape, generator=g).to(dev) for _ in range(3))
    nu += 0.5
    bc = torch.zeros(shape, dtype=torch.uint8, device=dev)
    bc[..., 0] = 1; bc[..., -1] = 1; bc[..., 0, :] = 1; bc[..., -1, :] = 1; bc[:, :, 0] = 1; bc[:, :, -1] = 1
    src = (torch.rand(shape, generator=g) < 0.02).to(torch.uint8).to(dev)
    cases = {"u8 x1": [(bc, 0.0)], "f32 x1": [(bc.float(), 0.0)], "u8 x2": [(src, 1.0), (bc, 0.0)], "f32 x2": [(src.float(), 1.0), (bc.float(), 0.0)]}
    ref = {}
    for t16 in ("", "1"):
        _lib.config_set("Q1_3D_T16", t16)
        for name, d in cases.items():
            fn = lambda: m.energy_loss_and_grad(u, nu, f, dirichlet=d, c=1.0)
            for _ in range(4):
                fn()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
            for a, b in evs:
                a.record(); out = fn(); b.record()
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
            key = name.split()[1]
            same = ""
            if t16 == "":
                if key in ref:
                    same = f"  grad == u8 form: {bool(torch.equal(out[1], ref[key][1]))}  loss rel diff {abs(float(out[0]) - float(ref[key][0])) / abs(float(ref[key][0])):.1e}"
                else:
                    ref[key] = out
            print(f"{n}^3 B={B} {'T16 kernel ' if t16 else 'node-owner '} masks {name:7s}: median {ts[15]:.1f} us  min {ts[0]:.1f}{same}", flush=True)
    _lib.config_set("Q1_3D_T16", "")
