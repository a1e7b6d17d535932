#!/usr/bin/env python3
"""A/B timing of kernel variants selected through dn_config_set, one process, interleaved launches, HIP events.
usage: python tools/ab_kernels.py 3d|2d|fsdt [KEY=VALUE ...]   (each KEY=VALUE is one variant; "" = default)"""
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import DiffNet2DFEM, DiffNet3DFEM, _lib   # noqa: E402


def time_variants(fn, variants, reps=60, rounds=3):
    res = {v: [] for v in variants}
    for _ in range(rounds):
        for v in variants:
            if v:
                k, val = v.split("=", 1)
                _lib.config_set(k, val)
            for _ in range(5):
                fn()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for a, b in evs:
                a.record(); fn(); b.record()
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
            res[v].append((ts[len(ts) // 2], ts[0], sum(ts) / len(ts)))
            if v:
                _lib.config_set(k, "")
    return res


def main():
    what = sys.argv[1]
    variants = [""] + sys.argv[2:]
    dev = torch.device("cuda:0")
    cases = {"3d": [(dict(domain_size=128, nsd=3), 1), (dict(domain_size=128, nsd=3), 4), (dict(domain_size=256, nsd=3), 1),
                    (dict(domain_size=129, nsd=3), 2), (dict(domain_size=96, nsd=3, ngp_1d=3), 2)],
             "2d": [(dict(domain_size=512, ngp_1d=3), 64), (dict(domain_size=512, ngp_1d=3), 16), (dict(domain_size=1024, ngp_1d=2), 16)]}[what]
    for kw, B in cases:
        cls = DiffNet3DFEM if kw.get("nsd", 2) == 3 else DiffNet2DFEM
        m = cls(None, **kw).to(dev)
        shape = (B, 1, *m.geom.node_shape)
        g = torch.Generator().manual_seed(1)
        u, nu, f = (torch.rand(shape, generator=g).to(dev) for _ in range(3))
        nu += 0.5
        bc = torch.zeros(shape, dtype=torch.uint8, device=dev)
        for d in range(2, len(shape)):
            idx = [slice(None)] * len(shape); idx[d] = 0; bc[tuple(idx)] = 1; idx[d] = -1; bc[tuple(idx)] = 1
        fn = lambda: m.energy_loss_and_grad(u, nu, f, dirichlet=[(bc, 0.0)], c=1.0)
        res = time_variants(fn, variants)
        nbytes = 16 * B * m.geom.nnode_total
        for v, rows in res.items():
            med = sorted(r[0] for r in rows)[len(rows) // 2]
            print(f"{kw} B={B} [{v or 'default'}] median {med:.1f} us  min {min(r[1] for r in rows):.1f}  mean {sum(r[2] for r in rows)/len(rows):.1f}"
                  f"  -> {nbytes / med / 1e3:.0f} GB/s = {nbytes / med / 1e3 / 8000:.3f} of HBM peak", flush=True)


if __name__ == "__main__":
    main()
