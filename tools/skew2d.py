#!/usr/bin/env python3
"""Does the relative placement of the four streamed arrays (u, nu, f, grad) in HBM matter?  The default allocations are 64 MiB
apart (identical channel / bank bits for the same node); this carves them from one pool with a byte skew between arrays."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import DiffNet2DFEM   # noqa: E402

dev = torch.device("cuda:0")
B, n = 64, 512
m = DiffNet2DFEM(None, domain_size=n, ngp_1d=3).to(dev)
shape = (B, 1, n, n)
numel = B * n * n
g = torch.Generator().manual_seed(1)
src = [torch.rand(shape, generator=g) for _ in range(3)]
src[1] += 0.5
bc = torch.zeros(shape, dtype=torch.uint8)
bc[..., 0] = 1; bc[..., -1] = 1; bc[..., 0, :] = 1; bc[..., -1, :] = 1
bc = bc.to(dev)
from diffnet_amd import BoxFaces   # noqa: E402
skews = [0, 256, 1024, 4096 + 256, 16384 + 1024, 65536 + 4096 + 256, (1 << 20) + 65536 + 4096 + 256, (2 << 20) + 4096, (8 << 20) + 65536]
sets = {}
for skew in skews:
    pool = torch.empty(4 * numel + 4 * (skew // 4 + 64) * 4, dtype=torch.float32, device=dev)
    ts = []
    for k in range(4):
        off = k * (numel + skew // 4)
        ts.append(pool[off:off + numel].view(shape))
    for t, s_ in zip(ts[:3], src):
        t.copy_(s_)
    sets[skew] = ts
sets["torch allocations"] = [t.to(dev) for t in src] + [torch.empty(shape, device=dev)]
res = {k: [] for k in sets}
for form, d in (("box", [(BoxFaces(), 0.0)]), ("u8", [(bc, 0.0)])):
    for rnd in range(3):
        for key, (u, nu, f, out) in sets.items():
            fn = lambda: m.energy_loss_and_grad(u, nu, f, dirichlet=d, c=1.0, out=out)
            for _ in range(5):
                fn()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
            for a, b in evs:
                a.record(); fn(); b.record()
            torch.cuda.synchronize()
            tms = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
            res[key].append((form, tms[25], tms[0]))
    for key in sets:
        rows = [r for r in res[key] if r[0] == form]
        med = sorted(r[1] for r in rows)[1]
        print(f"{form:3s} skew {str(key):>18s}  median-of-rounds {med:.1f} us  rounds {[round(r[1], 1) for r in rows]}  min {min(r[2] for r in rows):.1f}", flush=True)
