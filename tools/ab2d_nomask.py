#!/usr/bin/env python3
"""2-D bench workload with / without the uint8 Dirichlet mask stream (how much of the kernel time is the byte stream?)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import DiffNet2DFEM
dev = torch.device("cuda:0")
m = DiffNet2DFEM(None, domain_size=512, ngp_1d=3).to(dev)
shape = (64, 1, 512, 512)
g = torch.Generator().manual_seed(1)
u, nu, f = (torch.rand(shape, generator=g).to(dev) for _ in range(3))
bc = torch.zeros(shape, dtype=torch.uint8, device=dev); bc[..., 0] = 1; bc[..., -1] = 1; bc[..., 0, :] = 1; bc[..., -1, :] = 1
out = torch.empty_like(u)
for name, d in (("u8 mask", [(bc, 0.0)]), ("no mask", []), ("fp32 mask", [(bc.float(), 0.0)])):
    fn = lambda: m.energy_loss_and_grad(u, nu, f, dirichlet=d, c=1.0, out=out)
    for _ in range(10): fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    print(f"{name}: median {ts[50]:.1f} us min {ts[0]:.1f}", flush=True)
