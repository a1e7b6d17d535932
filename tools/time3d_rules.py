import os, sys, torch
sys.path.insert(0, os.getcwd())
from diffnet_amd import DiffNet3DFEM
dev = torch.device("cuda:0")
for n, ngp, B in [(129, 3, 2), (96, 3, 2), (65, 4, 2), (128, 3, 1)]:
    m = DiffNet3DFEM(None, domain_size=n, nsd=3, ngp_1d=ngp).to(dev)
    shape = (B, 1, n, n, n)
    g = torch.Generator().manual_seed(1)
    u, nu, f = (torch.rand(shape, generator=g).to(dev) for _ in range(3))
    nu += 0.5
    bc = torch.zeros(shape, dtype=torch.uint8, device=dev)
    bc[..., 0] = 1; bc[..., -1] = 1; bc[..., 0, :] = 1; bc[..., -1, :] = 1; bc[:, :, 0] = 1; bc[:, :, -1] = 1
    fn = lambda: m.energy_loss_and_grad(u, nu, f, dirichlet=[(bc, 0.0)], c=1.0)
    for _ in range(5): fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    print(f"3-D {n}^3 ngp={ngp} B={B}: median {ts[20]:.1f} us  min {ts[0]:.1f}", flush=True)
