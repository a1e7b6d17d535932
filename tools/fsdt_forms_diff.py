import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import DiffNet2DFEM, _lib
from diffnet_amd.elasticity import fsdt_residuals
dev = torch.device("cuda:0")
deg, ngp, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
m = DiffNet2DFEM(None, domain_size=n, fem_basis_deg=deg, ngp_1d=ngp).to(dev)
B = 3
shape = (B, 1, n, n)
g = torch.Generator().manual_seed(1)
fields = [torch.rand(shape, generator=g).to(dev) for _ in range(3)]
bc = torch.zeros(shape); bc[..., 0] = 1; bc[..., -1] = 1; bc[..., 0, :] = 1; bc[..., -1, :] = 1
bc[1, 0, n // 2, 2:5] = 1.0
bc = bc.to(dev)
wbc = torch.rand(shape, generator=g).to(dev)
for bcf in (False, True):
    kw = dict(w_bc=wbc if bcf else 0.75, phi_x_bc=0.25, phi_y_bc=-0.5, E=2.0, v=0.3, h=0.2, K_s=5.0 / 6.0, q=1.5, hx=m.h, hy=0.7 * m.h)
    _lib.config_set("FSDT_FORM", "elem")
    Re = [r.cpu().numpy() for r in fsdt_residuals(m, *fields, bc, **kw)]
    _lib.config_set("FSDT_FORM", "")
    Rs = [r.cpu().numpy() for r in fsdt_residuals(m, *fields, bc, **kw)]
    for k in range(3):
        d = np.abs(Re[k] - Rs[k])
        bad = np.argwhere(d > 1e-4 * np.abs(Re[k]).max())
        print(f"bcf={bcf} field {k}: max diff {d.max():.3e} scale {np.abs(Re[k]).max():.3e} bad {len(bad)}")
        if len(bad):
            ys = sorted(set(int(b[2]) for b in bad)); xs = sorted(set(int(b[3]) for b in bad))
            print("   rows", ys[:40], "cols", xs[:40], "samples", sorted(set(int(b[0]) for b in bad)))
            for b in bad[:6]:
                print("   ", tuple(int(i) for i in b), Re[k][tuple(b)], Rs[k][tuple(b)])
