import torch, sys, os
sys.path.insert(0, ".")
from diffnet_amd import DiffNet2DFEM
dev = torch.device("cuda:0")
n = int(os.environ.get("N", "512")); B = int(os.environ.get("B", "4"))
m = DiffNet2DFEM(None, domain_size=n, ngp_1d=3).to(dev)
g = torch.Generator().manual_seed(1)
shape = (B, 1, n, n)
u = torch.rand(shape, generator=g).to(dev); nu = (0.5 + torch.rand(shape, generator=g)).to(dev); f = torch.rand(shape, generator=g).to(dev)
bc = torch.zeros(shape, dtype=torch.uint8); bc[..., 0] = 1; bc[..., -1] = 1; bc[:, :, 0] = 1; bc[:, :, -1] = 1; bc = bc.to(dev)
mode = os.environ.get("MODE", "bc")
d = [(bc, 0.0)] if mode == "bc" else []
res = []
for i in range(4):
    l, gr = m.energy_loss_and_grad(u, nu, f, dirichlet=d, c=0.5)
    res.append((float(l), float(gr.double().abs().sum())))
print(os.environ.get("DN_PLAN2D"), os.environ.get("DN_NO_DMA"), mode, n, B, res)
