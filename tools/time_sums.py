import os, sys, torch
sys.path.insert(0, os.getcwd())
from diffnet_amd import BoxFaces, DiffNet2DFEM, DiffNet3DFEM, ops
dev = torch.device("cuda:0")
def run(m, shape, d, tag):
    g = torch.Generator().manual_seed(1)
    u, nu, f = (torch.rand(shape, generator=g).to(dev) for _ in range(3))
    nu += 0.5
    scale = 1.0 / (shape[0] * m.geom.nelem_total)
    plans = {"with sums": ops.PoissonPlan(m.geom, u, nu, f, None, d, alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=scale, want_out=True, want_sums=True, loss_scale=scale),
             "no sums": ops.PoissonPlan(m.geom, u, nu, f, None, d, alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=scale, want_out=True, want_sums=False)}
    for rnd in range(3):
        for name, pl in plans.items():
            for _ in range(5): pl.launch()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
            for a, b in evs:
                a.record(); pl.launch(); b.record()
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
            if rnd == 2: print(f"{tag} {name:10s} median {ts[25]:.1f} us  min {ts[0]:.1f}", flush=True)
m = DiffNet2DFEM(None, domain_size=512, ngp_1d=3).to(dev)
run(m, (64, 1, 512, 512), [(BoxFaces(), 0.0)], "2-D 512^2 B=64 box")
m3 = DiffNet3DFEM(None, domain_size=256, nsd=3).to(dev)
bc = torch.zeros((1, 1, 256, 256, 256), dtype=torch.uint8, device=dev); bc[..., 0] = 1; bc[..., -1] = 1
run(m3, (1, 1, 256, 256, 256), [(bc, 0.0)], "3-D 256^3")
m3b = DiffNet3DFEM(None, domain_size=128, nsd=3).to(dev)
bcb = torch.zeros((1, 1, 128, 128, 128), dtype=torch.uint8, device=dev); bcb[..., 0] = 1
run(m3b, (1, 1, 128, 128, 128), [(bcb, 0.0)], "3-D 128^3 B=1")
m2 = DiffNet2DFEM(None, domain_size=64, ngp_1d=2).to(dev)
run(m2, (1, 1, 64, 64), [(BoxFaces(), 0.0)], "2-D 64^2 B=1")
