import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from diffnet_amd import BoxFaces, DiffNet2DFEM, ops
dev = torch.device("cuda:0")
m = DiffNet2DFEM(None, domain_size=512, ngp_1d=3).to(dev)
shape = (64, 1, 512, 512)
g = torch.Generator().manual_seed(1)
u, nu, f = (torch.rand(shape, generator=g).to(dev) for _ in range(3))
nu += 0.5
scale = 1.0 / (64 * m.geom.nelem_total)
mk = lambda: ops.PoissonPlan(m.geom, u, nu, f, None, [(BoxFaces(), 0.0)], alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=scale, want_out=True, want_sums=True, loss_scale=scale)
for name, stream in (("default stream", torch.cuda.current_stream()), ("side stream", torch.cuda.Stream())):
    with torch.cuda.stream(stream):
        plans = [mk() for _ in range(5)]
        ref_grad, ref_sums, ref_loss = [t.clone() for t in plans[0].launch()]
        torch.cuda.synchronize()
        bad = 0
        t0 = time.perf_counter()
        for it in range(2000):
            plans[it % 5].launch()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2000 * 1e6
        for pl in plans:
            gr, sm, ls = pl.result
            bad += int(not torch.equal(ls, ref_loss)) + int(not torch.equal(sm, ref_sums)) + int(not torch.equal(gr, ref_grad))
        # every launch's loss, checked: copy it out on the same stream right after the launch
        keep = torch.empty(400, device=dev)
        for it in range(400):
            keep[it].copy_(plans[it % 5].launch()[2])
        torch.cuda.synchronize()
        bad2 = int((keep != ref_loss).sum())
        print(f"{name}: {dt:.2f} us per launch over 2000 back-to-back launches; mismatching final results {bad}; of 400 per-launch losses wrong: {bad2}", flush=True)
