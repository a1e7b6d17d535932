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
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    pl = ops.PoissonPlan(m.geom, u, nu, f, None, [(BoxFaces(), 0.0)], alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=scale, want_out=True, want_sums=True, loss_scale=scale)
    for _ in range(5): pl.launch()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for _ in range(50): pl.launch()
    for _ in range(2): graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4): graph.replay()
    torch.cuda.synchronize()
    print(f"graph of 50 launches: {(time.perf_counter() - t0) / 200 * 1e6:.2f} us per launch", flush=True)
    t0 = time.perf_counter()
    for _ in range(200): pl.launch()
    torch.cuda.synchronize()
    print(f"eager prepared launches: {(time.perf_counter() - t0) / 200 * 1e6:.2f} us per launch", flush=True)
