#!/usr/bin/env python3
"""Hot-path time of every BASELINE.json config on one GPU (kernel-level, HIP events; inputs resident in HBM).
Prints one line per config: avg us per evaluation (loss + gradient), algorithmic GB/s, fraction of the 8 TB/s HBM peak."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffnet_amd import DiffNet2DFEM, DiffNet3DFEM, ops
from diffnet_amd.elasticity import fsdt_loss, fsdt_residuals_composed

dev = torch.device("cuda", 0)


_SLEEP_CYC_PER_US = None


def timed(fn, n=40, warm=3):
    """Device time per call: the n calls are enqueued behind a blocker kernel (torch.cuda._sleep) that outlasts the host's
    enqueue work, so the GPU runs them back to back and HIP events see no host gaps.  Returns (device_us, host_us)."""
    import time
    global _SLEEP_CYC_PER_US
    if _SLEEP_CYC_PER_US is None:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1000); torch.cuda.synchronize()
        a.record(); torch.cuda._sleep(10_000_000); b.record(); torch.cuda.synchronize()
        _SLEEP_CYC_PER_US = 10_000_000 / (a.elapsed_time(b) * 1e3)
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    # 40 ms of load first: between 1.3 and 10 ms after the GPU leaves idle every kernel runs 10-25 % slower (tools/ramp2d.py)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.04:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    host_us = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if host_us > 60.0:
        # host-bound evaluations (several launches and torch ops per call, e.g. the FSDT loss + backward): the blocker approach reads
        # host gaps as device time when the enqueue work outlasts it.  Capture the n calls into ONE HIP graph and time its replay.
        try:
            graph, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn()
            torch.cuda.current_stream().wait_stream(side)
            with torch.cuda.graph(graph):
                for _ in range(n):
                    fn()
            graph.replay()
            torch.cuda.synchronize()
            a.record()
            graph.replay()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / n * 1e3, host_us
        except Exception:                      # not capturable: fall through to the blocker
            torch.cuda.synchronize()
    torch.cuda._sleep(int(_SLEEP_CYC_PER_US * host_us * n * 3.0) + 1000)     # the blocker must outlast the enqueue work (x1.5 was not enough on loaded hosts: 1.6 ms outliers)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3, host_us


def poisson(nsd, n, ngp, B, label):
    cls = DiffNet3DFEM if nsd == 3 else DiffNet2DFEM
    m = cls(None, domain_size=n, ngp_1d=ngp, nsd=nsd).to(dev)
    shape = (B, 1, *m.geom.node_shape)
    g = torch.Generator().manual_seed(1)
    u, nu, f = (torch.rand(shape, generator=g).to(dev) for _ in range(3))
    bc = torch.zeros(shape, dtype=torch.uint8, device=dev)
    bc[..., 0] = 1; bc[..., -1] = 1
    nu = nu + 0.5
    us, host = timed(lambda: m.energy_loss_and_grad(u, nu, f, dirichlet=[(bc, 0.0)], c=1.0))
    byt = 16 * B * m.geom.nnode_total
    return dict(config=label, us=us, host_us=host, alg_GBs=byt / us / 1e3, frac=byt / us / 1e3 / 8000.0,
                Gunits_s=B * m.geom.nelem_total * m.geom.ngp_total / us / 1e3)


def fsdt(n, deg, ngp, B, label, composed=False):
    m = DiffNet2DFEM(None, domain_size=n, fem_basis_deg=deg, ngp_1d=ngp).to(dev)
    shape = (B, 1, n, n)
    g = torch.Generator().manual_seed(2)
    fields = [torch.rand(shape, generator=g).to(dev).requires_grad_(True) for _ in range(3)]
    bc = torch.zeros(shape, device=dev)
    bc[..., 0] = 1; bc[..., -1] = 1; bc[..., 0, :] = 1; bc[..., -1, :] = 1

    def step():
        if composed:
            Rs = fsdt_residuals_composed(m, *fields, bc)
            loss = sum(torch.norm(R) for R in Rs)
        else:
            loss = sum(fsdt_loss(m, *fields, bc))
        torch.autograd.grad(loss, fields)

    us, host = timed(step, n=10 if composed else 40)
    byt = (3 * 4 + 4 + 3 * 4) * B * n * n * 2          # fwd + bwd launch: 3 fields + mask in, 3 fields out
    return dict(config=label, us=us, host_us=host, alg_GBs=byt / us / 1e3, frac=byt / us / 1e3 / 8000.0,
                Gunits_s=B * m.geom.nelem_total * m.geom.ngp_total / us / 1e3)


rows = [
    poisson(2, 64, 2, 1, "cfg0 2-D 64^2 Q1 2x2 B=1"),
    poisson(2, 512, 3, 64, "cfg1 2-D 512^2 Q1 3x3 B=64 (bench.py)"),
    poisson(2, 512, 3, 1, "cfg1 2-D 512^2 Q1 3x3 B=1"),
    poisson(3, 128, 2, 1, "cfg2 3-D 128^3 Q1 2x2x2 B=1"),
    poisson(3, 128, 2, 4, "cfg2 3-D 128^3 Q1 2x2x2 B=4"),
    poisson(3, 256, 2, 1, "cfg3 3-D 256^3 Q1 2x2x2 B=1 (one GPU holds the whole mesh)"),
    poisson(3, 129, 3, 2, "     3-D 129^3 Q1 3x3x3 B=2"),
    fsdt(1025, 2, 3, 1, "cfg4 FSDT 1025^2 nodes = 512^2 Q2 elements, 3x3, B=1 fused (fwd+bwd launches)"),
    fsdt(1025, 2, 3, 8, "cfg4 FSDT 1025^2 Q2 3x3 B=8 fused"),
    fsdt(513, 2, 3, 1, "cfg4 FSDT 513^2 Q2 3x3 B=1 fused (small variant)"),
    fsdt(513, 2, 3, 1, "cfg4 FSDT 513^2 Q2 3x3 B=1 composed from operators", composed=True),
    fsdt(1025, 2, 3, 1, "cfg4 FSDT 1025^2 Q2 3x3 B=1 composed from operators", composed=True),
    fsdt(512, 1, 2, 8, "     FSDT 512^2 Q1 2x2 B=8 fused (the reference script's element)"),
]
for r in rows:
    print(f"{r['config']:62s} {r['us']:9.1f} us (host {r['host_us']:6.1f})  {r['alg_GBs']:8.1f} GB/s  frac {r['frac']:.3f}  {r['Gunits_s']:8.2f} G elem*gp/s", flush=True)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
