#!/usr/bin/env python3
"""Benchmark of the FEM hot path: fused energy loss + gradient (one pass) on BASELINE.json configs[1]'s mesh.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--settle S]

`value` / `ms_per_step`: W untimed warm-up steps, then exactly K timed steps.  The same W + K steps are run a second time after S = 400
further untimed steps and reported as `steady_state` (for 1.3-10 ms after load onset the part runs every kernel 10-25 % slower, and
W = 5, K = 20 is 1.4 ms of load: a job of more than a few hundred steps is in the second regime, and so are the roofline launches).

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (the driver's form:
RANK / LOCAL_RANK / WORLD_SIZE come from the environment), or run plainly -- then this process, which has not touched the GPU
yet, starts that very launcher as a child process and exits with its code.

One "step" = one evaluation of the Poisson energy loss AND its gradient wrt u (forward + backward of the
reference's loss body IBN_2D.py:116-134 / e2_cib_neumann-style nu field) over one batch of synthetic nodal
fields already resident in HBM: 2-D Q1, 512 x 512 nodes, 3 x 3 Gauss points, B samples per GPU (weak
scaling: each rank owns its own batch shard, like the reference's DDP; the only exchange is the all-reduce
of the scalar losses, one collective per four steps).  metric = elements * gauss_pts / s summed over ranks.

Every launch of the run -- warm-up, timed steps, roofline -- goes through NROT DIFFERENT batches in rotation (own input, mask and
output arrays each): a training loop never re-reads the same u, and one batch's 268 MB of arrays is about the size of the 256 MB
Infinity Cache, so re-evaluating ONE batch is cache-assisted (round 2: 47 vs 57 us).  `value`, `ms_per_step` and `roofline` are
therefore one regime; the one-batch kernel time is a side field, and so are the same launches over 16 batches (nothing of a batch
left in the cache at its next turn; with 4, part of u still is) and over 3 streams (`roofline.deeper_rotation`).  The Dirichlet mask is held as a general
per-sample mask array (one bit per node, diffnet_amd.PackedMask); the geometry-derived form is a side field.

Prints ONE JSON line (rank 0).  Extra objects: `roofline` (time of the dominant kernel between timing-only HIP events over the
rotation vs the HBM peak of MI355X_MICROARCH.md, plus `stream_ceiling`: a plain 3-read / 1-write streaming kernel over the same
arrays), `cpu_baseline` (the CPU oracle = port of the reference formulation, timed on this box's host cores on a bounded sample
of the same workload), `configs` (device time of every BASELINE.json config) and `slab_3d` (configs[3] over the same ranks).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
ALG_BYTES_PER_NODE = 16        # read u, nu, f + write grad_u, fp32
PIPE = 4                       # loss all-reduces in flight (N > 1)


def timed_pairs(launch, n, settle=0):
    """Duration of n launches, each between its own pair of HIP events recorded on the current (launch) stream, after `settle` untimed
    launches issued AFTER the events have been created: creating 2 n events takes the host 10-20 ms, about what 400 queued launches take the
    GPU -- with the untimed launches issued before the events were created (rounds 2-4 until the last session) the queue could run dry just
    before the first timed launch, which then started from an idle GPU and put launches 30-100 into the load-onset transient (1.3-10 ms
    after onset, tools/ramp2d.py): 54 -> 60-65 us per launch on the same kernel, depending on which side of that race a build fell
    (gpurun_out/ab_cur3.json / ab_prev3.json of the round: launch-order series of both).  The events are
    created with hipEventDisableSystemFence -- HIP's flag for events "only being used to measure timing", which skips the system-scope
    cache write-back / invalidate a default event performs when it is recorded: with default events (torch.cuda.Event) that fence is
    charged to the launch between them, 1.8 us here (tools/event_cost.py, profiles/r2_event_cost.txt: 49.4 us per launch between default
    events, 47.6 between these, 47.1-48.1 per launch over 100 launches between ONE pair).  Returns (list of ms, kind of event used);
    falls back to torch.cuda.Event when the HIP runtime torch loaded cannot be reached through ctypes."""
    import ctypes
    try:
        path = next(ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64.so" in ln)       # the instance torch runs on
        hip = ctypes.CDLL(path)
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        evs = []
        for _ in range(2 * n):
            e = ctypes.c_void_p()
            if hip.hipEventCreateWithFlags(ctypes.byref(e), ctypes.c_uint(0x20000000)) != 0:           # hipEventDisableSystemFence
                raise OSError("hipEventCreateWithFlags")
            evs.append(e)
    except (StopIteration, OSError, AttributeError):
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for _ in range(settle):
            launch()
        for a, b in pairs:
            a.record()
            launch()
            b.record()
        torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in pairs], "torch.cuda.Event (hipEventDefault)"
    for _ in range(settle):
        launch()
    for i in range(n):
        hip.hipEventRecord(evs[2 * i], stream)
        launch()
        hip.hipEventRecord(evs[2 * i + 1], stream)
    torch.cuda.synchronize()
    out = []
    for i in range(n):
        ms = ctypes.c_float()
        if hip.hipEventElapsedTime(ctypes.byref(ms), evs[2 * i], evs[2 * i + 1]) != 0:
            raise RuntimeError("hipEventElapsedTime failed")
        out.append(ms.value)
    for e in evs:
        hip.hipEventDestroy(e)
    return out, "hipEventDisableSystemFence"


def make_inputs(shape, dev, seed):
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(shape, generator=g)
    nu = 0.5 + torch.rand(shape, generator=g)
    f = torch.rand(shape, generator=g)
    bc = torch.zeros(shape, dtype=torch.uint8)
    for d in range(2, len(shape)):
        idx = [slice(None)] * len(shape)
        idx[d] = 0
        bc[tuple(idx)] = 1
        idx[d] = -1
        bc[tuple(idx)] = 1
    return [t.to(dev) if dev is not None else t for t in (u, nu, f, bc)]


def host_cores():
    """Cores this process may actually use: the cgroup CPU quota when there is one, else the affinity mask
    (os.cpu_count() reports the whole host, and oversubscribing torch's intra-op pool makes the CPU leg crawl)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_baseline(kw, c, budget_s=15.0):
    """Time the oracle (torch-CPU port of the reference op sequence: per-GP conv + cat + elementwise + autograd
    backward) on a bounded sample: batch 2 of the same mesh, as many iterations as fit the budget."""
    from oracle.fem_oracle import Oracle
    ncores = min(host_cores(), int(os.environ.get("DN_CPU_THREADS", "32")))
    torch.set_num_threads(ncores)
    o = Oracle(**kw)
    Bs = 2
    shape = (Bs, 1, *[kw["domain_size"]] * kw.get("nsd", 2))
    u, nu, f, bc = make_inputs(shape, None, 42)
    bcf = bc.float()
    units = Bs * int(torch.tensor(o.spec.nel).prod()) * o.spec.ngp_total

    def step():
        ur = u.clone().requires_grad_(True)
        loss = o.energy(ur, nu, f, dirichlet=[(bcf, 0.0)], c=c)
        loss.backward()
        return float(loss)

    step()
    t0 = time.perf_counter()
    it = 0
    while True:
        step()
        it += 1
        el = time.perf_counter() - t0
        if el > budget_s or it >= 400:
            break
    return {"value": units * it / el, "unit": "elements*gauss_pts/s", "cores": ncores, "kind": "port",
            "sample": f"oracle/fem_oracle.py energy fwd+bwd, batch {Bs} of the same mesh, {it} iters in {el:.1f}s, "
                      f"torch {torch.__version__} CPU, {ncores} threads (os.cpu_count()={os.cpu_count()})"}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU) through torch.distributed.run
    as a CHILD process.  Nothing in this process has initialised the GPU at this point (argparse + `import torch` only), and
    the launcher is started as a subprocess, never exec'ed."""
    import socket
    import subprocess
    ngpu = torch.cuda.device_count()                 # does not initialise the device
    if os.environ.get("DN_DIST_BACKEND", "nccl") == "nccl" and ngpu < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {ngpu} GPU(s) visible; the RCCL run needs one GPU per rank "
                         "(DN_DIST_BACKEND=gloo rehearses the multi-rank logic on fewer GPUs)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def slab_leg(args, rank, world, dev, dist, n, B, ngp, steps, warmup):
    """BASELINE configs[3]: ONE n^3 Q1 mesh, 2x2x2 points, domain-decomposed into z-slabs over the ranks (strong scaling).
    Returns the result dict on rank 0 (None elsewhere)."""
    from diffnet_amd.slab import SlabPoisson
    sp = SlabPoisson(3, (n, n, n), (1.0, 1.0, 1.0), rank, world, ngp_1d=ngp, device=dev)
    nzl = sp.dec.n1 - sp.dec.n0 + 1
    shape = (B, 1, nzl, n, n)
    g = torch.Generator().manual_seed(42 + rank)
    u, nu, f = (torch.rand(shape, generator=g).to(dev) for _ in range(3))
    nu += 0.5
    bc = torch.zeros(shape, dtype=torch.uint8, device=dev)
    bc[..., 0] = 1; bc[..., -1] = 1; bc[..., 0, :] = 1; bc[..., -1, :] = 1
    if rank == 0:
        bc[:, :, 0] = 1
    if rank == world - 1:
        bc[:, :, -1] = 1

    def step():
        return sp.energy_loss_and_grad(u, nu, f, dirichlet=[(bc, 0.0)], c=1.0)

    for _ in range(warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    tmax = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    if rank != 0:
        return None
    units = B * sp.dec.nel_global * ngp ** 3
    alg = ALG_BYTES_PER_NODE * B * n ** 3
    return {"metric": "elements*gauss_pts/sec (FEM loss+grad)", "value": units * steps / dt, "unit": "elements*gauss_pts/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "dtype": "f32", "data": "synthetic",
            "hbm_frac_of_all_gpus": alg / (dt / steps) / 1e9 / (HBM_PEAK_GBS * world),
            "config": {"workload": f"3-D Poisson energy loss + gradient, Q1, ONE {n}^3 mesh x batch {B}, {ngp}^3 Gauss pts, z-slabs over "
                                   f"{world} rank(s): 4-byte loss all-reduce + interface-layer exchange per step, overlapped with the slab "
                                   "kernel (BASELINE.json configs[3])",
                       "nodes": [n, n, n], "parallelism": f"slab x{world}"}}


def slab_main(args, rank, world, dev, dist):
    out = slab_leg(args, rank, world, dev, dist, args.size, args.batch, args.ngp, args.steps, args.warmup + args.settle)
    if rank == 0:
        out["vs_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------------------------
# device time of every BASELINE.json config (appended to the line as "configs")
# ------------------------------------------------------------------------------------------------------------------------------
_SLEEP_CYC_PER_US = [None]


def device_us(fn, n=40, warm=3, settle_s=0.04):
    """Device time per call of `fn` in steady state.  40 ms of load first (between 1.3 and 10 ms after the GPU leaves idle every kernel
    runs 10-25 % slower, tools/ramp2d.py); then n calls enqueued behind a blocker kernel that outlasts the host's enqueue work, between
    one pair of events -- or, for calls whose host side is slower than the device (several launches + torch ops per call), the n calls
    captured into ONE HIP graph whose replay is timed.  Returns (device_us, host_us, method): every row says which of the three regimes it
    was timed in ("back-to-back" / "hip-graph" / "behind-blocker")."""
    if _SLEEP_CYC_PER_US[0] is None:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1000); torch.cuda.synchronize()
        a.record(); torch.cuda._sleep(10_000_000); b.record(); torch.cuda.synchronize()
        _SLEEP_CYC_PER_US[0] = 10_000_000 / (a.elapsed_time(b) * 1e3)
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < settle_s:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    host_us = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # device-bound calls (the host enqueues faster than the device executes): the queue never runs dry, so one pair of events around n calls issued
    # back to back IS the device time -- and the GPU stays in the loaded state the settle phase put it in (behind a blocker kernel it drops out of
    # it: the low-power spin of the blocker is followed by the 1.3-10 ms transient, which costs the power-limited 3-D kernel ~7 %)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    direct_us = a.elapsed_time(b) / n * 1e3
    if direct_us > 3.0 * host_us:
        return direct_us, host_us, "back-to-back"
    if host_us > 60.0:
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
            return a.elapsed_time(b) / n * 1e3, host_us, "hip-graph"
        except Exception:                      # not capturable: fall through to the blocker
            torch.cuda.synchronize()
    torch.cuda._sleep(int(_SLEEP_CYC_PER_US[0] * host_us * n * 3.0) + 1000)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3, host_us, "behind-blocker"


MFMA_F32_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: dense fp32 matrix-core peak (v_mfma_f32_*: the precision the reference trains in)


def conv_step_flops(net, x):
    """FLOPs of one training step's convolutions (forward + input gradient + weight gradient = 3 x forward), counted by forward hooks on every
    Conv / ConvTranspose module of `net` during one forward of x: 2 * Cout * (Cin / groups) * prod(kernel) * B * prod(output positions) per
    Conv, the same with the INPUT positions per ConvTranspose.  Functional convolutions (the fused output blocks) are not modules and are not
    counted: the figure is a lower bound."""
    from torch import nn
    total = [0]

    def hook(mod, inp, out):
        k = 1
        for v in mod.kernel_size:
            k *= v
        pos = (inp[0] if isinstance(mod, (nn.ConvTranspose2d, nn.ConvTranspose3d)) else out)
        npos = pos.shape[0]
        for v in pos.shape[2:]:
            npos *= v
        total[0] += 2 * mod.out_channels * (mod.in_channels // mod.groups) * k * npos

    hs = [m_.register_forward_hook(hook) for m_ in net.modules() if isinstance(m_, (nn.Conv2d, nn.Conv3d, nn.ConvTranspose2d, nn.ConvTranspose3d))]
    with torch.no_grad():
        net(x)
    for h in hs:
        h.remove()
    return 3 * total[0]


def config_rows(dev, budget_s=150.0):
    """One row per BASELINE.json config (and the batch sizes SURVEY 8(d) names): us per evaluation of the loss + gradient in steady
    state, fraction of the HBM peak on the config's algorithmic bytes, work units per second.  Inputs larger
    than 64 MB rotate over several sets of arrays (no Infinity-Cache help).  Public API calls (the cached prepared launches of
    diffnet_amd.fem / elasticity), not hand-prepared plans."""
    from diffnet_amd import DiffNet2DFEM, DiffNet3DFEM
    t_start = time.perf_counter()
    rows = []

    def add(name, us, host, alg_bytes, units, note=None, method="wall-clock"):
        r = {"name": name, "us_per_eval": round(us, 2), "host_us_per_call": round(host, 1), "frac": round(alg_bytes / us / 1e3 / HBM_PEAK_GBS, 4),
             "units_per_s": units / us * 1e6, "timing": method}
        if note:
            r["note"] = note
        rows.append(r)

    def poisson(name, nsd, n, ngp, B, c, nsets=1, load=False, box=False, fold=False):
        """load: the forcing of every sample as its assembled load vector (diffnet_amd.LoadVector, assembled once outside the timed calls);
        box: the Dirichlet condition as faces of the domain box (BoxFaces: no array); fold: prepared launches whose final reduction is done by
        the next launch of the rotation (PoissonPlan(pipelined_sums=True)).  Without them: the public one-shot call on the reference's inputs."""
        if time.perf_counter() - t_start > budget_s:
            return
        from diffnet_amd import BoxFaces, LoadVector
        from diffnet_amd import ops as _ops
        cls = DiffNet3DFEM if nsd == 3 else DiffNet2DFEM
        m = cls(None, domain_size=n, ngp_1d=ngp, nsd=nsd).to(dev)
        shape = (B, 1, *m.geom.node_shape)
        sets = [make_inputs(shape, dev, 7 + k) for k in range(nsets)]
        if load:
            sets = [(u, nu, LoadVector.assemble(m.geom, f), bc) for (u, nu, f, bc) in sets]
        outs = [torch.empty(shape, device=dev) for _ in range(nsets)]
        turn = [0]
        cond = (lambda bc: [(BoxFaces("all"), 0.0)]) if box else (lambda bc: [(bc, 0.0)])
        if fold:
            scale = 1.0 / (B * m.geom.nelem_total)
            plans = [_ops.PoissonPlan(m.geom, u, nu, f, None, cond(bc), alpha=2.0 * c, beta=1.0, c=c, wscale=1.0, out_scale=scale, want_out=True,
                                      want_sums=True, loss_scale=scale, out=outs[k], pipelined_sums=True) for k, (u, nu, f, bc) in enumerate(sets)]
            for k in range(nsets):
                plans[k].fold(plans[k - 1])

            def fn():
                k = turn[0]
                turn[0] = (k + 1) % nsets
                return plans[k].launch()
        else:
            def fn():
                k = turn[0]
                turn[0] = (k + 1) % nsets
                u, nu, f, bc = sets[k]
                return m.energy_loss_and_grad(u, nu, f, dirichlet=cond(bc), c=c, out=outs[k])

        us, host, how = device_us(fn)
        notes = ([f"{nsets} sets of arrays in rotation"] if nsets > 1 else []) + (["forcing as assembled load vector"] if load else []) + \
                (["Dirichlet condition as box faces (no array)"] if box else []) + (["prepared launches, final reduction folded into the next launch"] if fold else [])
        add(name, us, host, ALG_BYTES_PER_NODE * B * m.geom.nnode_total, B * m.geom.nelem_total * m.geom.ngp_total, "; ".join(notes) or None, how)

    def fsdt(name, n, B):
        if time.perf_counter() - t_start > budget_s:
            return
        from diffnet_amd.elasticity import fsdt_loss_and_grad, fsdt_total_loss
        m = DiffNet2DFEM(None, domain_size=n, fem_basis_deg=2, ngp_1d=3).to(dev)
        shape = (B, 1, n, n)
        g = torch.Generator().manual_seed(2)
        fields = [torch.rand(shape, generator=g).to(dev).requires_grad_(True) for _ in range(3)]
        bc = torch.zeros(shape, device=dev)
        bc[..., 0] = 1; bc[..., -1] = 1; bc[..., 0, :] = 1; bc[..., -1, :] = 1

        def fn():
            return fsdt_loss_and_grad(m, *fields, bc)

        us, host, how = device_us(fn)
        add(name, us, host, 68 * B * n * n, B * m.geom.nelem_total * m.geom.ngp_total, "fsdt_loss_and_grad: residual norms + gradient, two launches, no autograd graph", how)

        def fn_autograd():
            torch.autograd.grad(fsdt_total_loss(m, *fields, bc), fields)

        us, host, how = device_us(fn_autograd)
        add(name + " [autograd: fsdt_total_loss + backward]", us, host, 68 * B * n * n, B * m.geom.nelem_total * m.geom.ngp_total,
            "the same two launches behind torch.autograd (one Function node): host_us_per_call is the eager wall time per step", how)

    def dropin_ibn2d(name, n, B):
        """The loss() body of IBN/poisson-2d/parametric/IBN_2D.py:116-134 as the reference wrote it (two torch.where lines, five
        gauss_pt_evaluation* calls, the broadcast multiply / sum / mean), running on the drop-in operators (examples/ibn_2d_parametric.py,
        Poisson(dropin=True)), forward + backward wrt u: what an UNMODIFIED caller pays.  The fused spelling of the same loss is the
        `cfg2 ... B=16` row above."""
        if time.perf_counter() - t_start > budget_s:
            return
        from examples.ibn_2d_parametric import Poisson
        mod = Poisson(None, dropin=True, domain_size=n, ngp_1d=3).to(dev)
        g = torch.Generator().manual_seed(5)
        u = torch.rand((B, 1, n, n), generator=g).to(dev).requires_grad_(True)
        f = torch.rand((B, 1, n, n), generator=g).to(dev)
        sink = torch.zeros((B, 1, n, n), device=dev)
        sink[..., 0] = 1; sink[..., -1] = 1; sink[..., 0, :] = 1; sink[..., -1, :] = 1
        src = ((torch.rand((B, 1, n, n), generator=g) < 0.05).float().to(dev)) * (1 - sink)

        def fn():
            (gu,) = torch.autograd.grad(mod.loss(u, src, f, sink), u)
            return gu

        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = 10
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / k * 1e6
        add(name, us, us, ALG_BYTES_PER_NODE * B * n * n, B * mod.geom.nelem_total * mod.geom.ngp_total,
            "5 dn_gauss_pt_eval_fwd + 3 adjoint launches + the caller's own torch elementwise ops on (B, 9, 511, 511) tensors; wall clock per loss + backward; "
            "frac counts the fused path's 16 B/node")

    def unet(name, n, B):
        if time.perf_counter() - t_start > budget_s:
            return
        from diffnet_amd.networks.unets import UNet
        torch.manual_seed(0)
        net = UNet(2, 1).to(dev)
        fem = DiffNet2DFEM(net, domain_size=n, ngp_1d=3).to(dev)
        opt = torch.optim.Adam(net.parameters(), lr=1e-4)
        u0, nu, f, bc = make_inputs((B, 1, n, n), dev, 11)
        x = torch.cat([nu, bc.float()], 1)

        def fn():
            opt.zero_grad(set_to_none=True)
            loss = fem.energy_loss(net(x), nu, f, dirichlet=[(bc, 0.0)], c=1.0)
            loss.backward()
            opt.step()

        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = 10
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / k * 1e6
        flops = conv_step_flops(net, x)
        add(name, us, us, ALG_BYTES_PER_NODE * B * n * n, B * fem.geom.nelem_total * fem.geom.ngp_total,
            "whole training step (UNet forward + FEM loss + backward + Adam), wall clock; frac counts the FEM bytes only")
        rows[-1]["network"] = {"bound": "mfma", "achieved": round(flops / us / 1e6, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(flops / us / 1e6 / MFMA_F32_PEAK_TFLOPS, 4), "conv_gflop_per_step": round(flops / 1e9, 1),
                               "note": "module convolutions, forward + both gradients, over the WHOLE step time (norms, activations, FEM loss, Adam included)"}

    def gen3d(name, n, B):
        if time.perf_counter() - t_start > budget_s:
            return
        from diffnet_amd.networks.wgan3d import GoodGenerator
        torch.manual_seed(0)
        net = GoodGenerator(1, 1).to(dev)
        fem = DiffNet3DFEM(net, domain_size=n, ngp_1d=2, nsd=3).to(dev)
        opt = torch.optim.Adam(net.parameters(), lr=1e-4)
        nu = torch.rand(B, 1, n, n, n, device=dev) + 0.5
        bc = torch.zeros(B, 1, n, n, n, device=dev, dtype=torch.uint8)
        bc[..., 0] = 1; bc[..., -1] = 1

        def fn():
            opt.zero_grad(set_to_none=True)
            loss = fem.energy_loss(net(nu), nu, None, dirichlet=[(bc, 0.0)], c=0.5)
            loss.backward()
            opt.step()

        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = 10
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / k * 1e6
        flops = conv_step_flops(net, nu)
        add(name, us, us, ALG_BYTES_PER_NODE * B * n ** 3, B * fem.geom.nelem_total * fem.geom.ngp_total,
            "whole training step (GoodGenerator forward + FEM loss + backward + Adam: IBN_3D.py:114-136, wgan3d.py:23-98), wall clock; frac counts the FEM bytes only")
        rows[-1]["network"] = {"bound": "mfma", "achieved": round(flops / us / 1e6, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(flops / us / 1e6 / MFMA_F32_PEAK_TFLOPS, 4), "conv_gflop_per_step": round(flops / 1e9, 1),
                               "note": "module convolutions, forward + both gradients, over the WHOLE step time"}

    poisson("cfg1 2-D 64^2 Q1 2x2 B=1 energy c=1/2", 2, 64, 2, 1, 0.5)
    poisson("cfg2 2-D 512^2 Q1 3x3 B=1", 2, 512, 3, 1, 1.0)
    poisson("cfg2 2-D 512^2 Q1 3x3 B=16", 2, 512, 3, 16, 1.0, nsets=4)
    poisson("cfg2 2-D 512^2 Q1 3x3 B=64 (uint8 mask images, public API)", 2, 512, 3, 64, 1.0, nsets=4)
    poisson("cfg3 3-D 128^3 Q1 2x2x2 B=1 energy c=1/2", 3, 128, 2, 1, 0.5)
    poisson("cfg3 3-D 128^3 B=1, load vector + box faces, folded sums", 3, 128, 2, 1, 0.5, nsets=4, load=True, box=True, fold=True)
    poisson("cfg4 3-D 256^3 Q1 2x2x2 B=1 (whole mesh on one GPU)", 3, 256, 2, 1, 1.0, nsets=4)
    poisson("cfg4 3-D 256^3 B=1, forcing as assembled load vector (uint8 mask image)", 3, 256, 2, 1, 1.0, nsets=4, load=True)
    poisson("cfg4 3-D 256^3 B=1, load vector + box faces, folded sums", 3, 256, 2, 1, 1.0, nsets=4, load=True, box=True, fold=True)
    dropin_ibn2d("cfg2 UNCHANGED CALLER: the loss() body of IBN_2D.py:116-134 on the drop-in operators + backward, 512^2 B=16", 512, 16)
    fsdt("cfg5 FSDT plate 1025^2 nodes (512^2 Q2 elements) 3x3 B=1", 1025, 1)
    fsdt("cfg5 FSDT plate 1025^2 Q2 3x3 B=8", 1025, 8)
    unet("cfg2 UNet(2->1) + FEM loss training step, 512^2 B=16", 512, 16)
    gen3d("cfg3 GoodGenerator(1->1) + FEM loss training step, 128^3 B=1", 128, 1)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5, help="untimed steps directly before the timed ones")
    ap.add_argument("--settle", type=int, default=400,
                    help="untimed steps before the SECOND run of the warm-up + timed steps, reported as `steady_state` (about 22 ms of load: between about 1.3 and "
                         "10 ms after the GPU leaves idle the launch runs 10-25 %% slower -- power-management transient, tools/ramp2d.py, "
                         "profiles/r2_ramp2d.txt).  `value` is the W + K steps run first, from idle; this second run is reported as `steady_state`; "
                         "--settle 0 makes the two the same thing")
    ap.add_argument("--batch", type=int, default=64, help="samples per GPU")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--ngp", type=int, default=3)
    ap.add_argument("--nsd", type=int, default=2)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-config leg (\"configs\")")
    ap.add_argument("--sums", default="fold", choices=["fold", "kernel", "async"],
                    help="where the loss of a step is formed from the launch's per-workgroup partial sums.  fold (default, round 4): by the first "
                         "workgroup of the NEXT step's launch (dn_poisson_args.fold_prev; the last step of a timed region by one small kernel inside "
                         "the region) -- the ~3 us serial tail at the end of every launch leaves the critical path, no event, no side stream: 56.3 -> "
                         "53.7 us per step (profiles/r4_fold_sums.txt).  kernel: inside the launch (its last workgroup adds the partials up: rounds "
                         "1-3).  async: one-workgroup kernel on a side stream under the next launch (measured slower: 60.5 us, profiles/r3_async_sums.txt)")
    ap.add_argument("--async-sums", action="store_true", help="same as --sums async")
    ap.add_argument("--bc", default="auto", choices=["auto", "bits", "u8", "f32", "box"],
                    help="how the Dirichlet condition is held (auto: one bit per node for 2-D, uint8 image for 3-D: general mask arrays)")
    ap.add_argument("--slab", action="store_true",
                    help="strong-scaling variant (not the default metric run): ONE 3-D mesh of --size^3 nodes cut into z-slabs over "
                         "the ranks (diffnet_amd/slab.py): per step one 4-byte all-reduce + one node-layer exchange per interior face")
    ap.add_argument("--slab-size", type=int, default=256, help="mesh of the slab leg appended to the default run (0 = skip)")
    ap.add_argument("--slab-steps", type=int, default=100)
    ap.add_argument("--slab-warmup", type=int, default=300, help="untimed steps of the slab leg (past the load-onset transient: profiles/r2_ramp3d.txt)")
    ap.add_argument("--slab-batch", type=int, default=1, help="samples of the slab leg's mesh (BASELINE configs[3] is parametric: the reference trains it with batch 8)")
    ap.add_argument("--slab-timeout", type=float, default=180.0, help="watchdog of the slab leg, seconds")
    ap.add_argument("--no-slab-b8", action="store_true", help="skip the batch-8 repetition of the slab leg (`slab_3d_b8`)")
    args = ap.parse_args()
    if args.async_sums:
        args.sums = "async"
    args.sync_sums = args.sums != "async"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)                                      # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist, backend = None, None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DN_DIST_BACKEND", "nccl")      # "gloo" only to rehearse the N > 1 logic on one GPU
        local_rank %= max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    if args.slab:
        return slab_main(args, rank, world, dev, dist)

    from diffnet_amd import BoxFaces, DiffNet2DFEM, DiffNet3DFEM, PackedMask, _lib
    from diffnet_amd import ops as _ops
    kw = dict(domain_size=args.size, ngp_1d=args.ngp, nsd=args.nsd)
    cls = DiffNet3DFEM if args.nsd == 3 else DiffNet2DFEM
    m = cls(None, **kw).to(dev)
    B = args.batch
    shape = (B, 1, *m.geom.node_shape)
    c = 1.0
    units_per_step = B * m.geom.nelem_total * m.geom.ngp_total
    scale0 = 1.0 / (B * m.geom.nelem_total)
    # NROT batches in rotation, each with its own u, nu, f, mask and output arrays.  With N > 1: two GROUPS of PIPE batches whose losses lie next
    # to each other in one buffer -- the path's only exchange step, the all-reduce of the 4-byte loss, is issued ONCE PER GROUP over the group's PIPE
    # losses (asynchronously; waited one group later, before the group's slots are written again; all drained before a timed region closes, so
    # every step's loss IS reduced inside it).  One collective per step would put an RCCL workgroup onto the chip during every launch, and
    # the launch is exactly one round of resident waves: whichever of its workgroups finds its slot taken starts when the collective ends.
    grouped = dist is not None and args.sync_sums
    NROT = 4 if dist is None else (2 * PIPE if grouped else PIPE + 1)
    sets = [make_inputs(shape, dev, 1000 * k + 42 + rank) for k in range(NROT)]
    loss_buf = torch.zeros(NROT, dtype=torch.float32, device=dev) if grouped else None
    # The Dirichlet condition of BASELINE.md section 3 ("mask on all boundary faces") as the dataset keeps it in HBM:
    #   bits  one bit per node, per sample (diffnet_amd.PackedMask: packed once when the dataset is placed on the device; ANY mask) [default]
    #   u8    one byte per node     f32  the reference's fp32 image     box  no array: derived from the geometry (this workload only)
    bc_form = args.bc if args.bc != "auto" else ("bits" if args.nsd == 2 else "u8")
    forms = {"u8": lambda bc: [(bc, 0.0)], "f32": lambda bc: [(bc.float(), 0.0)], "bits": lambda bc: [(PackedMask.pack(bc), 0.0)],
             "box": lambda bc: [(BoxFaces("all"), 0.0)]}

    def make_plans(form, mode=None):
        # the prepared form of m.energy_loss_and_grad(u, nu, f, dirichlet, c) (diffnet_amd.ops.PoissonPlan: argument structs, outputs and
        # workspace set up once, one ctypes call per launch).  mode (--sums): "fold" = every launch leaves its per-workgroup partial sums and
        # forms the scalars of the launch BEFORE it in the rotation (its first workgroup, before its own march); "kernel" = in-kernel final
        # reduction; "async" = one-workgroup kernel on a side stream
        main = mode is None
        mode = args.sums if mode is None else mode
        pls = [_ops.PoissonPlan(m.geom, u, nu, f, None, forms[form](bc), alpha=2.0 * c, beta=1.0, c=c, wscale=1.0, out_scale=scale0,
                                want_out=True, want_sums=True, loss_scale=scale0, async_sums=mode == "async", pipelined_sums=mode == "fold",
                                loss_out=loss_buf[k:k + 1] if (loss_buf is not None and form == bc_form and main) else None)
               for k, (u, nu, f, bc) in enumerate(sets)]
        if mode == "fold":
            for k in range(len(pls)):
                pls[k].fold(pls[k - 1])
        return pls

    rot = make_plans(bc_form)
    turn = [0]
    pending = []
    last = [None]

    def launch_rot():
        k = turn[0]
        turn[0] = (k + 1) % NROT
        last[0] = rot[k]
        return rot[k].launch()

    first_unreduced = [0]          # grouped: the first slot of the current group whose loss no collective has taken yet
    fold = args.sums == "fold"     # the loss of evaluation k is formed by launch k + 1 (slot k of loss_buf is written one launch late)

    def close_last():
        """fold: the scalars of the last launched evaluation are still per-workgroup partial sums -- one small kernel forms them (called
        inside every timed region, so that every step's loss is formed inside it)"""
        if fold and last[0] is not None:
            last[0].finish_sums()

    def reduce_slots(a, b):
        if b <= a:                                         # [a, NROT): the range ends at the wrap of the rotation
            b = NROT
        view = loss_buf[a:b]
        if backend == "nccl":                              # mean over ranks inside the collective: no extra launch
            work = dist.all_reduce(view, op=dist.ReduceOp.AVG, async_op=True)
        else:
            view.div_(world)
            work = dist.all_reduce(view, async_op=True)
        pending.append((work, view))

    def step():
        if grouped:
            k = turn[0]
            # the collective that read this group's slots last must be done before the first of them is written again: by this launch
            # (in-kernel sums: launch k writes slot k) or by the next one (fold: launch k writes slot k - 1)
            if k % PIPE == (1 if fold else 0):
                while len(pending) > 1:
                    pending.pop(0)[0].wait()
            grad, _, loss = launch_rot()
            if fold:
                if k % PIPE == 0 and first_unreduced[0] != k:      # this launch has just formed the last loss of the group before it
                    reduce_slots(first_unreduced[0], k)
                    first_unreduced[0] = k
            elif (k + 1) % PIPE == 0:
                reduce_slots(first_unreduced[0], k + 1)
                first_unreduced[0] = (k + 1) % NROT
            return loss, grad
        grad, _, loss = launch_rot()
        if dist is not None and last[0].async_sums:
            # the loss lives on the side stream: its all-reduce is issued there too, so that the launch stream never waits for it
            with torch.cuda.stream(last[0].sums_stream):
                if backend == "nccl":
                    work = dist.all_reduce(loss, op=dist.ReduceOp.AVG, async_op=True)
                else:
                    loss.div_(world)
                    work = dist.all_reduce(loss, async_op=True)
            pending.append((work, loss))
            if len(pending) > PIPE:
                pending.pop(0)[0].wait()
        elif dist is not None:
            # the path's only exchange step: all-reduce of the 4-byte loss (RCCL).  Issued asynchronously so that the
            # next evaluation's kernel does not queue behind the collective; waited PIPE steps later (a small-message
            # all-reduce over xGMI is latency-bound at tens of microseconds, comparable to one step) and drained
            # before the timed region closes, i.e. every step's loss IS reduced inside the timed region.
            if backend == "nccl":                          # mean over ranks inside the collective: no extra launch per step
                work = dist.all_reduce(loss, op=dist.ReduceOp.AVG, async_op=True)
            else:
                loss.div_(world)
                work = dist.all_reduce(loss, async_op=True)
            pending.append((work, loss))
            if len(pending) > PIPE:
                pending.pop(0)[0].wait()
        return loss, grad

    def drain():
        close_last()
        if grouped and first_unreduced[0] != turn[0]:      # a group that is not full yet: its losses so far
            reduce_slots(first_unreduced[0], turn[0])
            first_unreduced[0] = turn[0]
        while pending:
            pending.pop(0)[0].wait()

    def timed_steps():
        """W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides; MAX over ranks."""
        for _ in range(args.warmup):
            step()
        drain()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        # one pair of HIP events on the launch stream around the K timed launches (two records in total, none between the launches)
        region = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        t0 = time.perf_counter()
        region[0].record()
        for _ in range(args.steps):
            step()
        close_last()                                       # (fold: the last step's loss; a no-op otherwise)
        region[1].record()
        drain()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        tmax = torch.tensor([dt_], device=dev, dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax), region[0].elapsed_time(region[1]) / args.steps

    # The GPU leaves idle when the first launch arrives, and between 1.3 and 10 ms after that every kernel runs 10-25 % slower (power-management
    # ramp, tools/ramp2d.py): the driver's W = 5, K = 20 is 1.4 ms of load, i.e. entirely inside that transient.  The W + K steps are therefore
    # run twice.  `value` / `ms_per_step` are the FIRST run -- exactly the contract: W untimed warm-up steps, then K timed steps, nothing before
    # them (round 4; rounds 2-3 reported the second run as `value` and this one as `cold_start`).  The second run, after --settle untimed steps
    # (default 400: 22 ms of the same load), is the steady state a job of more than a few hundred steps is in and the regime of the roofline
    # launches below: reported as `steady_state`.
    dt, region_ms = timed_steps()
    for _ in range(args.settle):
        step()
    drain()
    steady_dt, steady_timed_ms = timed_steps()

    # dominant-kernel time: timing-only HIP events on the launch stream around each dn_poisson_apply (ONE kernel: the fused Poisson
    # kernel, whose last workgroup also does the fixed-order final reduction), K launches over the rotation, each between its own pair
    # (timed_pairs above).  A short run (the driver's is 5 + 20 steps, 1.3 ms of load) would put these launches into the power-management
    # transient 1.3-10 ms after load onset (tools/ramp2d.py), which says nothing about the kernel; SETTLE untimed launches first carry
    # the GPU past it.  The transient is reported as roofline.kernel_avg_ms_first_launches.
    K, SETTLE = 100, 400
    first_ms, _ = timed_pairs(launch_rot, 20)
    kern_ms, event_kind = timed_pairs(launch_rot, K, settle=SETTLE)
    kern_series = [round(x * 1e3, 1) for x in kern_ms[:48]]          # in launch order (the rotation has NROT batches): a per-batch pattern shows here
    kern_ms.sort()
    rot_region = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    rot_region[0].record()
    for _ in range(2 * K):
        launch_rot()
    rot_region[1].record()
    torch.cuda.synchronize()
    steady_region_ms = rot_region[0].elapsed_time(rot_region[1]) / (2 * K)
    # side field: the same rotation with the loss formed inside the launch (in-kernel final reduction)
    sync_ms = None
    if args.sums != "kernel" and rank == 0:
        pls = make_plans(bc_form, "kernel")
        t = [0]

        def go_sync():
            pls[t[0]].launch()
            t[0] = (t[0] + 1) % NROT

        sync_ms = sorted(timed_pairs(go_sync, K, settle=200)[0])
        del pls
    # side fields: the same launch re-evaluating ONE batch (Infinity-Cache assisted: what rounds 1 and 2 reported as the step), and the
    # rotation with the mask held in the other formats (median of 60 after 200 untimed launches)
    same_ms = sorted(timed_pairs(rot[0].launch, K, settle=50)[0])
    bc_forms_us = {}
    if args.nsd == 2 and rank == 0:
        for name in forms:
            pls = rot if name == bc_form else make_plans(name, args.sums)
            t = [0]

            def go():
                pls[t[0]].launch()
                t[0] = (t[0] + 1) % NROT

            # 12 ms under load first: a few warm-up launches would leave the timed ones in the load-onset transient
            bc_forms_us[name] = round(sorted(timed_pairs(go, 60, settle=200)[0])[30] * 1e3, 2)
            del pls
    # stream ceiling: a plain streaming kernel (dn_probe_stream: out = a * b + c, 16-byte vectors) over the SAME arrays and rotation --
    # three arrays read once, one written once; best of its forms / cache policies
    stream = None
    if rank == 0:
        import ctypes
        fn = _lib.lib().dn_probe_stream
        nfl = sets[0][0].numel()
        best = None
        for mode in (0, 1, 4, 5, 8, 9, 7):
            t = [0]

            def go():
                u, nu, f, _ = sets[t[0]]
                rc = fn(u.data_ptr(), nu.data_ptr(), f.data_ptr(), rot[t[0]].result[0].data_ptr(), nfl, mode,
                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                if rc:
                    raise RuntimeError(f"dn_probe_stream rc={rc}")
                t[0] = (t[0] + 1) % NROT

            ms = sorted(timed_pairs(go, 60, settle=40)[0])
            avg = sum(ms) / len(ms)
            if best is None or avg < best[0]:
                best = (avg, mode, ms[len(ms) // 2])
        for _ in range(8):
            launch_rot()          # the probe wrote into the gradient arrays: leave them as the operator writes them
        torch.cuda.synchronize()
        stream = best
    # side fields: (a) the same launches over 16 batches in rotation on one stream (4.3 GB of arrays: nothing of a batch survives in the 256 MB
    # Infinity Cache until its next turn, and the TLB reach is exceeded too -- with 4 batches part of u still does, because nu and f are read
    # with non-temporal loads and no longer displace it); (b) 12 batches on 3 streams, 4 per stream: independent evaluations issued round-robin,
    # so that the ramp-up of one launch runs under the tail of another (last strips, final reduction, launch gap)
    deeper = None
    if rank == 0 and world == 1 and args.nsd == 2 and not args.no_configs:
        extra = [make_inputs(shape, dev, 1000 * k + 42) for k in range(NROT, 16)]
        all_sets = sets + extra
        mk = lambda st: _ops.PoissonPlan(m.geom, st[0], st[1], st[2], None, forms[bc_form](st[3]), alpha=2.0 * c, beta=1.0, c=c, wscale=1.0,
                                         out_scale=scale0, want_out=True, want_sums=True, loss_scale=scale0, async_sums=False)
        pl16 = [_ops.PoissonPlan(m.geom, st[0], st[1], st[2], None, forms[bc_form](st[3]), alpha=2.0 * c, beta=1.0, c=c, wscale=1.0, out_scale=scale0,
                                 want_out=True, want_sums=True, loss_scale=scale0, pipelined_sums=fold) for st in all_sets]
        if fold:
            for k in range(16):
                pl16[k].fold(pl16[k - 1])
        t = [0]

        def go16():
            pl16[t[0]].launch()
            t[0] = (t[0] + 1) % 16

        ms16 = sorted(timed_pairs(go16, K, settle=200)[0])
        NS, PER = 3, 4
        streams = [torch.cuda.Stream() for _ in range(NS)]
        pls = []
        for k in range(NS * PER):
            with torch.cuda.stream(streams[k % NS]):
                pls.append(mk(all_sets[k]))
        torch.cuda.synchronize()

        def burst(n):
            for i in range(n):
                k = i % (NS * PER)
                with torch.cuda.stream(streams[k % NS]):
                    pls[k].launch()

        burst(240)
        torch.cuda.synchronize()
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        for st_ in streams:
            st_.wait_stream(torch.cuda.current_stream())
        burst(480)
        for st_ in streams:
            torch.cuda.current_stream().wait_stream(st_)
        ev[1].record()
        torch.cuda.synchronize()
        ms3 = ev[0].elapsed_time(ev[1]) / 480
        deeper = {"rotation_16_kernel_avg_ms": sum(ms16) / len(ms16), "rotation_16_kernel_median_ms": ms16[len(ms16) // 2],
                  "rotation_16_frac": ALG_BYTES_PER_NODE * B * m.geom.nnode_total / (sum(ms16) / len(ms16) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                  "three_streams_12_batches_ms_per_evaluation": ms3, "three_streams_value": units_per_step / (ms3 * 1e-3),
                  "note": "rotation_16_*: the roofline launches over 16 instead of %d batches on one stream; three_streams_*: 12 batches, 4 per stream, "
                          "evaluations issued round-robin on 3 streams (whole-region time / evaluations)" % NROT}
        del pl16, pls, extra, all_sets
    kern_avg_ms = sum(kern_ms) / len(kern_ms)
    kern_med_ms = kern_ms[len(kern_ms) // 2]
    alg_bytes = ALG_BYTES_PER_NODE * B * m.geom.nnode_total
    achieved = alg_bytes / (kern_avg_ms * 1e-3) / 1e9

    traffic, traffic_src = None, None
    try:   # HBM bytes per launch from the committed rocprofv3 PMC summary of this exact workload (profiles/)
        prof = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json")))
        key = f"{args.nsd}d_n{args.size}_g{args.ngp}_b{B}" + ("" if bc_form == "u8" else "_" + bc_form)
        if key in prof:
            traffic, traffic_src = prof[key]["hbm_bytes_per_launch"], prof[key]["source"]
    except Exception:
        pass

    out = None
    if rank == 0:
        value = units_per_step * world * args.steps / dt
        mask_note = {'bits': 'held as one bit per node and sample (a general mask array)', 'u8': 'held as a uint8 image', 'f32': 'held as an fp32 image',
                     'box': 'derived from the geometry (no array)'}[bc_form]
        out = {
            "metric": "elements*gauss_pts/sec (FEM loss+grad)", "value": value, "unit": "elements*gauss_pts/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle": args.settle, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.nsd}-D Poisson energy loss + gradient wrt u, Q1, {args.size}^{args.nsd} nodes, "
                                   f"{args.ngp}^{args.nsd} Gauss pts, batch {B}/GPU, nu+f nodal fields, Dirichlet mask on all boundary faces "
                                   + mask_note + f", fused single pass (BASELINE.json configs[1] mesh); every step evaluates the next of {NROT} "
                                   "different batches in rotation (nu, f and the mask stream from HBM with non-temporal loads; of the 4 x 67 MB of u a part may survive in the 256 MB "
                                   "Infinity Cache until the batch's next turn: roofline.deeper_rotation has the 16-batch figure)"
                                   + ("; 2-D Q1 element evaluated in closed form: the rule's sums as polynomials of its moments, same value "
                                      "as the per-point sum (dn_config_set(\"Q1_RULE_KERNEL\") runs the per-point kernel)" if args.nsd == 2 else ""),
                       "batch_per_gpu": B, "nodes": list(m.geom.node_shape), "parallelism": f"batch-sharded x{world}",
                       "batches_in_rotation": NROT, "mask_format": bc_form},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "poisson fused kernel (one launch per dn_poisson_apply)", "kernel_avg_ms": kern_avg_ms,
                         "sums_mode": args.sums,
                         "sums": {"fold": "the loss of step k is formed from launch k's per-workgroup partial sums by the FIRST workgroup of launch k + 1, before its "
                                          "own march (dn_poisson_args.fold_prev): the ~3 us serial final reduction leaves the end of every launch; the last step of "
                                          "a timed region is closed by dn_poisson_finish_sums inside the region.  kernel_avg_ms is this launch (it includes the "
                                          "folding of the launch before it); kernel_avg_ms_with_in_kernel_sums is the round-1..3 form",
                                  "async": "the loss of a step is formed from the launch's per-workgroup partial sums by a one-workgroup kernel on a side stream, "
                                           "under the next step's launch (dn_poisson_finish_sums); value / ms_per_step include every step's reduction",
                                  "kernel": "the loss is formed inside the launch (its last workgroup adds up the partial sums)"}[args.sums],
                         "kernel_avg_ms_with_in_kernel_sums": None if sync_ms is None else sum(sync_ms) / len(sync_ms),
                         "frac_with_in_kernel_sums": None if sync_ms is None else alg_bytes / (sum(sync_ms) / len(sync_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "kernel_median_ms": kern_med_ms, "kernel_min_ms": kern_ms[0], "kernel_max_ms": kern_ms[-1], "kernel_us_first_48_in_launch_order": kern_series,
                         "frac_at_median": alg_bytes / (kern_med_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": alg_bytes,
                         "batches": "%d different batches in rotation, as in the timed steps" % NROT,
                         "steady_ms_per_launch_back_to_back": steady_region_ms,
                         "value_steady": units_per_step / (steady_region_ms * 1e-3),
                         "stream_ceiling": None if stream is None else {
                             "what": "dn_probe_stream: plain 3-read / 1-write fp32 streaming kernel over the same arrays and rotation",
                             "avg_ms": stream[0], "median_ms": stream[2], "mode": stream[1], "GBps": alg_bytes / (stream[0] * 1e-3) / 1e9,
                             "frac_of_peak": alg_bytes / (stream[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_over_stream": kern_avg_ms / stream[0]},
                         "one_batch_kernel_avg_ms": sum(same_ms) / len(same_ms), "one_batch_kernel_median_ms": same_ms[len(same_ms) // 2],
                         "frac_one_batch": alg_bytes / (sum(same_ms) / len(same_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "cache_note": "one_batch_*: the same launch re-evaluating ONE batch finds part of its 268 MB of arrays in the 256 MB "
                                       "Infinity Cache; not a regime a training loop is in, reported for comparison with rounds 1-2 only",
                         "kernel_avg_ms_first_launches": sum(first_ms) / len(first_ms),
                         "events": event_kind + " pair around each of %d launches, after %d untimed ones" % (K, SETTLE),
                         "timed_region_ms_per_launch": region_ms,
                         "frac_over_timed_region": alg_bytes / (region_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "rotation_kernel_median_us_by_mask_format": bc_forms_us, "deeper_rotation": deeper},
            "steady_state": {"value": units_per_step * world * args.steps / steady_dt, "ms_per_step": steady_dt / args.steps * 1e3,
                             "timed_region_ms_per_launch": steady_timed_ms, "untimed_steps_before": 2 * args.warmup + args.steps + args.settle,
                             "note": "the same W warm-up + K timed steps repeated after `settle` further untimed steps: `value` (W + K steps from an idle GPU: with "
                                     "the driver's W = 5, K = 20 all of it falls into the 1.3-10 ms power-management transient after load onset, tools/ramp2d.py) "
                                     "says what a 25-step job sees, this what a job of more than a few hundred steps sees.  Rounds 2-3 reported this one as "
                                     "`value` and the other as `cold_start`"},
        }
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(kw, c)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        if not args.no_configs and world == 1 and args.nsd == 2 and args.size == 512:
            del rot, sets
            torch.cuda.empty_cache()
            try:
                out["configs"] = config_rows(dev)
            except Exception as e:                      # noqa: BLE001 -- the headline above is complete; say what failed
                out["configs"] = {"error": f"{type(e).__name__}: {e}"[:400]}

    # strong-scaling leg of BASELINE configs[3] in the same run (all ranks take part): one 256^3 mesh cut into z-slabs.  The headline
    # measurement above is complete at this point; the leg runs under a watchdog so that a failure or a hang in its point-to-point
    # exchange (first exercised over RCCL on the driver's multi-GPU node) still leaves the ONE JSON line, with the reason in
    # "slab_3d.error".  Exit status (round 4): with ONE rank the headline is complete and the status stays 0; with WORLD_SIZE > 1 a failed or hung
    # slab leg ends the process with status 3 AFTER the line has been printed -- under torchrun a dead RCCL point-to-point exchange must not read as
    # success (status 1: the line could not be printed).  `emitted` makes sure exactly one line is printed, whoever gets there first.
    if args.slab_size and args.nsd == 2 and args.size == 512:
        import threading
        emit_lock, emitted = threading.Lock(), [False]

        def emit():
            with emit_lock:
                if emitted[0]:
                    return True
                emitted[0] = True
                try:
                    if rank == 0:
                        print(json.dumps(out), flush=True)
                    return True
                except Exception:
                    return False

        fail_status = 3 if world > 1 else 0

        def give_up():
            if rank == 0:
                out["slab_3d"] = {"error": f"slab leg did not finish within {args.slab_timeout} s (n_gpus={world})"}
            os._exit(fail_status if emit() else 1)

        dog = threading.Timer(args.slab_timeout, give_up)
        dog.daemon = True
        dog.start()
        try:
            slab = slab_leg(args, rank, world, dev, dist, args.slab_size, args.slab_batch, 2, args.slab_steps, args.slab_warmup)
            if rank == 0:
                out["slab_3d"] = slab
            # the same mesh with the batch the reference trains IBN_3D with (IBN_3D.py:167-179): per-rank kernels grow with B, the exchange does not
            if args.slab_batch != 8 and not args.no_slab_b8:
                slab8 = slab_leg(args, rank, world, dev, dist, args.slab_size, 8, 2, max(10, args.slab_steps // 4), max(20, args.slab_warmup // 4))
                if rank == 0:
                    out["slab_3d_b8"] = slab8
        except Exception as e:                       # noqa: BLE001 -- reported, not hidden: the line says what failed
            dog.cancel()
            if rank == 0:
                out.setdefault("slab_3d", {"error": f"{type(e).__name__}: {e}"[:500]})
                if "error" not in out["slab_3d"]:
                    out["slab_3d_b8"] = {"error": f"{type(e).__name__}: {e}"[:500]}
            os._exit(fail_status if emit() else 1)   # peers may be stuck in a collective: no orderly teardown (their watchdogs end them)
        dog.cancel()
        emit()
    elif rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
