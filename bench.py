#!/usr/bin/env python3
"""Benchmark of the FEM hot path: fused energy loss + gradient (one pass) on BASELINE.json configs[1]'s mesh.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (the driver's form:
RANK / LOCAL_RANK / WORLD_SIZE come from the environment), or run plainly -- then this process, which has not touched the GPU
yet, starts that very launcher as a child process and exits with its code.

One "step" = one evaluation of the Poisson energy loss AND its gradient wrt u (forward + backward of the
reference's loss body IBN_2D.py:116-134 / e2_cib_neumann-style nu field) over one batch of synthetic nodal
fields already resident in HBM: 2-D Q1, 512 x 512 nodes, 3 x 3 Gauss points, B samples per GPU (weak
scaling: each rank owns its own batch shard, like the reference's DDP; the only exchange is the all-reduce
of the scalar loss).  metric = elements * gauss_pts / s summed over ranks.

Prints ONE JSON line (rank 0).  Extra objects: `roofline` (time of the dominant kernel between timing-only HIP events,
taken over four DIFFERENT batches in rotation so that no launch finds its arrays in the Infinity Cache, vs the HBM peak of
MI355X_MICROARCH.md; the timed steps themselves re-evaluate one batch, whose kernel time is reported beside it) and `cpu_baseline` (the CPU oracle = port of the reference formulation, timed
on this box's host cores on a bounded sample of the same workload).
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


def timed_pairs(launch, n):
    """Duration of n launches, each between its own pair of HIP events recorded on the current (launch) stream.  The events are
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
        for a, b in pairs:
            a.record()
            launch()
            b.record()
        torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in pairs], "torch.cuda.Event (hipEventDefault)"
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
                                   f"{world} rank(s): 8-byte loss all-reduce + interface-layer exchange per step, overlapped with the slab "
                                   "kernel (BASELINE.json configs[3])",
                       "nodes": [n, n, n], "parallelism": f"slab x{world}"}}


def slab_main(args, rank, world, dev, dist):
    out = slab_leg(args, rank, world, dev, dist, args.size, args.batch, args.ngp, args.steps, args.warmup)
    if rank == 0:
        out["vs_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=400,
                    help="untimed steps before the timed ones.  Default 400 (about 20 ms of load): between about 1.3 and 10 ms after the "
                         "GPU leaves idle the launch runs 10-15 %% slower (power-management transient, tools/ramp2d.py, "
                         "profiles/r2_ramp2d.txt); 400 steps put the default run past it")
    ap.add_argument("--batch", type=int, default=64, help="samples per GPU")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--ngp", type=int, default=3)
    ap.add_argument("--nsd", type=int, default=2)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--bc", default="auto", choices=["auto", "bits", "u8", "f32", "box"],
                    help="how the Dirichlet condition is held (auto: box faces for 2-D, uint8 image for 3-D)")
    ap.add_argument("--slab", action="store_true",
                    help="strong-scaling variant (not the default metric run): ONE 3-D mesh of --size^3 nodes cut into z-slabs over "
                         "the ranks (diffnet_amd/slab.py): per step one 8-byte all-reduce + one node-layer exchange per interior face")
    ap.add_argument("--slab-size", type=int, default=256, help="mesh of the slab leg appended to the default run (0 = skip)")
    ap.add_argument("--slab-steps", type=int, default=100)
    ap.add_argument("--slab-warmup", type=int, default=300, help="untimed steps of the slab leg (past the load-onset transient: profiles/r2_ramp3d.txt)")
    ap.add_argument("--slab-batch", type=int, default=1, help="samples of the slab leg's mesh (BASELINE configs[3] is parametric: the reference trains it with batch 8)")
    ap.add_argument("--slab-timeout", type=float, default=180.0, help="watchdog of the slab leg, seconds")
    args = ap.parse_args()

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

    from diffnet_amd import DiffNet2DFEM, DiffNet3DFEM
    kw = dict(domain_size=args.size, ngp_1d=args.ngp, nsd=args.nsd)
    cls = DiffNet3DFEM if args.nsd == 3 else DiffNet2DFEM
    m = cls(None, **kw).to(dev)
    B = args.batch
    shape = (B, 1, *m.geom.node_shape)
    u, nu, f, bc = make_inputs(shape, dev, 42 + rank)
    c = 1.0
    units_per_step = B * m.geom.nelem_total * m.geom.ngp_total
    # The Dirichlet condition of BASELINE.md section 3 ("mask on all boundary faces") in the form the dataset keeps it in HBM:
    #   bits  one bit per node (diffnet_amd.PackedMask: packed once when the dataset is placed on the device; any mask)
    #   u8    one byte per node [default]   f32  the reference's fp32 image           box  derived from the geometry, no array
    from diffnet_amd import BoxFaces, PackedMask
    bc_form = args.bc if args.bc != "auto" else ("box" if args.nsd == 2 else "u8")
    # auto: BASELINE.md section 3 prescribes the condition "on all boundary faces", which is the box boundary the reference builds as an image
    # (IBN_2D.py:69-73); the kernel derives it from the geometry.  The same launch with the mask held as one bit per node (a general mask
    # array: 0-2 us slower depending on the box), as a uint8 or as the reference's fp32 image is timed below and reported in
    # roofline.one_batch_kernel_median_us_by_mask_format (the 3-D kernels read uint8 images).
    forms = {"u8": lambda: [(bc, 0.0)], "f32": lambda: [(bc.float(), 0.0)], "bits": lambda: [(PackedMask.pack(bc), 0.0)],
             "box": lambda: [(BoxFaces("all"), 0.0)]}
    dirichlet = forms[bc_form]()
    pending = []

    # The step is the prepared form of m.energy_loss_and_grad(u, nu, f, dirichlet, c) (diffnet_amd.ops.PoissonPlan: argument structs,
    # outputs and workspace set up once, one ctypes call per launch), PIPE + 1 of them in rotation so that a loss whose all-reduce is
    # still in flight is never overwritten by a later launch.
    from diffnet_amd import ops as _ops
    scale0 = 1.0 / (B * m.geom.nelem_total)
    plans = [_ops.PoissonPlan(m.geom, u, nu, f, None, dirichlet, alpha=2.0 * c, beta=1.0, c=c, wscale=1.0, out_scale=scale0,
                              want_out=True, want_sums=True, loss_scale=scale0) for _ in range(PIPE + 1 if dist is not None else 1)]
    turn = [0]

    def step():
        grad, _, loss = plans[turn[0]].launch()
        turn[0] = (turn[0] + 1) % len(plans)
        if dist is not None:
            # the path's only exchange step: all-reduce of the 4-byte loss (RCCL).  Issued asynchronously so that the
            # next evaluation's kernel does not queue behind the collective; waited PIPE steps later (a small-message
            # all-reduce over xGMI is latency-bound at tens of microseconds, comparable to one 75 us step) and drained
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
        while pending:
            pending.pop(0)[0].wait()

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
    region[1].record()
    drain()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    region_ms = region[0].elapsed_time(region[1]) / args.steps
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)

    # dominant-kernel time: HIP events on the launch stream around each dn_poisson_apply (ONE kernel: the fused
    # Poisson kernel, whose last workgroup also does the fixed-order final reduction), K launches
    # launches, each between its own pair of timing-only events (timed_pairs above).  A short run (the driver's is 5 + 20 steps, 1.3 ms of load) would put these launches into the
    # power-management transient 1.3-10 ms after load onset (tools/ramp2d.py: 55-58 us instead of 49.6), which says nothing about the
    # kernel; SETTLE untimed launches first carry the GPU past it.  The transient is reported as roofline.kernel_avg_ms_first_launches.
    from diffnet_amd import ops
    K, SETTLE, NROT = 100, 400, 4
    first_ms, _ = timed_pairs(plans[0].launch, 20)
    for _ in range(SETTLE):
        plans[0].launch()
    scale = 1.0 / (B * m.geom.nelem_total)
    same_ms, event_kind = timed_pairs(plans[0].launch, K)      # the same prepared launch as the timed steps (no allocation between the events)
    same_ms.sort()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    for a, b in evs:            # the same between default events (what earlier rounds reported)
        a.record()
        plans[0].launch()
        b.record()
    torch.cuda.synchronize()
    default_ev_ms = sum(a.elapsed_time(b) for a, b in evs) / K
    # The timed steps re-evaluate ONE batch, as the contract defines a step -- and its 268 MB of arrays are about the size of the 256 MB
    # Infinity Cache, so a launch finds part of its input there from the launch before (tools/rotate_batches.py: 46.3 us per launch on one
    # batch, 58.4 us on 2, 4 or 8 batches in rotation).  The ROOFLINE is about HBM: its kernel time is taken over NROT different batches
    # in rotation (own input and output arrays each), where no launch finds its data in a cache; the one-batch time is reported beside it.
    rot = [plans[0]]
    for k in range(1, NROT):
        uk, nuk, fk, _ = make_inputs(shape, dev, 1000 * k + 42 + rank)
        rot.append(_ops.PoissonPlan(m.geom, uk, nuk, fk, None, dirichlet, alpha=2.0 * c, beta=1.0, c=c, wscale=1.0, out_scale=scale0,
                                    want_out=True, want_sums=True, loss_scale=scale0))
    rot_turn = [0]

    def launch_rot():
        rot[rot_turn[0]].launch()
        rot_turn[0] = (rot_turn[0] + 1) % NROT

    for _ in range(SETTLE):
        launch_rot()
    kern_ms, _ = timed_pairs(launch_rot, K)
    kern_ms.sort()
    rot_region = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    rot_region[0].record()
    for _ in range(2 * K):
        launch_rot()
    rot_region[1].record()
    torch.cuda.synchronize()
    rot_region_ms = rot_region[0].elapsed_time(rot_region[1]) / (2 * K)
    bc_forms_us = {}
    if args.nsd == 2 and rank == 0:          # the same launch with the condition held in the other formats (median of 30, informational)
        for name, mk in forms.items():
            pl = plans[0] if name == bc_form else _ops.PoissonPlan(m.geom, u, nu, f, None, mk(), alpha=2.0 * c, beta=1.0, c=c, wscale=1.0,
                                                                   out_scale=scale0, want_out=True, want_sums=True, loss_scale=scale0)
            for _ in range(5):
                pl.launch()
            bc_forms_us[name] = round(sorted(timed_pairs(pl.launch, 30)[0])[15] * 1e3, 2)
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
        out = {
            "metric": "elements*gauss_pts/sec (FEM loss+grad)", "value": value, "unit": "elements*gauss_pts/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.nsd}-D Poisson energy loss + gradient wrt u, Q1, {args.size}^{args.nsd} nodes, "
                                   f"{args.ngp}^{args.nsd} Gauss pts, batch {B}/GPU, nu+f nodal fields, Dirichlet mask on all boundary faces "
                                   + {'bits': 'held as one bit per node', 'u8': 'held as a uint8 image', 'f32': 'held as an fp32 image', 'box': 'derived from the geometry (no array)'}[bc_form] + ", "
                                   "fused single pass (BASELINE.json configs[1] mesh)"
                                   + ("; 2-D Q1 element evaluated in closed form: the rule's sums as polynomials of its moments, same value "
                                      "as the per-point sum (dn_config_set(\"Q1_RULE_KERNEL\") runs the per-point kernel)" if args.nsd == 2 else ""),
                       "batch_per_gpu": B, "nodes": list(m.geom.node_shape), "parallelism": f"batch-sharded x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "poisson fused kernel (one launch per dn_poisson_apply)", "kernel_avg_ms": kern_avg_ms,
                         "kernel_median_ms": kern_med_ms, "kernel_min_ms": kern_ms[0], "kernel_max_ms": kern_ms[-1],
                         "frac_at_median": alg_bytes / (kern_med_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": alg_bytes,
                         "batches": "%d different batches in rotation (no launch finds its arrays in the Infinity Cache)" % NROT,
                         "rotating_batches_ms_per_launch_back_to_back": rot_region_ms,
                         "value_rotating_batches": units_per_step / (rot_region_ms * 1e-3),
                         "one_batch_kernel_avg_ms": sum(same_ms) / len(same_ms), "one_batch_kernel_median_ms": same_ms[len(same_ms) // 2],
                         "frac_one_batch": alg_bytes / (sum(same_ms) / len(same_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "cache_note": "the timed steps re-evaluate one batch (268 MB of arrays, the size of the 256 MB Infinity Cache): "
                                       "part of a launch's input is found there from the launch before; achieved / frac are taken on "
                                       "different batches in rotation instead, the one-batch kernel time is one_batch_kernel_avg_ms",
                         "kernel_avg_ms_first_launches": sum(first_ms) / len(first_ms),
                         "events": event_kind + " pair around each of %d launches, after %d untimed ones" % (K, SETTLE),
                         "kernel_avg_ms_default_events": default_ev_ms,
                         "timed_region_ms_per_launch": region_ms,
                         "frac_over_timed_region_one_batch": alg_bytes / (region_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "one_batch_kernel_median_us_by_mask_format": bc_forms_us},
        }
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(kw, c)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]

    # strong-scaling leg of BASELINE configs[3] in the same run (all ranks take part): one 256^3 mesh cut into z-slabs.  The headline
    # measurement above is complete at this point; the leg runs under a watchdog so that a failure or a hang in its point-to-point
    # exchange (first exercised over RCCL on the driver's multi-GPU node) still leaves the ONE JSON line, with the reason in "slab_3d".
    if args.slab_size and args.nsd == 2 and args.size == 512:
        import threading

        def give_up():
            if rank == 0:
                out["slab_3d"] = {"error": f"slab leg did not finish within {args.slab_timeout} s (n_gpus={world})"}
                print(json.dumps(out), flush=True)
            os._exit(0)

        dog = threading.Timer(args.slab_timeout, give_up)
        dog.daemon = True
        dog.start()
        try:
            slab = slab_leg(args, rank, world, dev, dist, args.slab_size, args.slab_batch, 2, args.slab_steps, args.slab_warmup)
            if rank == 0:
                out["slab_3d"] = slab
        except Exception as e:                       # noqa: BLE001 -- reported, not hidden: the line says what failed
            if rank == 0:
                out["slab_3d"] = {"error": f"{type(e).__name__}: {e}"[:500]}
                print(json.dumps(out), flush=True)
            os._exit(0)                              # peers may be stuck in a collective: no orderly teardown (their watchdogs end them)
        dog.cancel()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
