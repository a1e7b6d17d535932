#!/usr/bin/env python3
"""Round-4 timing harness: one fused Poisson launch (2-D or 3-D Q1) over NSETS field sets in rotation, prepared launches back to back,
steady state.  usage: r4_time.py <2|3> <n> <B> <bc: none|u8|f32|bits|box> [key=value ...]
keys: sums=1|0|defer|fold (in-kernel final reduction / no sums / per-workgroup partials only / partials folded by the next launch), load=1 (forcing as LoadVector), nsets=4, plan=<PLAN2D|PLAN3D override>,
      nu=1|0, f=1|0, pad=<bytes> (stagger the fields' start addresses), reps=3, iters=400, tag=<label>, cfg=KEY:VALUE[,...] (dn_config_set).  DN_LIB_PATH selects a variant build."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import BoxFaces, DiffNet2DFEM, DiffNet3DFEM, LoadVector, PackedMask, _lib, ops

nsd, n, B, form = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
kv = dict(a.split("=", 1) for a in sys.argv[5:])
dev = torch.device("cuda:0")
sizes = tuple(int(v) for v in kv["sizes"].split(",")) if "sizes" in kv else None      # sizes=nx,ny[,nz]: a non-cubic mesh (n is then ignored)
if sizes:
    m = (DiffNet3DFEM(None, domain_size=sizes[0], domain_sizes=sizes, domain_lengths=tuple(float(v - 1) for v in sizes), nsd=3) if nsd == 3 else
         DiffNet2DFEM(None, domain_size=sizes[0], domain_sizes=sizes, domain_lengths=tuple(float(v - 1) for v in sizes), ngp_1d=int(kv.get("ngp", 3)))).to(dev)
else:
    m = (DiffNet3DFEM(None, domain_size=n, nsd=3) if nsd == 3 else DiffNet2DFEM(None, domain_size=n, ngp_1d=int(kv.get("ngp", 3)))).to(dev)
for c in filter(None, kv.get("cfg", "").split(",")):          # cfg=KEY:VALUE[,KEY:VALUE]: dn_config_set switches (e.g. cfg=Q1_3D_N2:1)
    _lib.config_set(*c.split(":", 1))
if "plan" in kv:
    _lib.config_set("PLAN3D" if nsd == 3 else "PLAN2D", kv["plan"])
shape = (B, 1) + (tuple(reversed(sizes)) if sizes else (n,) * nsd)
g = torch.Generator().manual_seed(1)
nsets = int(kv.get("nsets", 4))
sets = []
pad = int(kv.get("pad", 0))             # pad=<bytes>: every field starts at a different multiple of this many bytes inside its allocation (array-to-array alignment experiments)
_slot = [0]


def _padded(t):
    if pad == 0:
        return t
    n_el = t.numel()
    buf = torch.empty(n_el + 16 * (pad // 4), device=dev)
    off = (_slot[0] % 16) * (pad // 4)
    _slot[0] += 5
    v = buf[off:off + n_el].view(t.shape)
    v.copy_(t)
    return v


for k in range(nsets):
    u, nu, f = (_padded(torch.rand(shape, generator=g).to(dev)) for _ in range(3))
    nu += 0.5
    sets.append((u, nu if kv.get("nu", "1") == "1" else None, f if kv.get("f", "1") == "1" else None))
bc = torch.zeros(shape, dtype=torch.uint8, device=dev)
for d in range(nsd):
    idx = [slice(None)] * len(shape)
    idx[2 + d] = 0
    bc[tuple(idx)] = 1
    idx[2 + d] = -1
    bc[tuple(idx)] = 1
cond = {"none": lambda: [], "box": lambda: [(BoxFaces(), 0.0)], "bits": lambda: [(PackedMask.pack(bc.clone()), 0.0)], "u8": lambda: [(bc.clone(), 0.0)],
        "f32": lambda: [(bc.float(), 0.0)]}[form]
scale = 1.0 / (B * m.geom.nelem_total)
sums = kv.get("sums", "1")
kw = dict(alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=scale, want_out=True, want_sums=sums != "0", loss_scale=scale if sums != "0" else None)
if kv.get("load", "0") == "1":         # the forcing as its assembled load vector (ops.LoadVector): checked against the nodal-forcing launch first
    ref = ops.PoissonPlan(m.geom, *sets[0], None, cond(), **dict(kw, want_sums=True, loss_scale=scale)).launch()
    ref = [t.clone() for t in ref]
    sets = [(u, nu, LoadVector.assemble(m.geom, f) if f is not None else None) for (u, nu, f) in sets]
    got = ops.PoissonPlan(m.geom, *sets[0], None, cond(), **dict(kw, want_sums=True, loss_scale=scale)).launch()
    dg = float((got[0] - ref[0]).abs().max() / ref[0].abs().max())
    print(f"load vector vs nodal forcing: max |d grad| / max |grad| = {dg:.2e}; loss {float(got[2]):.8e} vs {float(ref[2]):.8e}; sumsq rel {abs(float(got[1][1]) / float(ref[1][1]) - 1):.2e}", flush=True)
pipelined = sums == "fold"
plans = [ops.PoissonPlan(m.geom, *sets[k], None, cond(), pipelined_sums=pipelined, **kw) for k in range(nsets)]
if pipelined:                          # every launch closes the one before it (dn_poisson_args.fold_prev)
    for k in range(nsets):
        plans[k].fold(plans[k - 1])
    ref = ops.PoissonPlan(m.geom, *sets[0], None, cond(), **kw).launch()
    ref = [t.clone() for t in ref]
    plans[0].fold(None)
    plans[0].launch(); plans[1].launch()
    torch.cuda.synchronize()
    r0 = plans[0].result
    print(f"folded sums vs in-kernel sums: loss {float(r0[2]):.9e} vs {float(ref[2]):.9e}; energy rel {abs(float(r0[1][0]) / float(ref[1][0]) - 1):.1e}; sumsq rel {abs(float(r0[1][1]) / float(ref[1][1]) - 1):.1e}; "
          f"grad equal {bool(torch.equal(r0[0], ref[0]))}", flush=True)
    plans[0].fold(plans[-1])
if sums == "defer":
    for p in plans:
        p.args.defer_sums = 1
iters, reps = int(kv.get("iters", 400)), int(kv.get("reps", 3))
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.05:
    for i in range(24):
        plans[i % nsets].launch()
    torch.cuda.synchronize()
ts = []
for _ in range(reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        plans[i % nsets].launch()
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) * 1000.0 / iters)
nodes = B * (sizes[0] * sizes[1] * (sizes[2] if nsd == 3 else 1) if sizes else n ** nsd)
med = sorted(ts)[len(ts) // 2]
print(f"{kv.get('tag', '')} sizes={kv.get('sizes', '-')} lib={os.path.basename(os.environ.get('DN_LIB_PATH', 'default'))} {nsd}-D n={n} B={B} bc={form} sums={sums} plan={kv.get('plan', 'default')} cfg={kv.get('cfg', '-')} "
      f"nu={kv.get('nu', '1')} f={kv.get('f', '1')} nsets={nsets}: {med:.2f} us per launch  ({16.0 * nodes / med * 1e-6:.3f} TB/s of 16 B/node)  {[round(t, 2) for t in ts]}", flush=True)
