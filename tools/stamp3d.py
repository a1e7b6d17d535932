#!/usr/bin/env python3
"""Per-wave cycle budget of the 3-D T16 kernel (diagnostic build -DDN_STAMP3D, tools/variant_build.sh stamp poisson3d_q1_g2.hip -DDN_STAMP3D):
DN_LIB_PATH=variants/libdn_stamp.so python tools/stamp3d.py [n] [B] [plan]"""
import ctypes as C
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import DiffNet3DFEM, _lib, ops   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
plan = sys.argv[3] if len(sys.argv) > 3 else ""
dev = torch.device("cuda:0")
m = DiffNet3DFEM(None, domain_size=n, nsd=3).to(dev)
if plan:
    _lib.config_set("PLAN3D", plan)
shape = (B, 1, n, n, n)
g = torch.Generator().manual_seed(1)
u, nu, f = (torch.rand(shape, generator=g).to(dev) for _ in range(3))
nu += 0.5
bc = torch.zeros(shape, dtype=torch.uint8, device=dev)
bc[..., 0] = 1; bc[..., -1] = 1; bc[..., 0, :] = 1; bc[..., -1, :] = 1; bc[:, :, 0] = 1; bc[:, :, -1] = 1
for _ in range(5):
    m.energy_loss_and_grad(u, nu, f, dirichlet=[(bc, 0.0)], c=1.0)
torch.cuda.synchronize()
buf = np.zeros(8192 * 8, dtype=np.uint64)
h = _lib.lib()
h.dn_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
rc = h.dn_debug_stamps(buf.ctypes.data, buf.nbytes)
assert rc == 0, rc
r = buf.reshape(-1, 8)
r = r[(r[:, 5] & np.uint64(0xffff)) > 0]
hwid = ((r[:, 5] >> np.uint64(16)) & np.uint64(0xffffffff)).astype(np.int64)     # node-owner build: HW_REG_HW_ID of the wave (0 in the T16 build)
xcd = ((r[:, 5] >> np.uint64(48)) & np.uint64(0xf)).astype(np.int64)             # HW_REG_XCC_ID
cu = (hwid >> 8) & 0xf
sh = (hwid >> 12) & 0x1
se = (hwid >> 13) & 0x7
rt = (r[:, 4] >> np.uint64(40)).astype(np.float64)          # node-owner build: wave lifetime on the constant 100 MHz clock
r[:, 4] = r[:, 4] & np.uint64((1 << 40) - 1)
a1 = (r[:, 0] >> np.uint64(32)).astype(np.float64)            # closed-form kernel (round 4): the part of phase A before the memory requests
r[:, 0] = r[:, 0] & np.uint64((1 << 32) - 1)
r = r.astype(np.float64)
r[:, 5] = (r[:, 5].astype(np.uint64) & np.uint64(0xffff)).astype(np.float64)
print(f"n={n} B={B} plan={plan or 'default'}: {len(r)} sampled waves, layers per wave in the loop: {r[:,5].mean():.1f}")
per = r[:, :5] / r[:, 5:6]
names = ["A request loads+store", "B gather (LDS) + stage", "C layer arithmetic", "D hand-over+publish+barrier", "E xch read+finish"]      # node-owner form (poisson3d_q1n_kernel); the T16 form (DN Q1_3D_T16=1): A wait+stage, B issue, C, D, E
tot = per.sum(1)
for i, nm in enumerate(names):
    print(f"  {nm:22s} mean {per[:, i].mean():8.0f}  median {np.median(per[:, i]):8.0f}  p10 {np.percentile(per[:, i], 10):8.0f}  p90 {np.percentile(per[:, i], 90):8.0f}  cycles per layer")
if a1.any():
    print(f"  (closed-form kernel: of phase A, LDS gather issue + publish {np.mean(a1 / r[:, 5]):.0f}, memory requests + deferred store {np.mean(r[:, 0] / r[:, 5]):.0f})")
    per[:, 0] += a1 / r[:, 5]
    tot = per.sum(1)
print(f"  {'total per layer':22s} mean {tot.mean():8.0f}  median {np.median(tot):8.0f}")
life = r[:, 7] - r[:, 6]
if rt.any():
    mhz = life / (rt * 10e-3)          # shader-clock ticks per microsecond
    print(f"  SHADER CLOCK while the waves ran (s_memtime ticks / s_memrealtime): mean {mhz.mean():.0f} MHz  min {mhz.min():.0f}  max {mhz.max():.0f};  wave lifetime {rt.mean() * 0.01:.1f} us")
print(f"  wave lifetime mean {life.mean():.0f} ticks (s_memtime: 100 MHz constant clock on gfx9? compare with total*layers = {(tot * r[:,5]).mean():.0f})")
keys = list(zip(xcd.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
print("  distinct (xcc, se, sh, cu) of the sampled waves:", len(set(keys)))
if hwid.any():          # per-CU timelines (node-owner build: one record per workgroup)
    import collections
    byc = collections.defaultdict(list)
    for kk, st, en in zip(keys, r[:, 6], r[:, 7]):
        byc[kk].append((st, en))
    cnt = np.array([len(v) for v in byc.values()])
    print(f"  workgroups per CU: min {cnt.min()} median {int(np.median(cnt))} max {cnt.max()}  (CUs used: {len(byc)})")
    spans, conc = [], []
    for v in byc.values():
        v.sort()
        t0c = v[0][0]
        spans.append(max(e for _, e in v) - t0c)
        # how many workgroups of this CU had started before the first one ended
        conc.append(sum(1 for st, _ in v if st < v[0][1]))
    spans = np.array(spans)
    print(f"  per-CU busy span (first start -> last end): median {np.median(spans):.0f}  max {spans.max():.0f} ticks;  workgroups started before the CU's first one ended: median {int(np.median(conc))} max {max(conc)}")
for x in range(8):
    sel = xcd == x
    if sel.sum() < 2:
        continue
    st = r[sel, 6] - r[sel, 6].min()
    en = r[sel, 7] - r[sel, 6].min()
    print(f"  XCD {x}: {sel.sum():3d} sampled waves, start after the first one: median {np.median(st):9.0f}  p75 {np.percentile(st, 75):9.0f}  max {st.max():9.0f};  last end {en.max():9.0f} ticks")
