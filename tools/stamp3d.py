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
r = r[r[:, 5] > 0].astype(np.float64)
print(f"n={n} B={B} plan={plan or 'default'}: {len(r)} sampled waves, layers per wave in the loop: {r[:,5].mean():.1f}")
per = r[:, :5] / r[:, 5:6]
names = ["A wait+stage plane", "B issue loads+store", "C layer arithmetic", "D xch write+barrier", "E xch read+finish"]
tot = per.sum(1)
for i, nm in enumerate(names):
    print(f"  {nm:22s} mean {per[:, i].mean():8.0f}  median {np.median(per[:, i]):8.0f}  p10 {np.percentile(per[:, i], 10):8.0f}  p90 {np.percentile(per[:, i], 90):8.0f}  cycles per layer")
print(f"  {'total per layer':22s} mean {tot.mean():8.0f}  median {np.median(tot):8.0f}")
life = r[:, 7] - r[:, 6]
print(f"  wave lifetime mean {life.mean():.0f} ticks (s_memtime: 100 MHz constant clock on gfx9? compare with total*layers = {(tot * r[:,5]).mean():.0f})")
t0 = r[:, 6] - r[:, 6].min()
print(f"  wave start times: min 0  median {np.median(t0):.0f}  max {t0.max():.0f};  end max {(r[:,7]-r[:,6].min()).max():.0f}")
