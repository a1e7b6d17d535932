#!/usr/bin/env python3
"""Steady-state rate of the 3-D kernel vs resident workgroups per CU: 241^3 nodes = 16 x 16 tiles of 16 x 16 threads = 256 workgroups
per sample, ONE strip of 240 layers per workgroup, B = k samples -> k workgroups per CU all resident for the whole launch (no
ramp, no tail).  Prints SIMD-cycles per VALU wave-instruction (260 per element-layer, 2.4 GHz)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import _lib   # noqa: E402
if os.environ.get("DN_LIB_PATH"):          # timing experiments: a variant build (tools/variant_build.sh)
    _lib.LIB_PATH = os.path.abspath(os.environ["DN_LIB_PATH"])
from diffnet_amd import DiffNet3DFEM, ops   # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 241
R = n - 1
m = DiffNet3DFEM(None, domain_size=n, nsd=3).to(dev)
try:
    print("hipOccupancyMaxActiveBlocksPerMultiprocessor(poisson3d_q1n_kernel, 256 threads) =", _lib.lib().dn_debug_occupancy_q1n(), flush=True)
except AttributeError:
    pass
_lib.config_set("PLAN3D", f"16,16,1,{R}")
for B in [int(v) for v in (sys.argv[2].split(',') if len(sys.argv) > 2 else '1,2,3,4,5,6,7,8'.split(','))]:
    shape = (B, 1, n, n, n)
    g = torch.Generator().manual_seed(1)
    u, nu, f = (torch.rand(shape, generator=g).to(dev) for _ in range(3))
    nu += 0.5
    bc = torch.zeros(shape, dtype=torch.uint8, device=dev)
    bc[..., 0] = 1; bc[..., -1] = 1; bc[..., 0, :] = 1; bc[..., -1, :] = 1; bc[:, :, 0] = 1; bc[:, :, -1] = 1
    fn = lambda: m.energy_loss_and_grad(u, nu, f, dirichlet=[(bc, 0.0)], c=1.0)
    ops._POISSON_WS_BYTES.clear()
    for _ in range(3):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    med = ts[len(ts) // 2]
    tiles = -(-(n - 1) // 15)
    wgs = tiles * tiles * B
    instr_per_simd = wgs * 4 * R * 260 / 1024
    print(f"n={n} B={B} workgroups={wgs} ({wgs/256:.2f}/CU) median_us={med:.1f} min_us={ts[0]:.1f}  cycles/VALU-instr/SIMD={med*1e-6*2.4e9/instr_per_simd:.2f}", flush=True)
