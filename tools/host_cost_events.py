import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from diffnet_amd import DiffNet2DFEM, ops, PackedMask
dev = torch.device("cuda:0")
m = DiffNet2DFEM(None, domain_size=64, ngp_1d=3).to(dev)
B = 2
sh = (B, 1, 64, 64)
u, nu, f = (torch.rand(sh, device=dev) for _ in range(3))
bc = torch.zeros(sh, device=dev, dtype=torch.uint8); bc[..., 0] = 1
pls = [ops.PoissonPlan(m.geom, u, nu, f, None, [(PackedMask.pack(bc), 0.0)], alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=1.0, want_out=True, want_sums=True,
                       loss_scale=1.0, pipelined_sums=True) for _ in range(4)]
for k in range(4):
    pls[k].fold(pls[k - 1])
for _ in range(200):
    pls[0].launch()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(4000):
        pls[i & 3].launch()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host enqueue {1e6 * (t1 - t0) / 4000:.2f} us per launch, with drain {1e6 * (t2 - t0) / 4000:.2f}", flush=True)
import ctypes
path = next(ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64.so" in ln)
print("hip instance used by timed_pairs:", path)
print("all hip instances:", sorted(set(ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln)))
hip = ctypes.CDLL(path)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
evs = []
for _ in range(2000):
    e = ctypes.c_void_p(); hip.hipEventCreateWithFlags(ctypes.byref(e), ctypes.c_uint(0x20000000)); evs.append(e)
t0 = time.perf_counter()
for i in range(1000):
    hip.hipEventRecord(evs[2 * i], stream); pls[i & 3].launch(); hip.hipEventRecord(evs[2 * i + 1], stream)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host per bracketed launch {1e6 * (t1 - t0) / 1000:.2f} us", flush=True)
