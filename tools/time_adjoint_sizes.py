import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import DiffNet2DFEM, _lib
dev = torch.device("cuda", 0)
def timed(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(20_000_000)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
L = _lib.lib()
import ctypes as C
for ngp in (2, 3, 4):
    for n in (384, 448, 500, 512, 513, 520, 576):
        B = 16
        m = DiffNet2DFEM(None, domain_size=n, ngp_1d=ngp).to(dev)
        G = ngp * ngp
        nel = n - 1
        gout = torch.rand((B, G, nel, nel), device=dev)
        gin = torch.empty((B, 1, n, n), device=dev)
        tab = torch.rand((G, 4), device=dev)
        nn = (C.c_int32 * 3)(n, n, 1)
        s = torch.cuda.current_stream().cuda_stream
        res = []
        for tiled in ("", "1"):
            _lib.config_set("GPE_TILED", tiled)
            fn = lambda: L.dn_gauss_pt_eval_bwd(C.c_void_p(gout.data_ptr()), C.c_void_p(tab.data_ptr()), C.c_void_p(gin.data_ptr()), B, 2, nn, 2, 1, G, C.c_void_p(s))
            assert fn() == 0
            t = timed(fn)
            res.append(t)
        byt = (gout.numel() + gin.numel()) * 4
        print(f"ngp={ngp} n={n}: march {res[0]:6.1f} us ({byt / res[0] / 1e3:5.0f} GB/s)  tiled {res[1]:6.1f} us ({byt / res[1] / 1e3:5.0f} GB/s)", flush=True)
_lib.config_set("GPE_TILED", "")
