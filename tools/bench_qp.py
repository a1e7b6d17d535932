import sys; sys.path.insert(0,'/root/repo')
import torch
from diffnet_amd import DiffNet2DFEM
dev=torch.device('cuda',0)
def timed(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(20_000_000); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n*1e3
for deg,n,ngp,B in [(2,513,3,16),(2,1025,3,16),(3,511,4,16),(2,513,4,16),(1,512,3,16)]:
    m=DiffNet2DFEM(None,domain_size=n,fem_basis_deg=deg,ngp_1d=ngp).to(dev)
    u,nu,f=(torch.rand(B,1,n,n,device=dev) for _ in range(3))
    bc=torch.zeros(B,1,n,n,device=dev,dtype=torch.uint8); bc[...,0]=1
    t=timed(lambda: m.energy_loss_and_grad(u,nu,f,dirichlet=[(bc,0.0)],c=1.0))
    print(f"Q{deg} n={n} ngp={ngp} B={B}: {t:8.1f} us  {16*B*n*n/t/1e3:7.0f} GB/s algorithmic", flush=True)
