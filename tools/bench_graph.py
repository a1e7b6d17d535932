#!/usr/bin/env python3
"""Launch-bound regime (BASELINE configs[0]: 64 x 64 non-parametric Poisson): iterations/s of the eager fit loop vs the
HIP-graph replay of the same iteration (forward + fused loss kernel + backward + Adam update)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
from diffnet_amd import DiffNet2DFEM
from diffnet_amd.trainer import Trainer


class P(DiffNet2DFEM):
    def training_step(self, batch, idx):
        nu, f, bc = batch
        return self.energy_loss(self.network[0], nu, f, dirichlet=[(bc, 0.0)], c=0.5)

    def configure_optimizers(self):
        return [torch.optim.Adam(self.network.parameters(), lr=1e-2)], []


for n in (64, 256):
    res = {}
    for graph in (False, True):
        torch.manual_seed(0)
        net = nn.ParameterList([nn.Parameter(torch.zeros(1, 1, n, n))])
        m = P(net, domain_size=n, ngp_1d=2)
        bc = torch.zeros(1, 1, n, n, dtype=torch.uint8)
        bc[..., 0] = bc[..., -1] = 1
        batch = (torch.ones(1, 1, n, n), torch.ones(1, 1, n, n), bc)
        steps = 2000
        tr = Trainer(max_epochs=steps, graph=graph, log_every=steps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.fit(m, [batch])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[graph] = (steps / dt, tr.history[-1])
    print(f"{n}x{n} Q1 2x2, Adam, {steps} iterations: eager {res[False][0]:8.0f} it/s   graph {res[True][0]:8.0f} it/s   "
          f"x{res[True][0] / res[False][0]:.1f}   final loss {res[False][1]:.6f} / {res[True][1]:.6f}", flush=True)


# parametric case: U-Net (2 -> 1) on one static batch; the captured iteration includes the MIOpen convolutions, the HIP
# InstanceNorm / output-block kernels, dropout (graph-safe Philox offsets) and Adam
from diffnet_amd.networks.unets import UNet  # noqa: E402


class PU(DiffNet2DFEM):
    def training_step(self, batch, idx):
        x, nu, f, bc = batch
        return self.energy_loss(self.network(x), nu, f, dirichlet=[(bc, 0.0)], c=0.5)

    def configure_optimizers(self):
        return [torch.optim.Adam(self.network.parameters(), lr=1e-4)], []


for n, B in ((64, 8), (128, 8)):
    res = {}
    for graph in (False, True):
        torch.manual_seed(0)
        m = PU(UNet(2, 1), domain_size=n, ngp_1d=3)
        nu = torch.rand(B, 1, n, n) + 0.5
        bc = torch.zeros(B, 1, n, n, dtype=torch.uint8)
        bc[..., 0] = bc[..., -1] = 1
        batch = (torch.cat([nu, bc.float()], 1), nu, torch.rand(B, 1, n, n), bc)
        steps = 300
        tr = Trainer(max_epochs=steps, graph=graph, log_every=steps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.fit(m, [batch])
        torch.cuda.synchronize()
        res[graph] = steps / (time.perf_counter() - t0)
    print(f"U-Net {n}x{n} batch {B}, Adam, {steps} iterations: eager {res[False]:7.0f} it/s   graph {res[True]:7.0f} it/s   x{res[True] / res[False]:.1f}", flush=True)
