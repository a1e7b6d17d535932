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
