#!/usr/bin/env python3
"""Non-parametric 2-D Poisson by energy minimisation on the fused HIP loss -- the shape of the reference's
`examples/poisson/single_instance/e8_2d_poisson_mms.py` (field-as-parameter network, manufactured solution
u = sin(pi x) sin(pi y), f = 2 pi^2 u, Dirichlet u = 0 on the boundary), without Lightning.

    python examples/poisson_2d_energy.py [--size 65] [--epochs 60] [--dropin]

--dropin runs the reference's own loss body (gauss_pt_evaluation* calls + torch ops) instead of the fused kernel."""
import argparse
import math
import os
import sys

import numpy as np
import torch
from torch import nn

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from DiffNet.DiffNetFEM import DiffNet2DFEM  # noqa: E402  (reference import path, MI355X implementation)
from diffnet_amd.trainer import Trainer  # noqa: E402


class Poisson(DiffNet2DFEM):
    def __init__(self, network, dropin=False, **kwargs):
        super().__init__(network, **kwargs)
        self.dropin = dropin
        self.u_exact = np.sin(math.pi * self.xx.numpy()) * np.sin(math.pi * self.yy.numpy())

    def exact_solution(self, x, y):
        return torch.sin(math.pi * x) * torch.sin(math.pi * y)

    def loss(self, u, inputs_tensor, forcing_tensor):
        nu, bc = inputs_tensor[:, 0:1].contiguous(), inputs_tensor[:, 1:2].contiguous()
        jac = (0.5 * self.hx) * (0.5 * self.hy)
        if not self.dropin:
            return self.energy_loss(u, nu, forcing_tensor, dirichlet=[(bc, 0.0)], c=0.5, jac=jac)
        # the reference formulation, unchanged, on the drop-in operators
        u = torch.where(bc > 0.5, u * 0.0, u)
        nu_gp, f_gp, u_gp = self.gauss_pt_evaluation(nu), self.gauss_pt_evaluation(forcing_tensor), self.gauss_pt_evaluation(u)
        u_x_gp, u_y_gp = self.gauss_pt_evaluation_der_x(u), self.gauss_pt_evaluation_der_y(u)
        w = (self.gpw * jac).unsqueeze(-1).unsqueeze(-1).unsqueeze(0).type_as(u)
        return torch.mean(torch.sum(w * (0.5 * nu_gp * (u_x_gp ** 2 + u_y_gp ** 2) - u_gp * f_gp), 1))

    def forward(self, batch):
        inputs_tensor, forcing_tensor = batch
        return self.network[0], inputs_tensor, forcing_tensor

    def training_step(self, batch, batch_idx):
        u, inputs_tensor, forcing_tensor = self.forward(batch)
        loss_val = self.loss(u, inputs_tensor, forcing_tensor).mean()
        self.log("loss", loss_val)
        return {"loss": loss_val}

    def configure_optimizers(self):
        return [torch.optim.LBFGS(self.network, lr=1.0, max_iter=20, line_search_fn="strong_wolfe")], []


def run(size=65, epochs=60, dropin=False, device="cuda:0", verbose=True):
    net = nn.ParameterList([nn.Parameter(torch.zeros(1, 1, size, size))])
    m = Poisson(net, dropin=dropin, domain_size=size, ngp_1d=2)
    bc = torch.zeros(1, 1, size, size)
    bc[..., 0, :] = bc[..., -1, :] = bc[..., :, 0] = bc[..., :, -1] = 1
    inputs = torch.cat([torch.ones(1, 1, size, size), bc], 1)
    forcing = (2 * math.pi ** 2) * torch.as_tensor(m.u_exact, dtype=torch.float32)[None, None]
    tr = Trainer(max_epochs=epochs, device=device).fit(m, [(inputs, forcing)])
    u = m.network[0].detach()
    eL2, uL2, _ = m._l2_terms(u)
    if verbose:
        print(f"size {size}  epochs {epochs}  loss {tr.history[0]:.6f} -> {tr.history[-1]:.6f}  relative L2 error {float(eL2 / uL2):.3e}")
    return float(eL2 / uL2), tr.history


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=65)
    ap.add_argument("--epochs", type=int, default=60)
    ap.add_argument("--dropin", action="store_true")
    a = ap.parse_args()
    run(a.size, a.epochs, a.dropin)
