#!/usr/bin/env python3
"""First-order shear-deformation (Mindlin) plate under a uniform load, clamped on all four edges, solved by minimising the norms of the three
weak-form residuals over the nodal fields (w, phi_x, phi_y) -- the shape of the reference's
`examples/elasticity/single_instance/e1_plate_bending_fsdt.py` (fields as parameters, one Adam optimiser per field stepping on its own
residual norm, the material constants of its `calc_residuals`), without Lightning, on the fused HIP loss: the script's loss body
(:128-232 -- nine gauss_pt_evaluation calls, ~40 elementwise ops, three assemblies, three norms) is ONE launch, its backward another.

    python examples/plate_bending_fsdt.py [--size 33] [--degree 1] [--epochs 250] [--lr 4e-3] [--mode reference|total|plan]

--mode reference  the reference's training scheme: per epoch, for each of the three optimisers, loss_k = ||R_k|| -> backward -> step
       total      one optimiser over all three fields on ||R1|| + ||R2|| + ||R3|| (fsdt_total_loss: one autograd node)
       plan       the same objective without autograd: ops.FsdtPlan (two prepared launches, sums deferred to the second) + Adam on the
                  gradients it returns
"""
import argparse
import os
import sys
import time

import torch
from torch import nn

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from DiffNet.DiffNetFEM import DiffNet2DFEM  # noqa: E402  (reference import path, MI355X implementation)
from diffnet_amd import ops  # noqa: E402
from diffnet_amd.elasticity import _constants, fsdt_loss, fsdt_total_loss  # noqa: E402


class ElasticFSDT(DiffNet2DFEM):
    """The reference's `Elastic_FSDT` (e1_plate_bending_fsdt.py:89-232): three field "networks", clamped edges, unit load."""

    def __init__(self, fields, **kwargs):
        super().__init__(None, **kwargs)
        self.net_w, self.net_phi_x, self.net_phi_y = fields
        n = self.domain_size
        bc = torch.zeros((1, 1, n, n))
        bc[..., 0, :] = 1.0
        bc[..., -1, :] = 1.0
        bc[..., :, 0] = 1.0
        bc[..., :, -1] = 1.0
        self.register_buffer("bc", bc)

    def fields(self):
        return self.net_w[0], self.net_phi_x[0], self.net_phi_y[0]

    def loss(self):
        """(||R1||, ||R2||, ||R3||) -- e1_plate_bending_fsdt.py:230-232"""
        return fsdt_loss(self, *self.fields(), self.bc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=33)
    ap.add_argument("--degree", type=int, default=1)
    ap.add_argument("--epochs", type=int, default=250)
    ap.add_argument("--lr", type=float, default=4e-3)
    ap.add_argument("--mode", choices=("reference", "total", "plan"), default="reference")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(42)
    n = args.size
    mk = lambda: nn.ParameterList([nn.Parameter(torch.zeros((1, 1, n, n)))])          # the reference starts from zero fields
    model = ElasticFSDT((mk(), mk(), mk()), domain_size=n, fem_basis_deg=args.degree, ngp_1d=2 if args.degree == 1 else 3).to(dev)
    params = [model.net_w, model.net_phi_x, model.net_phi_y]
    t0 = time.perf_counter()
    if args.mode == "reference":
        opts = [torch.optim.Adam(p.parameters(), lr=args.lr) for p in params]
        for ep in range(args.epochs):
            for k, opt in enumerate(opts):                      # Lightning's multiple-optimiser loop: optimiser k steps on loss_vals[k]
                opt.zero_grad(set_to_none=True)
                norms = model.loss()
                norms[k].backward()
                opt.step()
            if ep % 25 == 0 or ep == args.epochs - 1:
                print(f"epoch {ep:4d}  ||R1|| {float(norms[0]):.6e}  ||R2|| {float(norms[1]):.6e}  ||R3|| {float(norms[2]):.6e}", flush=True)
    elif args.mode == "total":
        opt = torch.optim.Adam([q for p in params for q in p.parameters()], lr=args.lr)
        for ep in range(args.epochs):
            opt.zero_grad(set_to_none=True)
            loss = fsdt_total_loss(model, *model.fields(), model.bc)
            loss.backward()
            opt.step()
            if ep % 25 == 0 or ep == args.epochs - 1:
                print(f"epoch {ep:4d}  ||R1|| + ||R2|| + ||R3|| {float(loss):.6e}", flush=True)
    else:
        flds = [p[0] for p in params]
        opt = torch.optim.Adam(flds, lr=args.lr)
        with torch.no_grad():
            plan = ops.FsdtPlan(model.geom, *[f.data for f in flds], model.bc, (0.0, 0.0, 0.0), q=1.0, wscale=(0.5 * model.h) ** 2,
                                **_constants(1.0, 0.25, 0.1, 1.0))
        for ep in range(args.epochs):
            norms, grads = plan.launch()                        # reads the parameters' storage in place: two launches, no autograd graph
            for f, g in zip(flds, grads):
                f.grad = g
            opt.step()
            if ep % 25 == 0 or ep == args.epochs - 1:
                print(f"epoch {ep:4d}  ||R1|| + ||R2|| + ||R3|| {float(norms.sum()):.6e}", flush=True)
    torch.cuda.synchronize()
    w = model.net_w[0].detach()
    print(f"{args.epochs} epochs in {time.perf_counter() - t0:.2f} s; centre deflection w = {float(w[0, 0, n // 2, n // 2]):.6e}, max |w| = {float(w.abs().max()):.6e}")
    return model


if __name__ == "__main__":
    main()
