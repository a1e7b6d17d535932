#!/usr/bin/env python3
"""Parametric immersed-boundary Poisson training in 3-D, the flow of the reference's IBN/poisson-3d/parametric/IBN_3D.py:109-162
(voxel object -> 3-D U-Net generator -> energy loss) on MI355X kernels, without Lightning:

    python examples/ibn_3d_parametric.py [--size 64] [--objects 16] [--epochs 6] [--batch 2] [--dropin]

The object library is synthetic (unions of ellipsoids written in the reference's one-npz-per-sample layout)."""
import argparse
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from DiffNet.DiffNetFEM import DiffNet3DFEM  # noqa: E402
from DiffNet.networks.wgan3d import GoodGenerator  # noqa: E402
from DiffNet.datasets.parametric.topo3d import TopoDataset3D, write_blob_objects  # noqa: E402
from diffnet_amd.datasets import DeviceLoader  # noqa: E402
from diffnet_amd.trainer import Trainer  # noqa: E402


class Poisson(DiffNet3DFEM):
    def __init__(self, network, dropin=False, **kwargs):
        super().__init__(network, **kwargs)
        self.dropin = dropin

    def loss(self, u, source_tensor, sink_tensor, forcing_tensor):
        f = forcing_tensor
        source = (source_tensor > 0.5).to(u.dtype)
        sink = torch.where(source == sink_tensor, sink_tensor * 0., sink_tensor)          # a source on the boundary wins (IBN_3D.py:121)
        if not self.dropin:
            return self.energy_loss(u, None, f, dirichlet=[(source, 1.0), (sink, 0.0)], c=1.0)
        u = torch.where(source > 0.5, 1. + (u * 0.), u)
        u = torch.where(sink > 0.5, u * 0., u)
        f_gp, u_gp = self.gauss_pt_evaluation(f), self.gauss_pt_evaluation(u)
        ux, uy, uz = self.gauss_pt_evaluation_der_x(u), self.gauss_pt_evaluation_der_y(u), self.gauss_pt_evaluation_der_z(u)
        w = self.gpw.unsqueeze(-1).unsqueeze(-1).unsqueeze(-1).unsqueeze(0).type_as(u)
        return torch.mean(torch.sum(w * (1. * (ux ** 2 + uy ** 2 + uz ** 2) - u_gp * f_gp), 1))

    def forward(self, batch):
        source_tensor, sink_tensor, forcing_tensor = batch
        return self.network(source_tensor), source_tensor, sink_tensor, forcing_tensor

    def training_step(self, batch, batch_idx):
        u, source_tensor, sink_tensor, forcing_tensor = self.forward(batch)
        loss = self.loss(u, source_tensor, sink_tensor, forcing_tensor).mean()
        self.log("train_loss", loss)
        return loss

    def configure_optimizers(self):
        opt = torch.optim.Adam(self.network.parameters(), lr=self.learning_rate)
        return [opt], [torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[10, 15, 30], gamma=0.1)]


def run(size=64, objects=16, epochs=6, batch=2, dropin=False, device="cuda:0", seed=0, verbose=True):
    torch.manual_seed(seed)
    with tempfile.TemporaryDirectory() as tmp:
        write_blob_objects(tmp, n_objects=objects, domain_size=size, seed=seed)
        ds = TopoDataset3D(tmp, domain_size=size, mode='train')
        loader = DeviceLoader(ds, batch_size=batch, device=device, shuffle=True)
    model = Poisson(GoodGenerator(1, 1), dropin=dropin, domain_size=size, nsd=3, ngp_1d=2, learning_rate=3e-4)
    tr = Trainer(max_epochs=epochs, device=device).fit(model, loader)
    per_epoch = [sum(tr.history[e * len(loader):(e + 1) * len(loader)]) / len(loader) for e in range(epochs)]
    if verbose:
        print(f"GoodGenerator {size}^3, {objects} objects, batch {batch}: epoch losses " + " ".join(f"{v:.4f}" for v in per_epoch))
    return per_epoch, model, loader


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--objects", type=int, default=16)
    ap.add_argument("--epochs", type=int, default=6)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--dropin", action="store_true")
    a = ap.parse_args()
    run(a.size, a.objects, a.epochs, a.batch, a.dropin)
