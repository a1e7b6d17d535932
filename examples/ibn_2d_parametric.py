#!/usr/bin/env python3
"""Parametric immersed-boundary Poisson training, the flow of the reference's flagship script
IBN/poisson-2d/parametric/IBN_2D.py:111-170, on MI355X kernels end to end and without Lightning:

    boundary point cloud --dn_winding_nodes--> inside/outside mask --network--> u --dn_poisson_apply--> loss, dloss/du

    python examples/ibn_2d_parametric.py [--size 64] [--shapes 64] [--epochs 8] [--batch 16] [--net unet|ae] [--dropin]

The shape library is synthetic (star-shaped closed curves written in the reference's npz layout).  --dropin evaluates the
reference's loss body unchanged on the drop-in `gauss_pt_evaluation*` operators instead of the fused kernel."""
import argparse
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from DiffNet.DiffNetFEM import DiffNet2DFEM  # noqa: E402
from DiffNet.networks.autoencoders import AE  # noqa: E402
from DiffNet.networks.unets import UNet  # noqa: E402
from DiffNet.datasets.parametric.pointclouds import PointClouds, write_star_shapes  # noqa: E402
from diffnet_amd.datasets import DeviceLoader  # noqa: E402
from diffnet_amd.ops import compute_winding_nodes  # noqa: E402
from diffnet_amd.trainer import Trainer  # noqa: E402


class Poisson(DiffNet2DFEM):
    def __init__(self, network, dropin=False, **kwargs):
        super().__init__(network, **kwargs)
        self.dropin = dropin

    def loss(self, u, source_tensor, f, sink_tensor):
        if not self.dropin:        # u = 1 on the object, u = 0 on the outer boundary, nu = 1, c = 1 (IBN_2D.py:116-134)
            return self.energy_loss(u, None, f, dirichlet=[(source_tensor, 1.0), (sink_tensor, 0.0)], c=1.0)
        u = torch.where(source_tensor > 0.5, 1. + u * 0., u)
        u = torch.where(sink_tensor > 0.5, u * 0., u)
        nu_gp, f_gp, u_gp = self.gauss_pt_evaluation(torch.ones_like(u)), self.gauss_pt_evaluation(f), self.gauss_pt_evaluation(u)
        u_x_gp, u_y_gp = self.gauss_pt_evaluation_der_x(u), self.gauss_pt_evaluation_der_y(u)
        w = self.gpw.unsqueeze(-1).unsqueeze(-1).unsqueeze(0).type_as(u)
        return torch.mean(torch.sum(w * (nu_gp * (u_x_gp ** 2 + u_y_gp ** 2) - u_gp * f_gp), 1))

    def forward(self, batch):
        inputs_tensor, forcing_tensor, sink_tensor = batch
        pc = inputs_tensor[:, :, 0:2].unsqueeze(1)
        normals = inputs_tensor[:, :, 2:4].unsqueeze(1)
        area = inputs_tensor[:, :, 4:5].unsqueeze(1)
        nodes = torch.stack((self.xx, self.yy), 0).type_as(pc)
        source_tensor = compute_winding_nodes(pc, normals, area, nodes)
        source_tensor = (source_tensor > 0.005).to(source_tensor.dtype)
        return self.network(source_tensor), source_tensor, forcing_tensor, sink_tensor

    def training_step(self, batch, batch_idx):
        u, source_tensor, forcing_tensor, sink_tensor = self.forward(batch)
        loss = self.loss(u, source_tensor, forcing_tensor, sink_tensor).mean()
        self.log("train_loss", loss)
        return {"loss": loss}

    def configure_optimizers(self):
        opt = torch.optim.Adam(self.network.parameters(), lr=self.learning_rate)
        return [opt], [torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[10, 15, 30], gamma=0.1)]


def run(size=64, shapes=64, epochs=8, batch=16, net="unet", dropin=False, device="cuda:0", seed=0, verbose=True):
    torch.manual_seed(seed)
    with tempfile.TemporaryDirectory() as tmp:
        prefix = tmp + os.sep
        write_star_shapes(prefix, n_shapes=shapes, n_points=400, seed=seed)
        PointClouds.n_val = 0
        ds = PointClouds(prefix, type='train', domain_size=size)
    network = UNet(1, 1) if net == "unet" else AE(1, 1, n_downsample=2)
    model = Poisson(network, dropin=dropin, domain_size=size, ngp_1d=3, learning_rate=3e-4)
    loader = DeviceLoader(ds, batch_size=batch, device=device, shuffle=True)
    tr = Trainer(max_epochs=epochs, device=device).fit(model, loader)
    per_epoch = [sum(tr.history[e * len(loader):(e + 1) * len(loader)]) / len(loader) for e in range(epochs)]
    if verbose:
        print(f"{net} {size}x{size}, {shapes} shapes, batch {batch}: epoch losses " + " ".join(f"{v:.5f}" for v in per_epoch))
    return per_epoch, model, loader


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--shapes", type=int, default=64)
    ap.add_argument("--epochs", type=int, default=8)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--net", default="unet", choices=["unet", "ae"])
    ap.add_argument("--dropin", action="store_true")
    a = ap.parse_args()
    run(a.size, a.shapes, a.epochs, a.batch, a.net, a.dropin)
