"""Networks: same parameter count, state_dict keys, seeded default initialisation and
eval-mode outputs / gradients as the reference (golden vectors from tools/gen_golden.py)."""
import os
import warnings

import numpy as np
import pytest
import torch

from conftest import GOLDEN

SPECS = {
    "unet_2_1_n64": ("unets", "UNet", dict(in_channels=2, out_channels=1)),
    "unet_2_1_n96": ("unets", "UNet", dict(in_channels=2, out_channels=1)),
    "ae_1_1_d2_n32": ("autoencoders", "AE", dict(in_channels=1, out_channels=1, n_downsample=2)),
    "goodgen3d_1_1_n32": ("wgan3d", "GoodGenerator", dict(in_channels=1, out_channels=1)),
}


def build(name):
    import importlib
    mod, cls, kw = SPECS[name]
    torch.manual_seed(2024)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return getattr(importlib.import_module("DiffNet.networks." + mod), cls)(**kw).eval()


def check(net, z, dev):
    net = net.to(dev)
    x = torch.from_numpy(z["x"]).to(dev).requires_grad_(True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        y = net(x)
    tol = dict(rtol=2e-4, atol=2e-5) if dev.type == "cuda" else dict(rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(y.detach().cpu().numpy(), z["y"], **tol)
    cot = torch.from_numpy(z["cot"]).to(dev)
    gx, = torch.autograd.grad(y, x, cot, retain_graph=True)
    gw, = torch.autograd.grad(y, list(net.parameters())[0], cot)
    # conv backward reduction order depends on the thread count / library (oneDNN here, MIOpen on the GPU)
    gtol = 2e-3 if dev.type == "cuda" else 1e-3
    np.testing.assert_allclose(gx.cpu().numpy(), z["grad_x"], rtol=gtol, atol=gtol * 0.1 * float(np.abs(z["grad_x"]).max()))
    np.testing.assert_allclose(gw.cpu().numpy(), z["grad_w0"], rtol=gtol, atol=gtol * 0.1 * float(np.abs(z["grad_w0"]).max()))


@pytest.mark.parametrize("name", list(SPECS))
def test_network_matches_reference_cpu(name):
    z = np.load(os.path.join(GOLDEN, f"net_{name}.npz"))
    net = build(name)
    sd = net.state_dict()
    assert list(sd.keys()) == list(z["keys"])
    assert sum(p.numel() for p in net.parameters()) == int(z["nparam"])
    cs = np.array([float(sum(v.double().sum() for v in sd.values())), float(sum(v.double().abs().sum() for v in sd.values()))])
    np.testing.assert_allclose(cs, z["checksum"], rtol=1e-12)          # identical seeded initialisation
    check(net, z, torch.device("cpu"))


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(SPECS))
def test_network_matches_reference_gpu(name):
    z = np.load(os.path.join(GOLDEN, f"net_{name}.npz"))
    check(build(name), z, torch.device("cuda:0"))
