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


GRAD_SPECS = ["unet_2_1_n64", "ae_1_1_d2_n32", "goodgen3d_1_1_n32"]


def _param_stats(grads):
    """Same statistics as tools/gen_golden.py:param_stats."""
    out = []
    for i, g in enumerate(grads):
        gd = g.detach().double().reshape(-1).cpu().numpy()
        sign = np.random.default_rng(1000 + i).integers(0, 2, gd.size) * 2.0 - 1.0
        out.append([gd.sum(), np.sqrt((gd * gd).sum()), (gd * sign).sum()])
    return np.array(out)


def _check_stats(got, ref, names, rel):
    """Every parameter's gradient: norm to `rel`, sum and random projection to `rel` x norm x a factor for the number of terms."""
    # gradients that are zero in exact arithmetic (a bias in front of an InstanceNorm: the AE's convolutions) are rounding noise of the
    # order 1e-7 of the network's largest gradient in both implementations: compared against that floor, not against each other
    floor = 1e-5 * float(ref[:, 1].max())
    for k, name in enumerate(names):
        norm = max(ref[k, 1], floor)
        assert abs(got[k, 1] - ref[k, 1]) <= rel * norm + 1e-12, f"{name}: |grad| {got[k, 1]} vs {ref[k, 1]}"
        for c in (0, 2):
            assert abs(got[k, c] - ref[k, c]) <= 4.0 * rel * norm + 1e-12, f"{name}: statistic {c}: {got[k, c]} vs {ref[k, c]} (|grad| {norm})"


def check_all_param_grads(name, dev):
    """Gradient wrt EVERY parameter against the reference's (eval mode), through per-parameter statistics (netgrads_*.npz)."""
    z = np.load(os.path.join(GOLDEN, f"net_{name}.npz"))
    zg = np.load(os.path.join(GOLDEN, f"netgrads_{name}.npz"))
    net = build(name).to(dev)
    assert [k for k, _ in net.named_parameters()] == list(zg["names"])
    x = torch.from_numpy(z["x"]).to(dev)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        y = net(x)
    grads = torch.autograd.grad(y, list(net.parameters()), torch.from_numpy(z["cot"]).to(dev))
    _check_stats(_param_stats(grads), zg["stats"], list(zg["names"]), 2e-3 if dev.type == "cuda" else 2e-4)


@pytest.mark.parametrize("name", GRAD_SPECS)
def test_every_parameter_gradient_matches_reference_cpu(name):
    check_all_param_grads(name, torch.device("cpu"))


@pytest.mark.parametrize("name", GRAD_SPECS)
def test_train_mode_matches_reference_cpu(name):
    """Dropout live (the reference trains with Dropout 0.5 in four U-Net blocks, unets.py:13-45): with the CPU generator seeded as in the
    fixture the rebuilt network must draw the same masks -- same Dropout modules, same places, same order -- and reproduce the
    reference's train-mode output and gradients."""
    z = np.load(os.path.join(GOLDEN, f"net_{name}.npz"))
    zg = np.load(os.path.join(GOLDEN, f"netgrads_{name}.npz"))
    net = build(name).train()
    x = torch.from_numpy(z["x"]).requires_grad_(True)
    torch.manual_seed(777)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        y = net(x)
    np.testing.assert_allclose(y.detach().numpy(), zg["y_train"], rtol=1e-5, atol=1e-6)
    g = torch.autograd.grad(y, [x] + list(net.parameters()), torch.from_numpy(z["cot"]))
    np.testing.assert_allclose(g[0].numpy(), zg["grad_x_train"], rtol=1e-3, atol=1e-4 * float(np.abs(zg["grad_x_train"]).max()))
    _check_stats(_param_stats(g[1:]), zg["stats_train"], list(zg["names"]), 2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("name", GRAD_SPECS)
def test_every_parameter_gradient_matches_reference_gpu(name):
    check_all_param_grads(name, torch.device("cuda:0"))


@pytest.mark.gpu
@pytest.mark.parametrize("name", GRAD_SPECS)
def test_train_mode_hip_blocks_equal_stock_modules_gpu(name):
    """Train mode on the GPU (Dropout live): the network on the hand-written HIP layers against the SAME network object on torch's own
    layers (fused.stock_layers(): the stock-module composition the reference is made of), same seed before each forward -> same
    Dropout masks (nn.Dropout is the stock module in both): output, input gradient and every parameter gradient."""
    from diffnet_amd.networks import fused
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(GOLDEN, f"net_{name}.npz"))
    net = build(name).to(dev).train()
    x = torch.from_numpy(z["x"]).to(dev).requires_grad_(True)
    cot = torch.from_numpy(z["cot"]).to(dev)

    def run():
        torch.manual_seed(4242)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            y = net(x)
        return y, torch.autograd.grad(y, [x] + list(net.parameters()), cot)

    y_hip, g_hip = run()
    with fused.stock_layers():
        y_ref, g_ref = run()
    assert float((y_hip - y_ref).abs().max()) > 0.0 or name.startswith("ae")      # two different code paths really ran
    np.testing.assert_allclose(y_hip.detach().cpu().numpy(), y_ref.detach().cpu().numpy(), rtol=2e-4, atol=2e-5)
    names = ["x"] + [k for k, _ in net.named_parameters()]
    floor = 1e-5 * max(float(b.abs().max()) for b in g_ref[1:])       # exactly-zero gradients (bias in front of InstanceNorm) are noise in both
    for n, a, b in zip(names, g_hip, g_ref):
        scale = max(float(b.abs().max()), floor)
        assert float((a - b).abs().max()) <= 2e-3 * scale, f"{name} {n}: {float((a - b).abs().max())} vs scale {scale}"
    # and the dropout really was live: a second seed gives another output
    torch.manual_seed(1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        y2 = net(x)
    if name != "ae_1_1_d2_n32":        # the AE has no Dropout (autoencoders.py:7-95)
        assert float((y2 - y_hip).abs().max()) > 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(SPECS))
def test_network_matches_reference_gpu(name):
    z = np.load(os.path.join(GOLDEN, f"net_{name}.npz"))
    check(build(name), z, torch.device("cuda:0"))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 5, 8, 8), (2, 4, 7, 9), (2, 3, 1, 2), (1, 2, 128, 128), (1, 3, 129, 129), (2, 3, 6, 5, 4),
                                   (1, 2, 33, 32, 32), (1, 1, 64, 64, 64)])
@pytest.mark.parametrize("slope", [0.2, 0.0, 1.0])
def test_instnorm_act_hip_matches_torch(shape, slope):
    """dn_instnorm_act_fwd/bwd (one fused kernel each way) against torch's InstanceNorm + LeakyReLU on the same device and
    against the float64 CPU evaluation; odd spatial sizes take the scalar path, multiples of 4 the float4 path."""
    import torch.nn.functional as F
    from diffnet_amd.networks.fused import InstanceNormAct
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(shape, generator=g) * 3 + 1.5)
    cot = torch.randn(shape, generator=g)
    xd = x.double().requires_grad_(True)
    yd = F.instance_norm(xd, eps=1e-5)
    yd = yd if slope == 1.0 else F.leaky_relu(yd, slope)
    gd, = torch.autograd.grad(yd, xd, cot.double())
    xg = x.to(dev).requires_grad_(True)
    y = InstanceNormAct(shape[1], slope=slope)(xg)
    gx, = torch.autograd.grad(y, xg, cot.to(dev))
    np.testing.assert_allclose(y.detach().cpu().numpy(), yd.detach().numpy(), rtol=2e-5, atol=2e-6)
    # points within rounding of the activation kink may flip sides between fp32 and fp64: compare away from them
    far = (yd.detach().abs() > 1e-4).numpy() | (slope == 1.0)
    scale = float(gd.abs().max())
    assert np.abs(gx.cpu().numpy() - gd.numpy())[far].max() <= 2e-5 * scale + 1e-6
    # bitwise repeatable
    y2 = InstanceNormAct(shape[1], slope=slope)(xg)
    assert torch.equal(y, y2)


@pytest.mark.gpu
def test_instnorm_act_rejects_single_element():
    from diffnet_amd.networks.fused import InstanceNormAct
    with pytest.raises(ValueError):
        InstanceNormAct(4, slope=0.2)(torch.zeros(2, 4, 1, 1, device="cuda"))


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,h,w,cout", [(2, 64, 16, 16, 1), (1, 5, 7, 9, 1), (2, 64, 32, 24, 2), (1, 70, 12, 66, 1), (3, 8, 1, 1, 1)])
@pytest.mark.parametrize("sigmoid", [True, False])
def test_fused_output_block_matches_torch_modules(B, C, h, w, cout, sigmoid):
    """dn_upconv_out_fwd/bwd against Upsample -> ZeroPad2d((1,0,1,0)) -> Conv2d(4x4, padding 1) -> Sigmoid evaluated in
    float64 on the CPU (the reference's `UNet.final`, unets.py:68-74): output, input gradient, weight and bias gradients."""
    from torch import nn
    from diffnet_amd.networks.fused import upsample_pad_conv4
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, C, h, w, generator=g)
    conv = nn.Conv2d(C, cout, 4, padding=1)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.2)
        conv.bias.copy_(torch.randn(conv.bias.shape, generator=g))
    cot = torch.randn(B, cout, 2 * h, 2 * w, generator=g)
    ref = nn.Sequential(nn.Upsample(scale_factor=2), nn.ZeroPad2d((1, 0, 1, 0)), conv, *([nn.Sigmoid()] if sigmoid else [])).double()
    xd = x.double().requires_grad_(True)
    yd = ref(xd)
    gxd, gwd, gbd = torch.autograd.grad(yd, [xd, ref[2].weight, ref[2].bias], cot.double())
    xg = x.to(dev).requires_grad_(True)
    wg = conv.weight.detach().float().to(dev).requires_grad_(True)
    bg = conv.bias.detach().float().to(dev).requires_grad_(True)
    y = upsample_pad_conv4(xg, wg, bg, sigmoid=sigmoid)
    gx, gw, gb = torch.autograd.grad(y, [xg, wg, bg], cot.to(dev))

    def close(a, b, tol):
        a, b = a.detach().cpu().double().numpy(), b.detach().numpy()
        assert np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-30), (np.abs(a - b).max(), np.abs(b).max())

    close(y, yd, 2e-6)
    close(gx, gxd, 5e-6)
    close(gw, gwd, 2e-5)
    close(gb, gbd, 2e-5)
    y2 = upsample_pad_conv4(xg, wg, bg, sigmoid=sigmoid)
    gw2, = torch.autograd.grad(y2, [wg], cot.to(dev))
    assert torch.equal(y, y2) and torch.equal(gw, gw2)            # fixed-order reductions


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,d,h,w,cout", [(2, 32, 6, 5, 8, 1), (1, 3, 4, 7, 9, 2), (1, 40, 3, 4, 66, 1), (2, 4, 1, 1, 1, 1), (1, 32, 16, 16, 16, 1)])
@pytest.mark.parametrize("sigmoid", [True, False])
def test_fused_output_block_3d_matches_torch_modules(B, C, d, h, w, cout, sigmoid):
    """dn_upconv3d_out_fwd/bwd against Upsample -> Conv3d(3^3, padding 1) -> Sigmoid in float64 on the CPU (the reference's
    `GoodGenerator.final`, wgan3d.py:88-92): output, input gradient, weight and bias gradients."""
    from torch import nn
    from diffnet_amd.networks.fused import upsample_conv3
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(13)
    x = torch.randn(B, C, d, h, w, generator=g)
    conv = nn.Conv3d(C, cout, 3, padding=1)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.2)
        conv.bias.copy_(torch.randn(conv.bias.shape, generator=g))
    cot = torch.randn(B, cout, 2 * d, 2 * h, 2 * w, generator=g)
    ref = nn.Sequential(nn.Upsample(scale_factor=2), conv, *([nn.Sigmoid()] if sigmoid else [])).double()
    xd = x.double().requires_grad_(True)
    yd = ref(xd)
    gxd, gwd, gbd = torch.autograd.grad(yd, [xd, ref[1].weight, ref[1].bias], cot.double())
    xg = x.to(dev).requires_grad_(True)
    wg = conv.weight.detach().float().to(dev).requires_grad_(True)
    bg = conv.bias.detach().float().to(dev).requires_grad_(True)
    y = upsample_conv3(xg, wg, bg, sigmoid=sigmoid)
    gx, gw, gb = torch.autograd.grad(y, [xg, wg, bg], cot.to(dev))

    def close(a, b, tol):
        a, b = a.detach().cpu().double().numpy(), b.detach().numpy()
        assert np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-30), (np.abs(a - b).max(), np.abs(b).max())

    close(y, yd, 2e-6)
    close(gx, gxd, 5e-6)
    close(gw, gwd, 2e-5)
    close(gb, gbd, 2e-5)
    y2 = upsample_conv3(xg, wg, bg, sigmoid=sigmoid)
    gw2, = torch.autograd.grad(y2, [wg], cot.to(dev))
    assert torch.equal(y, y2) and torch.equal(gw, gw2)


@pytest.mark.gpu
@pytest.mark.parametrize("B,cin,cout,n", [(1, 1, 16, 16), (2, 16, 32, 8), (1, 3, 5, 6), (2, 64, 128, 4), (1, 20, 40, 10)])
@pytest.mark.parametrize("transposed", [False, True])
def test_conv3d_k4s2_weight_gradient_matches_float64(B, cin, cout, n, transposed):
    """dn_conv3d_k4s2_wrw (through Conv3dS2 / ConvTranspose3dS2) against the float64 CPU gradient of the stock torch
    layer; forward output and input gradient (MIOpen) are checked alongside."""
    from torch import nn
    from diffnet_amd.networks.fused import Conv3dS2, ConvTranspose3dS2
    dev = torch.device("cuda", 0)
    torch.manual_seed(17)
    if transposed:
        ref, mod = nn.ConvTranspose3d(cin, cout, 4, 2, 1, bias=False), ConvTranspose3dS2(cin, cout, 4, 2, 1, bias=False)
        shape = (B, cin, n // 2, n // 2, n // 2) if n >= 4 else (B, cin, n, n, n)
    else:
        ref, mod = nn.Conv3d(cin, cout, 4, 2, 1, bias=False), Conv3dS2(cin, cout, 4, 2, 1, bias=False)
        shape = (B, cin, n, n, n)
    mod.load_state_dict(ref.state_dict())
    x = torch.randn(shape)
    xd = x.double().requires_grad_(True)
    refd = ref.double()
    yd = refd(xd)
    cot = torch.randn(yd.shape)
    gxd, gwd = torch.autograd.grad(yd, [xd, refd.weight], cot.double())
    mod = mod.to(dev)
    xg = x.to(dev).requires_grad_(True)
    y = mod(xg)
    gx, gw = torch.autograd.grad(y, [xg, mod.weight], cot.to(dev))

    def close(a, b, tol):
        a, b = a.detach().cpu().double().numpy(), b.detach().numpy()
        assert a.shape == b.shape and np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-30), (np.abs(a - b).max(), np.abs(b).max())

    close(y, yd, 1e-4)
    close(gx, gxd, 1e-4)
    close(gw, gwd, 2e-5)
    gw2, = torch.autograd.grad(mod(xg), [mod.weight], cot.to(dev))
    assert torch.equal(gw, gw2)


@pytest.mark.gpu
def test_instnorm_act_backward_reads_channel_slices_in_place():
    """The gradient reaching an up-block's InstanceNorm is a channel slice of the skip concatenation's gradient
    (contiguous per sample, wider batch stride): same result as with a contiguous copy, for the one-pass and the
    sliced (few large instances) kernels."""
    from diffnet_amd.networks.fused import InstanceNormAct
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(23)
    for shape in [(3, 5, 12, 12), (2, 4, 128, 128)]:
        B, Cn, H, W = shape
        x = torch.randn(shape, generator=g).to(dev).requires_grad_(True)
        wide = torch.randn((B, Cn + 3, H, W), generator=g).to(dev)
        y = InstanceNormAct(Cn, slope=0.0)(x)
        g_slice, = torch.autograd.grad(y, x, wide[:, :Cn], retain_graph=True)
        g_copy, = torch.autograd.grad(y, x, wide[:, :Cn].contiguous())
        assert not wide[:, :Cn].is_contiguous() and torch.equal(g_slice, g_copy)


@pytest.mark.gpu
def test_unet_at_the_baseline_mesh_512_matches_reference():
    """BASELINE configs[1]'s network at its own size: U-Net(2 -> 1) on a 512 x 512 sample (eval mode), output and gradients
    against the imported reference's CPU result (fixture keeps every 8th row / column; tools/gen_golden.py --net512)."""
    z = np.load(os.path.join(GOLDEN, "net_unet_2_1_n512.npz"))
    dev = torch.device("cuda:0")
    net = build("unet_2_1_n64").to(dev)
    sd = net.state_dict()
    cs = np.array([float(sum(v.double().sum() for v in sd.values())), float(sum(v.double().abs().sum() for v in sd.values()))])
    np.testing.assert_allclose(cs, z["checksum"], rtol=1e-12)
    n, st = int(z["n"]), int(z["stride"])
    yy, xx = torch.meshgrid(torch.linspace(0, 1, n), torch.linspace(0, 1, n), indexing="ij")
    x = torch.stack([0.5 + 0.4 * torch.sin(7 * xx + 3 * yy), (torch.cos(5 * xx * yy) > 0.3).float()], 0)[None].to(dev).requires_grad_(True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        y = net(x)
    cot = torch.cos(11 * xx - 4 * yy)[None, None].to(dev)
    gx, = torch.autograd.grad(y, x, cot, retain_graph=True)
    gw, = torch.autograd.grad(y, list(net.parameters())[0], cot)
    np.testing.assert_allclose(y.detach().cpu().numpy()[..., ::st, ::st], z["y"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(float(y.double().sum()), float(z["y_sum"]), rtol=1e-5)
    gtol = 2e-3
    # every convolution of this network is the repo's own fp32-MFMA kernel now (csrc/conv2d_k4s2.hip, upconv_out.hip).  The input
    # gradient of nine normalised conv levels at 262 144 pixels is sensitive to summation order (the oneDNN reference is not exact
    # either) and to ReLU / LeakyReLU kinks (a pre-activation within rounding of zero takes the other branch): median error < 1e-5
    # of the gradient's scale, 99 % of the samples within 1e-3, all within 2e-2
    ex = np.abs(gx.cpu().numpy()[..., ::st, ::st] - z["grad_x"]) / float(np.abs(z["grad_x"]).max())
    assert np.median(ex) <= 1e-5 and np.quantile(ex, 0.99) <= 1e-3 and ex.max() <= 2e-2, (np.median(ex), np.quantile(ex, 0.99), ex.max())
    ew = np.abs(gw.cpu().numpy() - z["grad_w0"]) / float(np.abs(z["grad_w0"]).max())
    assert ew.max() <= 5e-4, ew.max()
    np.testing.assert_allclose(float(gx.double().abs().sum()), float(z["gx_abs_sum"]), rtol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,M,H,W", [(2, 2, 32, 16, 16), (1, 32, 64, 8, 8), (2, 5, 3, 3, 5), (1, 64, 128, 4, 4), (3, 130, 70, 5, 3),
                                       (1, 256, 256, 2, 2), (2, 1, 1, 1, 1), (1, 16, 40, 33, 17)])
def test_conv2d_k4s2_family_matches_float64_reference(B, C, M, H, W):
    """dn_conv2d_k4s2_down / _up / _wrw (fp32 MFMA implicit GEMMs) against torch's Conv2d / ConvTranspose2d evaluated in float64 on
    the CPU: forward, input gradient and weight gradient of both layers, ragged channel / position counts included.  fp32 products
    with fp32 accumulation: 2e-5 of the result's scale."""
    import torch.nn.functional as F
    from diffnet_amd.networks.fused import Conv2dS2, ConvTranspose2dS2
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(11)
    fine = torch.randn((B, C, 2 * H, 2 * W), generator=g)
    coarse = torch.randn((B, M, H, W), generator=g)
    w = torch.randn((M, C, 4, 4), generator=g) * 0.1

    def check(got, ref, what):
        scale = float(ref.abs().max()) + 1e-30
        err = float((got.cpu().double() - ref).abs().max())
        assert err <= 2e-5 * scale, (what, err, scale)

    # Conv2d: fine -> coarse
    conv = Conv2dS2(C, M, 4, 2, 1, bias=False).to(dev)
    with torch.no_grad():
        conv.weight.copy_(w)
    xg = fine.to(dev).requires_grad_(True)
    y = conv(xg)
    gx, gw = torch.autograd.grad(y, (xg, conv.weight), coarse.to(dev))
    xd, wd = fine.double().requires_grad_(True), w.double().requires_grad_(True)
    yd = F.conv2d(xd, wd, None, 2, 1)
    gxd, gwd = torch.autograd.grad(yd, (xd, wd), coarse.double())
    check(y.detach(), yd.detach(), "conv fwd")
    check(gx, gxd, "conv dgrad")
    check(gw, gwd, "conv wgrad")
    # ConvTranspose2d: coarse -> fine (weight (cin = M, cout = C, 4, 4))
    convt = ConvTranspose2dS2(M, C, 4, 2, 1, bias=False).to(dev)
    with torch.no_grad():
        convt.weight.copy_(w)
    cg = coarse.to(dev).requires_grad_(True)
    z = convt(cg)
    gc, gwt = torch.autograd.grad(z, (cg, convt.weight), fine.to(dev))
    cd, wtd = coarse.double().requires_grad_(True), w.double().requires_grad_(True)
    zd = F.conv_transpose2d(cd, wtd, None, 2, 1)
    gcd, gwtd = torch.autograd.grad(zd, (cd, wtd), fine.double())
    check(z.detach(), zd.detach(), "convT fwd")
    check(gc, gcd, "convT dgrad")
    check(gwt, gwtd, "convT wgrad")
    # bitwise repeatable (fixed-order partial sums)
    gw2 = torch.autograd.grad(conv(xg), conv.weight, coarse.to(dev))[0]
    assert torch.equal(gw, gw2)


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,M,D,H,W", [(1, 1, 16, 8, 8, 8), (2, 16, 32, 4, 4, 4), (1, 5, 3, 3, 2, 5), (1, 64, 128, 2, 2, 2), (2, 33, 70, 2, 3, 2),
                                         (1, 128, 128, 1, 1, 1), (2, 64, 256, 2, 2, 2)])
def test_conv3d_k4s2_family_matches_float64_reference(B, C, M, D, H, W):
    """dn_conv3d_k4s2_down / _up (+ the existing _wrw) against Conv3d / ConvTranspose3d in float64 on the CPU: forward, input
    gradient, weight gradient of both layers of the 3-D generator."""
    import torch.nn.functional as F
    from diffnet_amd.networks.fused import Conv3dS2, ConvTranspose3dS2
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(12)
    fine = torch.randn((B, C, 2 * D, 2 * H, 2 * W), generator=g)
    coarse = torch.randn((B, M, D, H, W), generator=g)
    w = torch.randn((M, C, 4, 4, 4), generator=g) * 0.05

    def check(got, ref, what):
        scale = float(ref.abs().max()) + 1e-30
        err = float((got.cpu().double() - ref).abs().max())
        assert err <= 2e-5 * scale, (what, err, scale)

    conv = Conv3dS2(C, M, 4, 2, 1, bias=False).to(dev)
    with torch.no_grad():
        conv.weight.copy_(w)
    xg = fine.to(dev).requires_grad_(True)
    y = conv(xg)
    gx, gw = torch.autograd.grad(y, (xg, conv.weight), coarse.to(dev))
    xd, wd = fine.double().requires_grad_(True), w.double().requires_grad_(True)
    yd = F.conv3d(xd, wd, None, 2, 1)
    gxd, gwd = torch.autograd.grad(yd, (xd, wd), coarse.double())
    check(y.detach(), yd.detach(), "conv3d fwd")
    check(gx, gxd, "conv3d dgrad")
    check(gw, gwd, "conv3d wgrad")
    convt = ConvTranspose3dS2(M, C, 4, 2, 1, bias=False).to(dev)
    with torch.no_grad():
        convt.weight.copy_(w)
    cg = coarse.to(dev).requires_grad_(True)
    z = convt(cg)
    gc, gwt = torch.autograd.grad(z, (cg, convt.weight), fine.to(dev))
    cd, wtd = coarse.double().requires_grad_(True), w.double().requires_grad_(True)
    zd = F.conv_transpose3d(cd, wtd, None, 2, 1)
    gcd, gwtd = torch.autograd.grad(zd, (cd, wtd), fine.double())
    check(z.detach(), zd.detach(), "convT3d fwd")
    check(gc, gcd, "convT3d dgrad")
    check(gwt, gwtd, "convT3d wgrad")


@pytest.mark.gpu
@pytest.mark.parametrize("B,Ci,Co,H,W,K", [(2, 1, 128, 38, 38, 7), (2, 128, 1, 40, 40, 3), (1, 1, 1, 38, 38, 7), (3, 3, 5, 9, 11, 5), (1, 2, 2, 7, 7, 7)])
def test_conv2d_valid_matches_float64_reference(B, Ci, Co, H, W, K):
    """dn_conv2d_valid_* (the AE's stride-1 stem / head convolutions, with bias) against float64 torch on the CPU."""
    import torch.nn.functional as F
    from diffnet_amd.networks.fused import Conv2dValid
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(13)
    x = torch.randn((B, Ci, H, W), generator=g)
    w = torch.randn((Co, Ci, K, K), generator=g) * 0.1
    bias = torch.randn((Co,), generator=g)
    cot = torch.randn((B, Co, H - K + 1, W - K + 1), generator=g)
    conv = Conv2dValid(Ci, Co, K).to(dev)
    with torch.no_grad():
        conv.weight.copy_(w)
        conv.bias.copy_(bias)
    xg = x.to(dev).requires_grad_(True)
    y = conv(xg)
    gx, gw, gb = torch.autograd.grad(y, (xg, conv.weight, conv.bias), cot.to(dev))
    xd, wd, bd = x.double().requires_grad_(True), w.double().requires_grad_(True), bias.double().requires_grad_(True)
    yd = F.conv2d(xd, wd, bd)
    gxd, gwd, gbd = torch.autograd.grad(yd, (xd, wd, bd), cot.double())
    for got, ref, what in ((y.detach(), yd.detach(), "fwd"), (gx, gxd, "dgrad"), (gw, gwd, "wgrad"), (gb, gbd, "bias grad")):
        scale = float(ref.abs().max()) + 1e-30
        assert float((got.cpu().double() - ref).abs().max()) <= 2e-5 * scale, what


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,M,D,H,W", [(1, 128, 128, 4, 4, 4), (1, 64, 128, 8, 8, 8), (2, 33, 70, 2, 3, 2), (1, 256, 64, 8, 8, 8), (1, 16, 32, 16, 16, 16), (1, 9, 5, 2, 2, 2)])
def test_conv3d_split_contraction_equals_unsplit(B, C, M, D, H, W):
    """dn_conv3d_k4s2_down_ws / _up_ws (round 4: the contraction of the deep, narrow layers split over workgroups, slices summed in order)
    against the unsplit launches (NULL workspace) through the C ABI: same numbers to fp32 summation order, bitwise repeatable, a workspace
    that is too small is refused."""
    import ctypes as C_
    from diffnet_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(3)
    fine = torch.randn((B, C, 2 * D, 2 * H, 2 * W), generator=g).to(dev)
    coarse = torch.randn((B, M, D, H, W), generator=g).to(dev)
    w = (torch.randn((M, C, 4, 4, 4), generator=g) * 0.05).to(dev)
    s = C_.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C_.c_void_p(t.data_ptr())
    for up in (0, 1):
        src, shape = (coarse, fine.shape) if up else (fine, coarse.shape)
        fn, fn_ws = (L.dn_conv3d_k4s2_up, L.dn_conv3d_k4s2_up_ws) if up else (L.dn_conv3d_k4s2_down, L.dn_conv3d_k4s2_down_ws)
        ref = torch.empty(shape, device=dev)
        assert fn(p(src), p(w), p(ref), B, C, M, D, H, W, s) == 0
        nbytes = L.dn_conv3d_k4s2_workspace_bytes(up, B, C, M, D, H, W)
        assert nbytes >= 0
        got, got2 = torch.empty(shape, device=dev), torch.empty(shape, device=dev)
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        assert fn_ws(p(src), p(w), p(got), B, C, M, D, H, W, p(ws), nbytes, s) == 0
        assert fn_ws(p(src), p(w), p(got2), B, C, M, D, H, W, p(ws), nbytes, s) == 0
        assert torch.equal(got, got2)
        scale = float(ref.abs().max())
        assert float((got - ref).abs().max()) <= 1e-5 * scale, ("up" if up else "down", nbytes)          # (up to 16384 products per output: summation order)
        if nbytes:
            assert fn_ws(p(src), p(w), p(got), B, C, M, D, H, W, p(ws), nbytes - 4, s) == -3          # DN_E_WORKSPACE
