"""Network building blocks on the HIP kernels of libdiffnet_hip.so: every convolution of the three
reference networks -- Conv2d / ConvTranspose2d 4x4 s2 (csrc/conv2d_k4s2.hip), Conv3d / ConvTranspose3d 4^3 s2 (conv3d_k4s2.hip,
conv3d_wrw.hip), the stride-1 stem / head convolutions of the auto-encoder (conv2d_direct.hip), the fused Upsample -> Conv -> Sigmoid
output blocks (upconv_out.hip, upconv3d_out.hip) -- forward, input gradient and weight gradient, plus InstanceNorm + activation
(instnorm_act.hip).  The modules subclass torch's (same parameters, initialisation and state_dict keys); on CPU tensors they fall
through to torch's implementation (parity tests against the reference's golden vectors run there too)."""
import torch
from torch.autograd.function import once_differentiable
from torch import nn
import torch.nn.functional as F

from .. import _lib
from ..ops import _p, _stream


# Test switch: with HIP_LAYERS False every module below takes torch's own kernels (MIOpen on the GPU) -- the stock-module composition
# the reference networks are made of.  `with stock_layers(): ...` is how the GPU tests run the SAME network object both ways.
HIP_LAYERS = True


def _hip(x):
    return HIP_LAYERS and x.is_cuda and x.dtype == torch.float32


class stock_layers:
    def __enter__(self):
        global HIP_LAYERS
        self.prev, HIP_LAYERS = HIP_LAYERS, False

    def __exit__(self, *exc):
        global HIP_LAYERS
        HIP_LAYERS = self.prev


def _workspace(x, n_inst, S):
    """fp64 partial-sum scratch for the sliced path (few, large instances); None when the C side needs none."""
    nbytes = _lib.lib().dn_instnorm_workspace_bytes(n_inst, S)
    if nbytes <= 0:
        return None, 0
    return torch.empty(nbytes, dtype=torch.uint8, device=x.device), nbytes


class _InstNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps, slope):
        x = x.contiguous()
        n_inst = x.shape[0] * x.shape[1]
        S = x[0, 0].numel()
        y = torch.empty_like(x)
        mean = torch.empty(n_inst, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        ws, wsb = _workspace(x, n_inst, S)
        rc = _lib.lib().dn_instnorm_act_fwd(_p(x), _p(y), _p(mean), _p(rstd), n_inst, S, eps, slope, _p(ws), wsb, _stream(x))
        _lib.check(rc, "dn_instnorm_act_fwd")
        ctx.save_for_backward(x, mean, rstd)
        ctx.slope = slope
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, mean, rstd = ctx.saved_tensors
        Cn, S = x.shape[1], x[0, 0].numel()
        # the gradient of a skip concatenation arrives as a channel slice of the wider tensor: contiguous per sample, with a
        # larger batch stride -- read in place instead of copied
        inner = x.stride()[1:]
        if gy.stride()[1:] == inner and gy.stride(0) >= Cn * S:
            gbs = gy.stride(0)
        else:
            gy, gbs = gy.contiguous(), 0
        gx = torch.empty_like(x)
        ws, wsb = _workspace(x, mean.numel(), S)
        rc = _lib.lib().dn_instnorm_act_bwd(_p(x), _p(mean), _p(rstd), _p(gy), _p(gx), mean.numel(), S, ctx.slope, Cn, gbs,
                                            _p(ws), wsb, _stream(x))
        _lib.check(rc, "dn_instnorm_act_bwd")
        return gx, None, None


class InstanceNormAct(nn.Module):
    """InstanceNorm{2,3}d(affine=False, track_running_stats=False, eps) followed by LeakyReLU(slope) / ReLU (slope 0) /
    nothing (slope 1).  `num_features` is accepted and ignored, like torch does for affine=False."""

    def __init__(self, num_features=None, slope=0.0, eps=1e-5):
        super().__init__()
        self.num_features, self.slope, self.eps = num_features, float(slope), eps

    def forward(self, x):
        if x[0, 0].numel() <= 1:
            raise ValueError(f"Expected more than 1 spatial element when training, got input size {x.size()}")   # torch's message
        if _hip(x):
            return _InstNormAct.apply(x, self.eps, self.slope)
        y = F.instance_norm(x, eps=self.eps)
        return y if self.slope == 1.0 else F.leaky_relu(y, self.slope)

    def extra_repr(self):
        return f"slope={self.slope}, eps={self.eps}"


class _UpConvOut(torch.autograd.Function):
    """Upsample(x2) -> ZeroPad2d((1,0,1,0)) -> Conv2d(C -> 1, 4x4, padding 1) -> optional Sigmoid, one output channel."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        x, weight = x.contiguous(), weight.contiguous()
        B, Cn, h, w = x.shape
        out = torch.empty((B, 1, 2 * h, 2 * w), dtype=torch.float32, device=x.device)
        nbytes = _lib.lib().dn_upconv_out_workspace_bytes(B, Cn, h, w)
        if nbytes < 0:
            _lib.check(int(nbytes), "dn_upconv_out_workspace_bytes")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        rc = _lib.lib().dn_upconv_out_fwd(_p(x), _p(weight), _p(bias), _p(out), B, Cn, h, w, int(act), _p(ws), nbytes, _stream(x))
        _lib.check(rc, "dn_upconv_out_fwd")
        ctx.save_for_backward(x, weight, out)
        ctx.act, ctx.has_bias = int(act), bias is not None
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        x, weight, out = ctx.saved_tensors
        gout = gout.contiguous()
        B, Cn, h, w = x.shape
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        gx = torch.empty_like(x) if need_x else None
        gw = torch.empty_like(weight) if need_w else None
        gb = torch.empty(1, dtype=torch.float32, device=x.device) if (need_w and ctx.has_bias) else None
        nbytes = _lib.lib().dn_upconv_out_workspace_bytes(B, Cn, h, w)
        if nbytes < 0:
            _lib.check(int(nbytes), "dn_upconv_out_workspace_bytes")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        rc = _lib.lib().dn_upconv_out_bwd(_p(x), _p(weight), _p(out), _p(gout), _p(gx), _p(gw), _p(gb), B, Cn, h, w, ctx.act, _p(ws), nbytes,
                                          _stream(x))
        _lib.check(rc, "dn_upconv_out_bwd")
        return gx, gw, gb, None


def upsample_pad_conv4(x, weight, bias=None, sigmoid=True):
    """The U-Net's output block on the HIP kernels: x (B,C,h,w), weight (Cout,C,4,4), bias (Cout) -> (B,Cout,2h,2w).
    One launch pair per output channel (the reference networks use Cout = 1)."""
    if not (_hip(x) and weight.shape[-2:] == (4, 4) and x.dim() == 4):
        raise _lib.DiffNetHipError("upsample_pad_conv4: float32 CUDA tensors (B,C,h,w) and 4x4 weights only")
    outs = [_UpConvOut.apply(x, weight[co:co + 1], None if bias is None else bias[co:co + 1], sigmoid) for co in range(weight.shape[0])]
    return outs[0] if len(outs) == 1 else torch.cat(outs, 1)


class _UpConv3dOut(torch.autograd.Function):
    """Upsample(x2) -> Conv3d(C -> 1, 3x3x3, padding 1) -> optional Sigmoid, one output channel."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        x, weight = x.contiguous(), weight.contiguous()
        B, Cn, d, h, w = x.shape
        out = torch.empty((B, 1, 2 * d, 2 * h, 2 * w), dtype=torch.float32, device=x.device)
        nbytes = _lib.lib().dn_upconv3d_out_workspace_bytes(B, Cn, d, h, w)
        if nbytes < 0:
            _lib.check(int(nbytes), "dn_upconv3d_out_workspace_bytes")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        rc = _lib.lib().dn_upconv3d_out_fwd(_p(x), _p(weight), _p(bias), _p(out), B, Cn, d, h, w, int(act), _p(ws), nbytes, _stream(x))
        _lib.check(rc, "dn_upconv3d_out_fwd")
        ctx.save_for_backward(x, weight, out)
        ctx.act, ctx.has_bias = int(act), bias is not None
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        x, weight, out = ctx.saved_tensors
        gout = gout.contiguous()
        B, Cn, d, h, w = x.shape
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        gx = torch.empty_like(x) if need_x else None
        gw = torch.empty_like(weight) if need_w else None
        gb = torch.empty(1, dtype=torch.float32, device=x.device) if (need_w and ctx.has_bias) else None
        nbytes = _lib.lib().dn_upconv3d_out_workspace_bytes(B, Cn, d, h, w)
        if nbytes < 0:
            _lib.check(int(nbytes), "dn_upconv3d_out_workspace_bytes")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        rc = _lib.lib().dn_upconv3d_out_bwd(_p(x), _p(weight), _p(out), _p(gout), _p(gx), _p(gw), _p(gb), B, Cn, d, h, w, ctx.act, _p(ws),
                                            nbytes, _stream(x))
        _lib.check(rc, "dn_upconv3d_out_bwd")
        return gx, gw, gb, None


def upsample_conv3(x, weight, bias=None, sigmoid=True):
    """The 3-D generator's output block on the HIP kernels: x (B,C,d,h,w), weight (Cout,C,3,3,3), bias (Cout) ->
    (B,Cout,2d,2h,2w); one launch set per output channel."""
    if not (_hip(x) and tuple(weight.shape[-3:]) == (3, 3, 3) and x.dim() == 5):
        raise _lib.DiffNetHipError("upsample_conv3: float32 CUDA tensors (B,C,d,h,w) and 3x3x3 weights only")
    outs = [_UpConv3dOut.apply(x, weight[co:co + 1], None if bias is None else bias[co:co + 1], sigmoid) for co in range(weight.shape[0])]
    return outs[0] if len(outs) == 1 else torch.cat(outs, 1)


def _wrw3d(fine, coarse):
    """grad_weight (M, CN, 4,4,4) of a 4^3 / stride 2 / padding 1 (transposed) convolution: dn_conv3d_k4s2_wrw."""
    fine, coarse = fine.contiguous(), coarse.contiguous()
    B, CN = fine.shape[:2]
    M, (d, h, w) = coarse.shape[1], coarse.shape[2:]
    if M > 128:        # the kernel tiles at most 128 coarse channels per launch: wider layers (ConvTranspose3d 256 -> 64) go in channel blocks
        return torch.cat([_wrw3d(fine, coarse[:, m0:m0 + 128]) for m0 in range(0, M, 128)], 0)
    gw = torch.empty((M, CN, 4, 4, 4), dtype=torch.float32, device=fine.device)
    nbytes = _lib.lib().dn_conv3d_k4s2_wrw_workspace_bytes(B, CN, M, d, h, w)
    if nbytes < 0:
        _lib.check(int(nbytes), "dn_conv3d_k4s2_wrw_workspace_bytes")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=fine.device) if nbytes else None
    rc = _lib.lib().dn_conv3d_k4s2_wrw(_p(fine), _p(coarse), _p(gw), B, CN, M, d, h, w, _p(ws), nbytes, _stream(fine))
    _lib.check(rc, "dn_conv3d_k4s2_wrw")
    return gw


def _c3_workspace(up, B, Cn, M, d, h, wd, device):
    """Workspace of the split contraction (dn_conv3d_k4s2_workspace_bytes; None where the library does not split): a fresh block from the caching
    allocator, ordered on the launch stream like the output."""
    nbytes = _lib.lib().dn_conv3d_k4s2_workspace_bytes(up, B, Cn, M, d, h, wd)
    if nbytes < 0:
        _lib.check(int(nbytes), "dn_conv3d_k4s2_workspace_bytes")
    return (torch.empty(nbytes, dtype=torch.uint8, device=device) if nbytes else None), nbytes


def _c3_down(fine, w):
    """coarse = w (*)_s2 fine (dn_conv3d_k4s2_down): Conv3d forward / ConvTranspose3d input gradient."""
    fine, w = fine.contiguous(), w.contiguous()
    B, Cn = fine.shape[:2]
    d, h, wd = (s_ // 2 for s_ in fine.shape[2:])
    M = w.shape[0]
    out = torch.empty((B, M, d, h, wd), dtype=torch.float32, device=fine.device)
    ws, nbytes = _c3_workspace(0, B, Cn, M, d, h, wd, fine.device)
    rc = _lib.lib().dn_conv3d_k4s2_down_ws(_p(fine), _p(w), _p(out), B, Cn, M, d, h, wd, _p(ws), nbytes, _stream(fine))
    _lib.check(rc, "dn_conv3d_k4s2_down_ws")
    return out


def _c3_up(coarse, w):
    """fine = w (*)^T coarse (dn_conv3d_k4s2_up): ConvTranspose3d forward / Conv3d input gradient."""
    coarse, w = coarse.contiguous(), w.contiguous()
    B, M, d, h, wd = coarse.shape
    Cn = w.shape[1]
    out = torch.empty((B, Cn, 2 * d, 2 * h, 2 * wd), dtype=torch.float32, device=coarse.device)
    ws, nbytes = _c3_workspace(1, B, Cn, M, d, h, wd, coarse.device)
    rc = _lib.lib().dn_conv3d_k4s2_up_ws(_p(coarse), _p(w), _p(out), B, Cn, M, d, h, wd, _p(ws), nbytes, _stream(coarse))
    _lib.check(rc, "dn_conv3d_k4s2_up_ws")
    return out



class _Conv3dK4S2(torch.autograd.Function):
    """Conv3d(4^3, stride 2, padding 1, no bias): forward = `down`, input gradient = `up` (csrc/conv3d_k4s2.hip), weight gradient =
    dn_conv3d_k4s2_wrw -- all three on the fp32 matrix cores."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return _c3_down(x, weight)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _c3_up(gy, weight)
        if ctx.needs_input_grad[1]:
            gw = _wrw3d(fine=x, coarse=gy)
        return gx, gw


class _ConvT3dK4S2(torch.autograd.Function):
    """ConvTranspose3d(4^3, stride 2, padding 1, no bias): forward = `up`, input gradient = `down`, weight gradient = wrw."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return _c3_up(x, weight)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _c3_down(gy, weight)
        if ctx.needs_input_grad[1]:
            gw = _wrw3d(fine=gy, coarse=x)
        return gx, gw


def _k4s2(m, x, coarse_channels):
    return (_hip(x) and m.bias is None and tuple(m.kernel_size) == (4, 4, 4) and tuple(m.stride) == (2, 2, 2)
            and tuple(m.padding) == (1, 1, 1) and tuple(m.dilation) == (1, 1, 1) and m.groups == 1
            and tuple(getattr(m, "output_padding", (0, 0, 0))) == (0, 0, 0))


class Conv3dS2(nn.Conv3d):
    """nn.Conv3d whose weight gradient runs on dn_conv3d_k4s2_wrw when the layer is 4^3 / stride 2 / padding 1 / no bias on
    even-sized float32 GPU inputs (every UNetDown of the 3-D generator); anything else takes torch's path."""

    def forward(self, x):
        if _k4s2(self, x, self.out_channels) and all(s % 2 == 0 for s in x.shape[2:]):
            return _Conv3dK4S2.apply(x, self.weight)
        return super().forward(x)


class ConvTranspose3dS2(nn.ConvTranspose3d):
    """nn.ConvTranspose3d counterpart (every UNetUp of the 3-D generator)."""

    def forward(self, x, output_size=None):
        if output_size is None and _k4s2(self, x, self.in_channels):
            return _ConvT3dK4S2.apply(x, self.weight)
        return super().forward(x, output_size)


# ---- 2-D 4 x 4 / stride 2 / padding 1 layers on the fp32 matrix cores (csrc/conv2d_k4s2.hip) ---------------------------------------
def _c2_down(fine, w):
    """coarse = w (*)_s2 fine: Conv2d forward with w = weight (cout, cin, 4, 4); ConvTranspose2d input gradient with w = weight (cin, cout, 4, 4)."""
    fine, w = fine.contiguous(), w.contiguous()
    B, Cn, H2, W2 = fine.shape
    M = w.shape[0]
    out = torch.empty((B, M, H2 // 2, W2 // 2), dtype=torch.float32, device=fine.device)
    rc = _lib.lib().dn_conv2d_k4s2_down(_p(fine), _p(w), _p(out), B, Cn, M, H2 // 2, W2 // 2, _stream(fine))
    _lib.check(rc, "dn_conv2d_k4s2_down")
    return out


def _c2_up(coarse, w):
    """fine = w (*)^T coarse: ConvTranspose2d forward / Conv2d input gradient, same weight layouts as `_c2_down`."""
    coarse, w = coarse.contiguous(), w.contiguous()
    B, M, H, W = coarse.shape
    Cn = w.shape[1]
    out = torch.empty((B, Cn, 2 * H, 2 * W), dtype=torch.float32, device=coarse.device)
    rc = _lib.lib().dn_conv2d_k4s2_up(_p(coarse), _p(w), _p(out), B, Cn, M, H, W, _stream(coarse))
    _lib.check(rc, "dn_conv2d_k4s2_up")
    return out


def _c2_wrw(fine, coarse):
    """grad_weight (M, C, 4, 4) of both layers."""
    fine, coarse = fine.contiguous(), coarse.contiguous()
    B, Cn = fine.shape[:2]
    M, H, W = coarse.shape[1:]
    gw = torch.empty((M, Cn, 4, 4), dtype=torch.float32, device=fine.device)
    nbytes = _lib.lib().dn_conv2d_k4s2_wrw_workspace_bytes(B, Cn, M, H, W)
    if nbytes < 0:
        _lib.check(int(nbytes), "dn_conv2d_k4s2_wrw_workspace_bytes")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=fine.device) if nbytes else None
    rc = _lib.lib().dn_conv2d_k4s2_wrw(_p(fine), _p(coarse), _p(gw), B, Cn, M, H, W, _p(ws), nbytes, _stream(fine))
    _lib.check(rc, "dn_conv2d_k4s2_wrw")
    return gw


class _Conv2dK4S2(torch.autograd.Function):
    """Conv2d(4 x 4, stride 2, padding 1, no bias): forward, input gradient and weight gradient are the `down`, `up` and `wrw`
    contractions of csrc/conv2d_k4s2.hip."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return _c2_down(x, weight)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = _c2_up(gy, weight) if ctx.needs_input_grad[0] else None
        gw = _c2_wrw(fine=x, coarse=gy) if ctx.needs_input_grad[1] else None
        return gx, gw


class _ConvT2dK4S2(torch.autograd.Function):
    """ConvTranspose2d(4 x 4, stride 2, padding 1, no bias): forward = `up`, input gradient = `down`, weight gradient = `wrw`."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return _c2_up(x, weight)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = _c2_down(gy, weight) if ctx.needs_input_grad[0] else None
        gw = _c2_wrw(fine=gy, coarse=x) if ctx.needs_input_grad[1] else None
        return gx, gw


def _k4s2_2d(m, x):
    return (_hip(x) and x.dim() == 4 and tuple(m.kernel_size) == (4, 4) and tuple(m.stride) == (2, 2)
            and tuple(m.padding) == (1, 1) and tuple(m.dilation) == (1, 1) and m.groups == 1 and m.padding_mode == "zeros"
            and tuple(getattr(m, "output_padding", (0, 0))) == (0, 0))


class Conv2dS2(nn.Conv2d):
    """nn.Conv2d that runs 4 x 4 / stride 2 / padding 1 layers on even-sized float32 GPU inputs through the HIP kernels (bias, where
    the layer has one, is added afterwards); anything else takes torch's path."""

    def forward(self, x):
        if _k4s2_2d(self, x) and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0:
            y = _Conv2dK4S2.apply(x, self.weight)
            return y if self.bias is None else y + self.bias.view(1, -1, 1, 1)
        return super().forward(x)


class ConvTranspose2dS2(nn.ConvTranspose2d):
    """nn.ConvTranspose2d counterpart."""

    def forward(self, x, output_size=None):
        if output_size is None and _k4s2_2d(self, x):
            y = _ConvT2dK4S2.apply(x, self.weight)
            return y if self.bias is None else y + self.bias.view(1, -1, 1, 1)
        return super().forward(x, output_size)


# ---- stride-1 "valid" k x k convolutions with bias (AE stem / head), csrc/conv2d_direct.hip ----------------------------------------
class _Conv2dValid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        x, weight = x.contiguous(), weight.contiguous()
        B, Ci, H, W = x.shape
        Co, K = weight.shape[0], weight.shape[-1]
        y = torch.empty((B, Co, H - K + 1, W - K + 1), dtype=torch.float32, device=x.device)
        rc = _lib.lib().dn_conv2d_valid_fwd(_p(x), _p(weight), _p(bias), _p(y), B, Ci, Co, H, W, K, _stream(x))
        _lib.check(rc, "dn_conv2d_valid_fwd")
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        B, Ci, H, W = x.shape
        Co, K = weight.shape[0], weight.shape[-1]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            rc = _lib.lib().dn_conv2d_valid_bwd_data(_p(gy), _p(weight), _p(gx), B, Ci, Co, H, W, K, _stream(x))
            _lib.check(rc, "dn_conv2d_valid_bwd_data")
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw = torch.empty_like(weight)
            gb = torch.empty(Co, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            rc = _lib.lib().dn_conv2d_valid_bwd_weight(_p(x), _p(gy), _p(gw), _p(gb), B, Ci, Co, H, W, K, _stream(x))
            _lib.check(rc, "dn_conv2d_valid_bwd_weight")
        return gx, gw, gb


class Conv2dValid(nn.Conv2d):
    """nn.Conv2d for square kernels up to 7 x 7, stride 1, no padding (the layer sits behind an explicit padding module): HIP kernels on
    float32 GPU tensors, torch's path otherwise."""

    def forward(self, x):
        k = self.kernel_size
        if (_hip(x) and x.dim() == 4 and k[0] == k[1] and k[0] <= 7 and tuple(self.stride) == (1, 1)
                and tuple(self.padding) == (0, 0) and tuple(self.dilation) == (1, 1) and self.groups == 1):
            return _Conv2dValid.apply(x, self.weight, self.bias)
        return super().forward(x)
