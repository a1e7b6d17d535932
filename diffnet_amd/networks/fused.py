"""Fused InstanceNorm + activation module backed by the HIP kernels `dn_instnorm_act_fwd/bwd` on the GPU
(one pass pair instead of torch's normalisation + activation kernels); on CPU tensors it evaluates the same formula
with torch ops (the networks are plumbing around the convolutions, which stay MIOpen / oneDNN in this round)."""
import torch
from torch import nn
import torch.nn.functional as F

from .. import _lib
from ..ops import _p, _stream


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
    def backward(ctx, gy):
        x, mean, rstd = ctx.saved_tensors
        gy = gy.contiguous()
        gx = torch.empty_like(x)
        ws, wsb = _workspace(x, mean.numel(), x[0, 0].numel())
        rc = _lib.lib().dn_instnorm_act_bwd(_p(x), _p(mean), _p(rstd), _p(gy), _p(gx), mean.numel(), x[0, 0].numel(), ctx.slope,
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
        if x.is_cuda and x.dtype == torch.float32:
            return _InstNormAct.apply(x, self.eps, self.slope)
        y = F.instance_norm(x, eps=self.eps)
        return y if self.slope == 1.0 else F.leaky_relu(y, self.slope)

    def extra_repr(self):
        return f"slope={self.slope}, eps={self.eps}"
