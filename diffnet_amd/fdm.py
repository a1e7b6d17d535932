"""Finite-difference operator class with the reference's surface (`DiffNet/DiffNetFDM.py`): `DiffNetFDM` with
`derivative_x / _y / _xx / _yy` on replicate-padded fields, the `sobel*` kernels, `h_corr / v_corr` correction matrices
and `pad / pad_d2` modules as attributes (state_dict keys preserved).  The arithmetic -- 3x3 stencil + boundary
fix-up, which the reference performs as conv2d followed by a dense N x N matmul -- is one HIP kernel per call
(`dn_fdm_stencil_fwd/bwd`).  Like the reference the class is hard-wired to nsd = 2, 'fdm' weights, 3-point stencils
(`DiffNetFDM.py:128-130`); its z-derivatives and `calc_laplacian` reference attributes that never exist there
 and raise here."""
import ctypes as C

import numpy as np
import torch
from torch.autograd.function import once_differentiable
from torch import nn

from . import _lib
from .base import PDE
from .ops import _p, _require, _stream


def get_deriv_kernels(nsd, ktype, num_pt, output_dim):
    """3-point kernels of DiffNetFDM.py:6-60 for nsd = 2 (first derivative: central difference x row-average)."""
    if nsd != 2 or num_pt != 3 or ktype not in ("fdm", "sobel"):
        raise NotImplementedError("only the configuration the reference class uses (nsd=2, 3-point, fdm/sobel) is built")
    stencil = np.array([-1.0, 0.0, 1.0], dtype=np.float32) * ((output_dim - 1) / 2.0)
    weights = np.array([1, 1, 1] if ktype == "fdm" else [1, 2, 1], dtype=np.float32)
    d2_stencil = ((output_dim - 1) ** 2) * np.array([1, -2, 1], dtype=np.float32)
    d2_weights = np.array([1, 1, 1], dtype=np.float32)
    ker_x = (np.kron(weights, stencil) / np.sum(weights)).reshape(3, 3)
    ker_xx = (np.kron(d2_weights, d2_stencil) / np.sum(d2_weights)).reshape(3, 3)
    return 1, ker_x, ker_x.T, np.zeros_like(ker_x), 1, ker_xx, ker_xx.T, np.zeros_like(ker_xx)


def get_sobel_correction_matrix(nsd, size, padding_xy, padding_xy_d2):
    """Boundary-correction matrices of DiffNetFDM.py:63-119 (padding 1): identity except the two corner 2x1 blocks."""
    w = size
    cm = np.eye(w, dtype=np.float32)
    cm[0, 0] = cm[w - 1, w - 1] = 4.0
    cm[1, 0] = cm[w - 2, w - 1] = -1.0
    cm2 = np.eye(w, dtype=np.float32)
    cm2[0, 0] = cm2[w - 1, w - 1] = 0.0
    cm2[1, 0] = cm2[w - 2, w - 1] = 1.0
    return cm, cm.T.copy(), cm2, cm2.T.copy()


class _Stencil(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, k9, axis, a, b):
        g = _require(g, "g", 4)
        if g.shape[1] != 1 or g.shape[2] < 3 or g.shape[3] < 3:
            raise ValueError(f"expected a replicate-padded single-channel field (B,1,N+2,N+2), got {tuple(g.shape)}")
        B, ny, nx = g.shape[0], g.shape[2] - 2, g.shape[3] - 2
        out = torch.empty((B, 1, ny, nx), dtype=torch.float32, device=g.device)
        karr = (C.c_float * 9)(*k9)
        rc = _lib.lib().dn_fdm_stencil_fwd(_p(g), _p(out), B, ny, nx, karr, axis, a, b, _stream(g))
        _lib.check(rc, "dn_fdm_stencil_fwd")
        ctx.meta = (k9, axis, a, b, B, ny, nx)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        k9, axis, a, b, B, ny, nx = ctx.meta
        go = _require(go, "grad_output", 4)
        gg = torch.empty((B, 1, ny + 2, nx + 2), dtype=torch.float32, device=go.device)
        karr = (C.c_float * 9)(*k9)
        rc = _lib.lib().dn_fdm_stencil_bwd(_p(go), _p(gg), B, ny, nx, karr, axis, a, b, _stream(go))
        _lib.check(rc, "dn_fdm_stencil_bwd")
        return gg, None, None, None, None


class _StencilFused(torch.autograd.Function):
    """pad + stencil + boundary fix-up in one launch each way (dn_fdm_fused_fwd/bwd): input and output on the same grid."""

    @staticmethod
    def forward(ctx, u, k9, axis, a, b):
        u = _require(u, "u", 4)
        if u.shape[1] != 1:
            raise ValueError(f"expected a single-channel field (B,1,N,N), got {tuple(u.shape)}")
        B, ny, nx = u.shape[0], u.shape[2], u.shape[3]
        out = torch.empty_like(u)
        rc = _lib.lib().dn_fdm_fused_fwd(_p(u), _p(out), B, ny, nx, (C.c_float * 9)(*k9), axis, a, b, _stream(u))
        _lib.check(rc, "dn_fdm_fused_fwd")
        ctx.meta = (k9, axis, a, b, B, ny, nx)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        k9, axis, a, b, B, ny, nx = ctx.meta
        go = _require(go, "grad_output", 4)
        gu = torch.empty_like(go)
        rc = _lib.lib().dn_fdm_fused_bwd(_p(go), _p(gu), B, ny, nx, (C.c_float * 9)(*k9), axis, a, b, _stream(go))
        _lib.check(rc, "dn_fdm_fused_bwd")
        return gu, None, None, None, None


class DiffNetFDM(PDE):
    def __init__(self, network, dataset=None, **kwargs):
        super().__init__(network, dataset, **kwargs)
        self.nsd = 2
        self.ktype = 'fdm'
        self.stencil_len = 3
        p1, kx, ky, kz, p2, kxx, kyy, kzz = get_deriv_kernels(self.nsd, self.ktype, self.stencil_len, self.domain_size)
        cX, cY, cX2, cY2 = get_sobel_correction_matrix(self.nsd, self.domain_size, p1, p2)

        def par(a):
            return nn.Parameter(torch.tensor(np.ascontiguousarray(a, dtype=np.float32)).unsqueeze(0).unsqueeze(1), requires_grad=False)

        self.sobelx, self.sobely, self.sobelz = par(kx), par(ky), par(kz)
        self.sobelxx, self.sobelyy, self.sobelzz = par(kxx), par(kyy), par(kzz)
        self.h_corr, self.v_corr = par(cX), par(cY)
        self.h_corr_d2, self.v_corr_d2 = par(cX2), par(cY2)
        self.pad = nn.ReplicationPad2d(padding=p1)
        self.pad_d2 = nn.ReplicationPad2d(padding=p2)
        self._k = {n: tuple(float(v) for v in np.asarray(k, dtype=np.float32).reshape(-1)) for n, k in
                   (("x", kx), ("y", ky), ("xx", kxx), ("yy", kyy))}

    def derivative_x(self, g):
        return _Stencil.apply(g, self._k["x"], 0, 4.0, -1.0)

    def derivative_y(self, g):
        return _Stencil.apply(g, self._k["y"], 1, 4.0, -1.0)

    def derivative_xx(self, g):
        return _Stencil.apply(g, self._k["xx"], 0, 0.0, 1.0)

    def derivative_yy(self, g):
        return _Stencil.apply(g, self._k["yy"], 1, 0.0, 1.0)

    # fused level (new): the replicate padding the reference applies with `self.pad(u)` is folded into the kernel
    def dx(self, u):
        """== derivative_x(self.pad(u)) in one launch, no padded tensor."""
        return _StencilFused.apply(u, self._k["x"], 0, 4.0, -1.0)

    def dy(self, u):
        return _StencilFused.apply(u, self._k["y"], 1, 4.0, -1.0)

    def dxx(self, u):
        return _StencilFused.apply(u, self._k["xx"], 0, 0.0, 1.0)

    def dyy(self, u):
        return _StencilFused.apply(u, self._k["yy"], 1, 0.0, 1.0)

    def derivative_z(self, g):
        raise NotImplementedError("the reference class is hard-wired to nsd = 2 (DiffNetFDM.py:128); z-derivatives are unreachable there")

    derivative_zz = derivative_z

    def calc_laplacian(self, g):
        raise AttributeError("'DiffNetFDM' object has no attribute 'laplacian' (same as the reference: DiffNetFDM.py:201-203)")
