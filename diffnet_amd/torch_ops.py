"""`torch.library` registration of the hot-path operators ("what a native replacement must export", item 1).

The C ABI (include/diffnet_hip.h) is bound with ctypes; an `autograd.Function` around a ctypes call is opaque to
`torch.compile` / `torch.export` (graph break at best).  Registering the same launches as custom operators in the
`diffnet_mi` namespace -- schema, fake (meta) implementation, autograd formula -- makes user `loss()` bodies that call
`gauss_pt_evaluation*`, `assemble`, `energy_loss`, `residual`, `residual_loss` traceable as ordinary graph nodes:

    diffnet_mi::gauss_pt_eval_fwd(Tensor u, Tensor tables, int nsd, int nbf, int stride) -> Tensor
    diffnet_mi::gauss_pt_eval_bwd(Tensor grad_out, Tensor tables, int[] shape, int nsd, int nbf, int stride) -> Tensor
    diffnet_mi::assemble(Tensor r_split, int nsd, int nbf) -> Tensor            (+ assemble_bwd, its gather adjoint)
    diffnet_mi::poisson_apply(Tensor u, Tensor? nu, Tensor? f, Tensor? f_gp, Tensor? mask0, Tensor? field0, float value0,
                              Tensor? mask1, Tensor? field1, float value1, int nsd, int deg, int ngp, int[] sizes, float[] hs,
                              float alpha, float beta, float c, float wscale, float out_scale, float loss_scale)
                              -> (Tensor out, Tensor sums, Tensor loss)

The functions of `diffnet_amd.ops` / the FEM classes call these operators; the kernels behind them are the same C-ABI
launches.  Nothing here computes on the CPU: the real implementations require GPU tensors (no fallback), the fake ones only
propagate shapes.
"""
from typing import List, Optional, Tuple

import torch
from torch.library import custom_op

from . import ops as _ops

NS = "diffnet_mi"


# ---- gauss_pt_eval and its adjoint (each is the other's backward: linear operators) -----------------------------------
@custom_op(f"{NS}::gauss_pt_eval_fwd", mutates_args=())
def gauss_pt_eval_fwd(u: torch.Tensor, tables: torch.Tensor, nsd: int, nbf: int, stride: int) -> torch.Tensor:
    return _ops._gpe_fwd(u, tables, nsd, nbf, stride)


@gauss_pt_eval_fwd.register_fake
def _(u, tables, nsd, nbf, stride):
    return u.new_empty((u.shape[0], tables.shape[0], *[(n - nbf) // stride + 1 for n in u.shape[2:]]))


@custom_op(f"{NS}::gauss_pt_eval_bwd", mutates_args=())
def gauss_pt_eval_bwd(grad_out: torch.Tensor, tables: torch.Tensor, shape: List[int], nsd: int, nbf: int, stride: int) -> torch.Tensor:
    return _ops._gpe_bwd(grad_out, tables, tuple(shape), nsd, nbf, stride)


@gauss_pt_eval_bwd.register_fake
def _(grad_out, tables, shape, nsd, nbf, stride):
    return grad_out.new_empty(tuple(shape))


def _gpe_fwd_setup(ctx, inputs, output):
    u, tables, nsd, nbf, stride = inputs
    ctx.save_for_backward(tables)
    ctx.meta = (list(u.shape), nsd, nbf, stride)


def _gpe_fwd_backward(ctx, g):
    (tables,) = ctx.saved_tensors
    shape, nsd, nbf, stride = ctx.meta
    return gauss_pt_eval_bwd(g, tables, shape, nsd, nbf, stride), None, None, None, None


def _gpe_bwd_setup(ctx, inputs, output):
    _, tables, _, nsd, nbf, stride = inputs
    ctx.save_for_backward(tables)
    ctx.meta = (nsd, nbf, stride)


def _gpe_bwd_backward(ctx, gg):
    (tables,) = ctx.saved_tensors
    nsd, nbf, stride = ctx.meta
    return gauss_pt_eval_fwd(gg, tables, nsd, nbf, stride), None, None, None, None, None


gauss_pt_eval_fwd.register_autograd(_gpe_fwd_backward, setup_context=_gpe_fwd_setup)
gauss_pt_eval_bwd.register_autograd(_gpe_bwd_backward, setup_context=_gpe_bwd_setup)


# ---- assembly and its gather adjoint -----------------------------------------------------------------------------------
@custom_op(f"{NS}::assemble", mutates_args=())
def assemble(r_split: torch.Tensor, nsd: int, nbf: int) -> torch.Tensor:
    return _ops._assemble_raw(r_split, nsd, nbf, None)


@assemble.register_fake
def _(r_split, nsd, nbf):
    return r_split.new_empty((r_split.shape[0], 1, *[n * (nbf - 1) + 1 for n in r_split.shape[2:]]))


@custom_op(f"{NS}::assemble_bwd", mutates_args=())
def assemble_bwd(grad_out: torch.Tensor, shape: List[int], nsd: int, nbf: int) -> torch.Tensor:
    return _ops._assemble_bwd_raw(grad_out, tuple(shape), nsd, nbf)


@assemble_bwd.register_fake
def _(grad_out, shape, nsd, nbf):
    return grad_out.new_empty(tuple(shape))


def _asm_setup(ctx, inputs, output):
    r, nsd, nbf = inputs
    ctx.meta = (list(r.shape), nsd, nbf)


def _asm_backward(ctx, g):
    shape, nsd, nbf = ctx.meta
    return assemble_bwd(g, shape, nsd, nbf), None, None


def _asmb_setup(ctx, inputs, output):
    _, _, nsd, nbf = inputs
    ctx.meta = (nsd, nbf)


def _asmb_backward(ctx, gg):
    nsd, nbf = ctx.meta
    return assemble(gg, nsd, nbf), None, None, None


@custom_op(f"{NS}::assemble_onto", mutates_args=())
def assemble_onto(r_split: torch.Tensor, base: torch.Tensor, nsd: int, nbf: int) -> torch.Tensor:
    """base + scatter_add(r_split), accumulated node by node in the reference's order (Aglobal[...] += R_split[:, a] for a = 0, 1,
    ...: e8_2d_poisson_mms.py:85-90), so the result is bit-identical to the reference helper called on a non-zero Aglobal."""
    return _ops._assemble_raw(r_split, nsd, nbf, base)


@assemble_onto.register_fake
def _(r_split, base, nsd, nbf):
    return torch.empty_like(base)


def _asmo_setup(ctx, inputs, output):
    r, _, nsd, nbf = inputs
    ctx.meta = (list(r.shape), nsd, nbf)


def _asmo_backward(ctx, g):
    shape, nsd, nbf = ctx.meta
    return assemble_bwd(g, shape, nsd, nbf), g, None, None


assemble_onto.register_autograd(_asmo_backward, setup_context=_asmo_setup)
assemble.register_autograd(_asm_backward, setup_context=_asm_setup)
assemble_bwd.register_autograd(_asmb_backward, setup_context=_asmb_setup)


# ---- the fused Poisson operator -----------------------------------------------------------------------------------------
_GEOMS = {}


def _geom(nsd, deg, ngp, sizes, hs):
    key = (nsd, deg, ngp, tuple(sizes), tuple(hs))
    g = _GEOMS.get(key)
    if g is None:
        from .fem import FemGeometry
        from .tables import gauss_rule
        gx, gw = gauss_rule(ngp)
        g = FemGeometry(nsd, sizes, hs, deg, ngp, gx, gw)
        _GEOMS[key] = g
    return g


@custom_op(f"{NS}::poisson_apply", mutates_args=())
def poisson_apply(u: torch.Tensor, nu: Optional[torch.Tensor], f: Optional[torch.Tensor], f_gp: Optional[torch.Tensor],
                  mask0: Optional[torch.Tensor], field0: Optional[torch.Tensor], value0: float,
                  mask1: Optional[torch.Tensor], field1: Optional[torch.Tensor], value1: float,
                  nsd: int, deg: int, ngp: int, sizes: List[int], hs: List[float],
                  alpha: float, beta: float, c: float, wscale: float, out_scale: float,
                  loss_scale: float) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    geom = _geom(nsd, deg, ngp, sizes, hs)
    d = []
    for m, fld, v in ((mask0, field0, value0), (mask1, field1, value1)):
        if m is not None:
            if m.dtype == torch.int32:          # bit-packed mask (ops.PackedMask.bits): (B|1, node rows per sample, ceil(nx / 32))
                rows = 1
                for n in geom.node_shape[:-1]:
                    rows *= int(n)
                if m.dim() != 3 or m.shape[0] not in (1, u.shape[0]) or tuple(m.shape[1:]) != (rows, (int(geom.node_shape[-1]) + 31) // 32):
                    raise TypeError(f"diffnet_mi::poisson_apply: an int32 mask is a bit-packed mask (ops.PackedMask.bits) of shape "
                                    f"(B|1, {rows}, {(int(geom.node_shape[-1]) + 31) // 32}); got {tuple(m.shape)} -- mask images are float32, uint8 or bool")
                m = _ops.PackedMask(m, (m.shape[0], 1, *geom.node_shape))
            d.append(_ops.Dirichlet(m, fld if fld is not None else v))
    out, sums, loss = _ops.poisson_apply(geom, u, nu, f, f_gp, d, alpha=alpha, beta=beta, c=c, wscale=wscale, out_scale=out_scale,
                                         want_out=True, want_sums=True, loss_scale=loss_scale)
    return out, sums, loss


@poisson_apply.register_fake
def _(u, nu, f, f_gp, mask0, field0, value0, mask1, field1, value1, nsd, deg, ngp, sizes, hs, alpha, beta, c, wscale, out_scale,
      loss_scale):
    return torch.empty_like(u), u.new_empty((2,), dtype=torch.float64), u.new_empty((), dtype=torch.float32)


def _pa_setup(ctx, inputs, output):
    (u, nu, f, f_gp, m0, f0, v0, m1, f1, v1, nsd, deg, ngp, sizes, hs, alpha, beta, c, wscale, out_scale, loss_scale) = inputs
    out, sums, loss = output
    ctx.set_materialize_grads(False)
    ctx.save_for_backward(out, nu, m0, m1)
    ctx.meta = (nsd, deg, ngp, sizes, hs, alpha, beta, c, wscale, out_scale, loss_scale)


def _pa_backward(ctx, g_out, g_sums, g_loss):
    """Cotangent wrt u of all three outputs.  With M the projector onto the free nodes the operator is out = s (alpha M K M u
    - beta M f + const): its Jacobian s alpha M K M is symmetric, so J^T v is the same launch on v with the masks made
    homogeneous and the forcing dropped.  energy: dE/du = out / s when alpha = 2c, beta = 1 (the energy-loss form);
    sumsq = |out / s|^2: d/du = 2 alpha M K M (out / s)."""
    out, nu, m0, m1 = ctx.saved_tensors
    nsd, deg, ngp, sizes, hs, alpha, beta, c, wscale, out_scale, loss_scale = ctx.meta

    def K(v, scale):
        return poisson_apply(v, nu, None, None, m0, None, 0.0, m1, None, 0.0, nsd, deg, ngp, sizes, hs, alpha, 0.0, 0.0, wscale, scale, 0.0)[0]

    gu = None
    if g_out is not None:
        gu = K(g_out.contiguous(), out_scale)
    coef = None
    if g_loss is not None:
        coef = g_loss * loss_scale
    if g_sums is not None:
        coef = g_sums[0].to(torch.float32) if coef is None else coef + g_sums[0].to(torch.float32)
        t = K(out * (2.0 / out_scale), 1.0) * g_sums[1].to(torch.float32)
        gu = t if gu is None else gu + t
    if coef is not None:
        if alpha != 2.0 * c or beta != 1.0:
            if g_loss is not None or c != 0.0:
                raise RuntimeError("diffnet_mi::poisson_apply: the energy output is differentiable only in the energy-loss form "
                                   "(alpha = 2c, beta = 1)")
        else:
            t = out * (coef / out_scale)
            gu = t if gu is None else gu + t
    return (gu,) + (None,) * 20


poisson_apply.register_autograd(_pa_backward, setup_context=_pa_setup)


def geometry_args(geom):
    """The plain-data description of a FemGeometry that the operator schema carries."""
    return geom.nsd, geom.deg, geom.ngp_1d, [int(s) for s in geom.sizes], [float(h) for h in geom.hs]


def dirichlet_args(dirichlet, like=None, compact=True):
    """(mask0, field0, value0, mask1, field1, value1) from up to two Dirichlet conditions.  The operator schema carries tensors
    only: a PackedMask travels as its int32 bit tensor, BoxFaces as the (cached) bit-packed image of the faces (`like`: the
    nodal field, for its shape and device).  compact=False (the call will not run a kernel that reads bits: 3-D, Q2/Q3, forcing at
    the Gauss points): the cached uint8 images travel instead, so that nothing is unpacked per call."""
    out = []
    ds = list(dirichlet) + [None] * (2 - len(dirichlet))
    for d in ds:
        if d is None:
            out += [None, None, 0.0]
            continue
        m = d.mask
        if isinstance(m, _ops.BoxFaces):
            if compact:
                key = ("bits", tuple(like.shape[2:]), str(like.device))
                pm = m._img.get(key)
                if pm is None:
                    pm = m._img[key] = _ops.PackedMask.pack(m.image(like.shape[2:], like.device))
                m = pm
            else:
                m = m.image(like.shape[2:], like.device)
        if isinstance(m, _ops.PackedMask):
            m = m.bits if (compact and not isinstance(d.value, torch.Tensor)) else m.image()
        elif isinstance(m, torch.Tensor) and m.dtype == torch.int32:
            raise TypeError("Dirichlet mask images are float32, uint8 or bool; int32 tensors are reserved for bit-packed masks (wrap them in ops.PackedMask)")
        if isinstance(d.value, torch.Tensor):
            out += [m, d.value, 0.0]
        else:
            out += [m, None, float(d.value)]
    return out
