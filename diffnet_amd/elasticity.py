"""First-order shear-deformation (FSDT / Mindlin) plate residuals on the HIP operators -- SURVEY.md 8(a) row a14,
reference `examples/elasticity/single_instance/e1_plate_bending_fsdt.py:128-232`.

Three nodal fields (w, phi_x, phi_y); 9 Gauss-point evaluations, the constitutive combinations (eq. 4 of the
script), three weak-form residuals and their element->node assembly, Dirichlet rows replaced by the boundary
values.  `fsdt_residuals` / `fsdt_loss` are ONE fused launch (`dn_fsdt_apply`, csrc/fsdt.hip; degree 1..3 -- the
reference script is Q1, BASELINE configs[4] asks for Q2) forward and one backward (the operator is the gradient of the
plate energy, so its VJP is the same kernel on the masked cotangents).  `fsdt_residuals_composed` is the same
computation spelled with the single-launch HIP operators (`gauss_pt_evaluation*`, `assemble`), kept as a second
implementation for cross-checks."""
import torch
from torch.autograd.function import once_differentiable

from . import ops


def _constants(E, v, h, K_s):
    D_11 = (E * h ** 3) / (12 * (1 - v ** 2))
    D_12 = (E * v * h ** 3) / (12 * (1 - v ** 2))
    D_66 = (E * h ** 3) / (12 * (1 + v))
    A = (E * h) / (2 * (1 + v))
    return dict(D11=D_11, D12=D_12, D22=D_11, D66=D_66, A44=K_s * A, A55=K_s * A)


class _FsdtResiduals(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, phi_x, phi_y, fem, bc, bc_values, consts, q, wscale):
        outs, _ = ops.fsdt_apply(fem.geom, w, phi_x, phi_y, bc, bc_values, q=q, wscale=wscale, want_sums=False, **consts)
        ctx.fem, ctx.bc, ctx.consts, ctx.wscale = fem, bc, consts, wscale
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, g1, g2, g3):
        # J = M K M with K symmetric and M the projector onto the free nodes: J^T g = M K (M g).  Zero Dirichlet values
        # give both projections (inputs and result rows on Dirichlet nodes become 0), q = 0 drops the load term.
        gs = [g.contiguous() for g in (g1, g2, g3)]
        outs, _ = ops.fsdt_apply(ctx.fem.geom, *gs, ctx.bc, (0.0, 0.0, 0.0), q=0.0, wscale=ctx.wscale, want_sums=False, **ctx.consts)
        return outs[0], outs[1], outs[2], None, None, None, None, None, None


def fsdt_residuals(fem, w, phi_x, phi_y, bc, w_bc=0.0, phi_x_bc=0.0, phi_y_bc=0.0, E=1.0, v=0.25, h=0.1, K_s=1.0, q=1.0,
                   hx=None, hy=None):
    """Assembled residuals (R1, R2, R3) of the three FSDT equations; `bc >= 0.5` marks Dirichlet nodes.  One fused launch;
    differentiable wrt the three fields."""
    hx = fem.h if hx is None else hx
    hy = fem.h if hy is None else hy
    return _FsdtResiduals.apply(w, phi_x, phi_y, fem, bc, (w_bc, phi_x_bc, phi_y_bc), _constants(E, v, h, K_s), q,
                                (0.5 * hx) * (0.5 * hy))


def fsdt_residuals_composed(fem, w, phi_x, phi_y, bc, w_bc=0.0, phi_x_bc=0.0, phi_y_bc=0.0, E=1.0, v=0.25, h=0.1, K_s=1.0, q=1.0,
                            hx=None, hy=None):
    """Same residuals from the single-launch HIP operators (9 gauss_pt_eval launches + torch elementwise + 3 assemblies)."""
    hx = fem.h if hx is None else hx
    hy = fem.h if hy is None else hy

    def fix(t, val):
        return torch.where(bc >= 0.5, val if isinstance(val, torch.Tensor) else torch.full_like(t, val), t)

    w, phi_x, phi_y = fix(w, w_bc), fix(phi_x, phi_x_bc), fix(phi_y, phi_y_bc)
    D_11 = (E * h ** 3) / (12 * (1 - v ** 2))
    D_22 = D_11
    D_12 = (E * v * h ** 3) / (12 * (1 - v ** 2))
    D_66 = (E * h ** 3) / (12 * (1 + v))
    A_44 = A_55 = (E * h) / (2 * (1 + v))
    ev, dx, dy = fem.gauss_pt_evaluation, fem.gauss_pt_evaluation_der_x, fem.gauss_pt_evaluation_der_y
    Q_x = K_s * A_55 * (ev(phi_x) + dx(w))
    Q_y = K_s * A_44 * (ev(phi_y) + dy(w))
    pxx, pxy, pyx, pyy = dx(phi_x), dy(phi_x), dx(phi_y), dy(phi_y)
    M_xx = D_11 * pxx + D_12 * pyy
    M_yy = D_12 * pxx + D_22 * pyy
    M_xy = D_66 * (pxy + pyx)
    dev = w.device
    N, Nx, Ny = (t.to(dev) for t in (fem.Nvalues, fem.dN_x_values, fem.dN_y_values))       # (1, nbf, ngp, 1, 1)
    jxw = (fem.gpw.to(dev) * (0.5 * hx) * (0.5 * hy)).reshape(1, 1, -1, 1, 1)

    def weak(a_x, a_y, a_0):
        """sum_g JxW ( dN_x a_x + dN_y a_y + N a_0 ), per local basis function -> (B, nbf, nelY, nelX)"""
        t = Nx * a_x.unsqueeze(1) + Ny * a_y.unsqueeze(1) + N * a_0.unsqueeze(1)
        return torch.sum(t * jxw, 2)

    qg = torch.full_like(Q_x, q)
    R1 = fem.assemble(weak(Q_x, Q_y, -qg))
    R2 = fem.assemble(weak(M_xx, M_xy, Q_x))
    R3 = fem.assemble(weak(M_xy, M_yy, Q_y))
    return fix(R1, w_bc), fix(R2, phi_x_bc), fix(R3, phi_y_bc)


class _FsdtLoss(torch.autograd.Function):
    """The three Frobenius norms (one (3,) tensor) from the launch that computes the residuals (in-kernel fixed-order
    fp64 sums); the VJP is one more launch: the kernel scales the saved residuals by gout_k / ||R_k|| as it loads them."""

    @staticmethod
    def forward(ctx, w, phi_x, phi_y, fem, bc, bc_values, consts, q, wscale):
        # the launch writes the three norms itself (sqrt of its in-kernel fixed-order fp64 sums): no torch op behind the kernel
        outs, _, norms = ops.fsdt_apply(fem.geom, w, phi_x, phi_y, bc, bc_values, q=q, wscale=wscale, want_sums=False, want_norms=True, **consts)
        ctx.save_for_backward(*outs, norms)
        ctx.fem, ctx.bc, ctx.consts, ctx.wscale = fem, bc, consts, wscale
        return norms

    @staticmethod
    @once_differentiable
    def backward(ctx, gnorms):
        *Rs, norms = ctx.saved_tensors
        # d||R_k||/dR_k = R_k / ||R_k||, with torch's norm_backward convention at ||R_k|| == 0 (zero subgradient): the
        # reference script starts from all-zero fields, where R2 = R3 = 0 exactly (e1_plate_bending_fsdt.py:341-349)
        # (the kernel forms gnorms[k] / norms[k] itself, 0 where the norm is 0)
        outs, _ = ops.fsdt_apply(ctx.fem.geom, *Rs, ctx.bc, (0.0, 0.0, 0.0), q=0.0, wscale=ctx.wscale, want_sums=False,
                                 in_num=gnorms.contiguous(), in_den=norms, **ctx.consts)
        return outs[0], outs[1], outs[2], None, None, None, None, None, None


def fsdt_loss(fem, w, phi_x, phi_y, bc, w_bc=0.0, phi_x_bc=0.0, phi_y_bc=0.0, E=1.0, v=0.25, h=0.1, K_s=1.0, q=1.0, hx=None, hy=None):
    """Frobenius norms of the three residuals (e1_plate_bending_fsdt.py:230-232); one launch forward, one backward."""
    hx = fem.h if hx is None else hx
    hy = fem.h if hy is None else hy
    norms = _FsdtLoss.apply(w, phi_x, phi_y, fem, bc, (w_bc, phi_x_bc, phi_y_bc), _constants(E, v, h, K_s), q, (0.5 * hx) * (0.5 * hy))
    return norms.unbind(0)


_ONES = {}


def fsdt_loss_and_grad(fem, w, phi_x, phi_y, bc, w_bc=0.0, phi_x_bc=0.0, phi_y_bc=0.0, E=1.0, v=0.25, h=0.1, K_s=1.0, q=1.0, hx=None, hy=None,
                       weights=None):
    """(norms, grads): the three residual norms as one (3,) tensor and the gradient of sum_k weights[k] * ||R_k|| (weights: a (3,) float32
    device tensor, default ones) with respect to (w, phi_x, phi_y) -- what `sum(fsdt_loss(...)).backward()` leaves in the fields' .grad --
    from two launches and no autograd graph (the eager autograd engine costs this path 100-200 us of host time per step against ~60 us of
    device time at 1025^2).  Reference: e1_plate_bending_fsdt.py:128-232 + its backward pass."""
    hx = fem.h if hx is None else hx
    hy = fem.h if hy is None else hy
    consts, wscale = _constants(E, v, h, K_s), (0.5 * hx) * (0.5 * hy)
    with torch.no_grad():
        # the first launch leaves per-workgroup partial sums only; the second forms the norms from them, scales by weights / norms and writes the norms
        # (no arrival protocol / final reduction at the end of the first launch: 21.3 -> 14.7 us at 1025^2 Q2, one sample)
        Rs, _, partials = ops.fsdt_apply(fem.geom, w, phi_x, phi_y, bc, (w_bc, phi_x_bc, phi_y_bc), q=q, wscale=wscale, want_sums=False,
                                         defer_norms=True, **consts)
        if weights is None:
            key = (w.device.type, w.device.index)
            weights = _ONES.get(key)
            if weights is None:
                weights = _ONES[key] = torch.ones(3, dtype=torch.float32, device=w.device)
        grads, _, norms = ops.fsdt_apply(fem.geom, *Rs, bc, (0.0, 0.0, 0.0), q=0.0, wscale=wscale, want_sums=False, want_norms=True, in_num=weights,
                                         norms_from=partials, **consts)
    return norms, grads


class _FsdtTotal(torch.autograd.Function):
    """sum_k ||R_k|| as ONE autograd node with a scalar output (fsdt_loss returns three scalars whose sum adds an unbind, two adds and their
    backward nodes to every step)."""

    @staticmethod
    def forward(ctx, w, phi_x, phi_y, fem, bc, bc_values, consts, q, wscale):
        outs, _, norms = ops.fsdt_apply(fem.geom, w, phi_x, phi_y, bc, bc_values, q=q, wscale=wscale, want_sums=False, want_norms=True, **consts)
        ctx.save_for_backward(*outs, norms)
        ctx.fem, ctx.bc, ctx.consts, ctx.wscale = fem, bc, consts, wscale
        return norms.sum()

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        *Rs, norms = ctx.saved_tensors
        outs, _ = ops.fsdt_apply(ctx.fem.geom, *Rs, ctx.bc, (0.0, 0.0, 0.0), q=0.0, wscale=ctx.wscale, want_sums=False,
                                 in_num=gout.expand(3).contiguous(), in_den=norms, **ctx.consts)
        return outs[0], outs[1], outs[2], None, None, None, None, None, None


def fsdt_total_loss(fem, w, phi_x, phi_y, bc, w_bc=0.0, phi_x_bc=0.0, phi_y_bc=0.0, E=1.0, v=0.25, h=0.1, K_s=1.0, q=1.0, hx=None, hy=None):
    """||R1|| + ||R2|| + ||R3|| (the loss of e1_plate_bending_fsdt.py:230-232) as one differentiable scalar."""
    hx = fem.h if hx is None else hx
    hy = fem.h if hy is None else hy
    return _FsdtTotal.apply(w, phi_x, phi_y, fem, bc, (w_bc, phi_x_bc, phi_y_bc), _constants(E, v, h, K_s), q, (0.5 * hx) * (0.5 * hy))
