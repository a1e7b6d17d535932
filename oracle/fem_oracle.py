"""CPU oracle for the FEM Gauss-quadrature hot path -- TEST INFRASTRUCTURE ONLY.

This file restates, in plain PyTorch-CPU / numpy, the algorithm of the reference
(`/root/reference/DiffNet/DiffNetFEM.py` and the `loss()` bodies of its example
scripts).  It is the checker for the HIP kernels and the `cpu_baseline` leg of
`bench.py`; nothing under `diffnet_amd/` may import it (the product path fails
loudly when the HIP library is missing).

Parity pinning: every function here is checked against golden vectors emitted by
`tools/gen_golden.py` from the *imported* reference (tests/golden/*.npz,
tests/test_oracle_golden.py).  Op sequence deliberately mirrors the reference --
one strided convolution per Gauss point, `cat`, broadcast multiplies, `sum`,
`mean`, sliced `+=` assembly -- so that timing it is timing the reference's
formulation (`cpu_baseline.kind = "port"`).

Reference citations are relative to /root/reference.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------------
# quadrature + 1-D bases
# ----------------------------------------------------------------------------------------------


def gauss_rule(ngp_1d: int):
    """DiffNet/DiffNetFEM.py:128-141 -- note the truncated 3- and 4-point literals."""
    if ngp_1d == 1:
        return np.array([0.0]), np.array([2.0])
    if ngp_1d == 2:
        a = 0.5773502691896258
        return np.array([-a, a]), np.array([1.0, 1.0])
    if ngp_1d == 3:
        a = 0.774596669
        return np.array([-a, 0.0, a]), np.array([5.0 / 9.0, 8.0 / 9.0, 5.0 / 9.0])
    if ngp_1d == 4:
        return (np.array([-0.861136, -0.339981, 0.339981, 0.861136]),
                np.array([0.347855, 0.652145, 0.652145, 0.347855]))
    raise ValueError("ngp_1d must be 1..4")


def default_ngp(deg: int, requested: int) -> int:
    """DiffNet/DiffNetFEM.py:29-37: ngp_1d = max(requested, {1:2, 2:3, 3:3}[deg])."""
    return max(requested, {1: 2, 2: 3, 3: 3}[deg])


def basis_1d(deg: int, x: float):
    """Values, first and second derivatives of the 1-D Lagrange basis on [-1, 1].

    Closed forms of DiffNet/DiffNetFEM.py:58-60 (Q1), :71-85 (Q2), :108-126 (Q3).
    Returns three float64 arrays of length deg+1.
    """
    if deg == 1:
        return (np.array([0.5 * (1.0 - x), 0.5 * (1.0 + x)]),
                np.array([-0.5, 0.5]),
                np.array([0.0, 0.0]))
    if deg == 2:
        return (np.array([0.5 * x * (x - 1.0), 1.0 - x * x, 0.5 * x * (x + 1.0)]),
                np.array([0.5 * (2.0 * x - 1.0), -2.0 * x, 0.5 * (2.0 * x + 1.0)]),
                np.array([1.0, -2.0, 1.0]))
    if deg == 3:
        a, b, t, n = 9.0 / 16.0, 27.0 / 16.0, 1.0 / 3.0, 1.0 / 9.0
        return (np.array([-a * (x ** 3 - x ** 2 - n * x + n),
                          b * (x ** 3 - t * x ** 2 - x + t),
                          -b * (x ** 3 + t * x ** 2 - x - t),
                          a * (x ** 3 + x ** 2 - n * x - n)]),
                np.array([-a * (3 * x ** 2 - 2 * x - n),
                          b * (3 * x ** 2 - (2.0 / 3.0) * x - 1),
                          -b * (3 * x ** 2 + (2.0 / 3.0) * x - 1),
                          a * (3 * x ** 2 + 2 * x - n)]),
                np.array([-a * (6.0 * x - 2.0), b * (6.0 * x - (2.0 / 3.0)),
                          -b * (6.0 * x + (2.0 / 3.0)), a * (6.0 * x + 2.0)]))
    raise ValueError("fem_basis_deg must be 1, 2 or 3")


# ----------------------------------------------------------------------------------------------
# geometry + tables
# ----------------------------------------------------------------------------------------------


class FemSpec:
    """Geometry bookkeeping of PDE.__init__ (DiffNet/base.py:16-32) and DiffNetFEM.__init__
    (DiffNet/DiffNetFEM.py:25-51).  kwargs order is X, Y, Z; tensors are laid out (Z, Y, X)."""

    def __init__(self, nsd=2, domain_size=64, domain_length=1.0, domain_sizes=None, domain_lengths=None,
                 ngp_1d=2, fem_basis_deg=1):
        self.nsd = nsd
        self.deg = fem_basis_deg
        self.domain_size = domain_size
        self.domain_length = domain_length
        sizes = domain_sizes if domain_sizes is not None else (domain_size,) * 3
        lens = domain_lengths if domain_lengths is not None else (domain_length,) * 3
        self.sizes = tuple(int(s) for s in sizes[:nsd])        # (X, Y[, Z])
        self.lengths = tuple(float(v) for v in lens[:nsd])
        self.ngp_1d = default_ngp(self.deg, ngp_1d)
        self.ngp_total = self.ngp_1d ** nsd
        self.nbf_1d = self.deg + 1
        self.nbf_total = self.nbf_1d ** nsd
        self.nel = tuple(int((s - 1) / self.deg) for s in self.sizes)
        self.hs = tuple(L / n for L, n in zip(self.lengths, self.nel))
        self.nelem = int((domain_size - 1) / self.deg)
        self.h = domain_length / self.nelem
        self.gpx_1d, self.gpw_1d = gauss_rule(self.ngp_1d)


def build_tables(spec: FemSpec):
    """Per-Gauss-point kernels and dense value tables.

    2-D: DiffNet/DiffNetFEM.py:196-227; 3-D: :405-453 including the quirks that the
    3-D second-derivative kernels are indexed [ibf, jbf, kbf] and that `d2N_z_gp`
    receives the d2N_x kernel (:430-435, :450).  Products are formed in float64 in the
    reference's association order and rounded once to float32 on store.
    Returns dict name -> float32 tensor: kernel lists stacked to (G, 1, 1, *nbf) and
    `*_values` shaped (1, nbf_total, G, 1, 1[, 1]); 'gpw' (G,).
    """
    nsd, ng, nb, deg = spec.nsd, spec.ngp_1d, spec.nbf_1d, spec.deg
    G = spec.ngp_total
    B = [basis_1d(deg, float(x)) for x in spec.gpx_1d]       # per 1-D gauss point: (val, d1, d2)
    sx = [2.0 / h for h in spec.hs]
    kshape = (nb,) * nsd
    t = {}
    if nsd == 2:
        names = ["N_gp", "dN_x_gp", "dN_y_gp", "d2N_x_gp", "d2N_y_gp", "d2N_xy_gp"]
        vals = ["Nvalues", "dN_x_values", "dN_y_values", "d2N_x_values", "d2N_y_values", "d2N_xy_values"]
        K = {n: np.zeros((G,) + kshape, dtype=np.float32) for n in names}
        V = {n: np.ones((1, spec.nbf_total, G, 1, 1), dtype=np.float32) for n in vals}
        gpw = np.zeros(G, dtype=np.float32)
        for jg in range(ng):
            for ig in range(ng):
                g = ng * jg + ig
                gpw[g] = spec.gpw_1d[ig] * spec.gpw_1d[jg]
                (bi, di, ei), (bj, dj, ej) = B[ig], B[jg]
                for jb in range(nb):
                    for ib in range(nb):
                        K["N_gp"][g, jb, ib] = bi[ib] * bj[jb]
                        K["dN_x_gp"][g, jb, ib] = di[ib] * bj[jb] * sx[0]
                        K["dN_y_gp"][g, jb, ib] = bi[ib] * dj[jb] * sx[1]
                        K["d2N_x_gp"][g, jb, ib] = ei[ib] * bj[jb] * sx[0] ** 2
                        K["d2N_y_gp"][g, jb, ib] = bi[ib] * ej[jb] * sx[1] ** 2
                        K["d2N_xy_gp"][g, jb, ib] = di[ib] * dj[jb] * sx[0] * sx[1]
                        a = nb * jb + ib
                        for kn, vn in zip(names, vals):
                            V[vn][0, a, g] = K[kn][g, jb, ib]
        # edge ("surface") tables, DiffNet/DiffNetFEM.py:244-269
        S = {n: np.zeros((ng, nb), dtype=np.float32) for n in ["N_gp_surf", "dN_x_gp_surf", "dN_y_gp_surf"]}
        for ig in range(ng):
            S["N_gp_surf"][ig] = B[ig][0]
            S["dN_x_gp_surf"][ig] = B[ig][1] * sx[0]
            S["dN_y_gp_surf"][ig] = B[ig][1] * sx[1]
        for n, arr in S.items():
            t[n] = torch.from_numpy(arr.reshape(ng, 1, 1, nb))
        t["gpw_surf"] = torch.from_numpy(spec.gpw_1d.astype(np.float32))
        for kn, vn in zip(["N_gp_surf", "dN_x_gp_surf", "dN_y_gp_surf"],
                          ["Nvalues_surf", "dN_x_values_surf", "dN_y_values_surf"]):
            t[vn] = torch.from_numpy(np.ascontiguousarray(S[kn].T).reshape(1, nb, ng, 1))
    else:
        names = ["N_gp", "dN_x_gp", "dN_y_gp", "dN_z_gp", "d2N_x_gp", "d2N_y_gp", "d2N_z_gp",
                 "d2N_xy_gp", "d2N_yz_gp", "d2N_zx_gp"]
        vals = ["Nvalues", "dN_x_values", "dN_y_values", "dN_z_values", "d2N_x_values", "d2N_y_values", "d2N_z_values"]
        K = {n: np.zeros((G,) + kshape, dtype=np.float32) for n in names}
        V = {n: np.ones((1, spec.nbf_total, G, 1, 1, 1), dtype=np.float32) for n in vals}
        gpw = np.zeros(G, dtype=np.float32)
        for kg in range(ng):
            for jg in range(ng):
                for ig in range(ng):
                    g = (kg * ng + jg) * ng + ig
                    gpw[g] = spec.gpw_1d[ig] * spec.gpw_1d[jg] * spec.gpw_1d[kg]
                    (bi, di, ei), (bj, dj, ej), (bk, dk, ek) = B[ig], B[jg], B[kg]
                    for kb in range(nb):
                        for jb in range(nb):
                            for ib in range(nb):
                                K["N_gp"][g, kb, jb, ib] = bi[ib] * bj[jb] * bk[kb]
                                K["dN_x_gp"][g, kb, jb, ib] = di[ib] * bj[jb] * bk[kb] * sx[0]
                                K["dN_y_gp"][g, kb, jb, ib] = bi[ib] * dj[jb] * bk[kb] * sx[1]
                                K["dN_z_gp"][g, kb, jb, ib] = bi[ib] * bj[jb] * dk[kb] * sx[2]
                                # reference quirk: transposed index order for every 2nd-derivative kernel
                                K["d2N_x_gp"][g, ib, jb, kb] = ei[ib] * bj[jb] * bk[kb] * sx[0] ** 2
                                K["d2N_y_gp"][g, ib, jb, kb] = bi[ib] * ej[jb] * bk[kb] * sx[1] ** 2
                                K["d2N_z_gp"][g, ib, jb, kb] = bi[ib] * bj[jb] * ek[kb] * sx[2] ** 2
                                K["d2N_xy_gp"][g, ib, jb, kb] = di[ib] * dj[jb] * bk[kb] * sx[0] * sx[1]
                                K["d2N_yz_gp"][g, ib, jb, kb] = bi[ib] * dj[jb] * dk[kb] * sx[1] * sx[2]
                                K["d2N_zx_gp"][g, ib, jb, kb] = di[ib] * bj[jb] * dk[kb] * sx[2] * sx[0]
                    # dense value tables are filled from the kernels *inside* the basis loop in the
                    # reference, i.e. d2 entries are read back at [kb,jb,ib] while only partly written;
                    # replay that read-after-partial-write order exactly.
                    K2 = {n: np.zeros(kshape, dtype=np.float32) for n in ["d2N_x_gp", "d2N_y_gp", "d2N_z_gp"]}
                    for kb in range(nb):
                        for jb in range(nb):
                            for ib in range(nb):
                                K2["d2N_x_gp"][ib, jb, kb] = K["d2N_x_gp"][g, ib, jb, kb]
                                K2["d2N_y_gp"][ib, jb, kb] = K["d2N_y_gp"][g, ib, jb, kb]
                                K2["d2N_z_gp"][ib, jb, kb] = K["d2N_z_gp"][g, ib, jb, kb]
                                a = (kb * nb + jb) * nb + ib
                                V["Nvalues"][0, a, g] = K["N_gp"][g, kb, jb, ib]
                                V["dN_x_values"][0, a, g] = K["dN_x_gp"][g, kb, jb, ib]
                                V["dN_y_values"][0, a, g] = K["dN_y_gp"][g, kb, jb, ib]
                                V["dN_z_values"][0, a, g] = K["dN_z_gp"][g, kb, jb, ib]
                                V["d2N_x_values"][0, a, g] = K2["d2N_x_gp"][kb, jb, ib]
                                V["d2N_y_values"][0, a, g] = K2["d2N_y_gp"][kb, jb, ib]
                                V["d2N_z_values"][0, a, g] = K2["d2N_z_gp"][kb, jb, ib]
        K["d2N_z_gp"] = K["d2N_x_gp"].copy()      # DiffNet/DiffNetFEM.py:450
    for n in names:
        t[n] = torch.from_numpy(K[n].reshape((G, 1, 1) + kshape))
    for n in vals:
        t[n] = torch.from_numpy(V[n])
    t["gpw"] = torch.from_numpy(gpw)
    return t


def node_coords(spec: FemSpec):
    """DiffNet/DiffNetFEM.py:229-233 (2-D), :455-462 + cuboid_mesh.py:8-19 (3-D): float32 (Z,Y,X) grids."""
    axes = [np.linspace(0, L, n) for L, n in zip(spec.lengths, spec.sizes)]
    if spec.nsd == 2:
        xx, yy = np.meshgrid(axes[0], axes[1])
        return torch.FloatTensor(xx), torch.FloatTensor(yy)
    nx, ny, nz = spec.sizes
    xx = np.broadcast_to(axes[0][None, None, :], (nz, ny, nx))
    yy = np.broadcast_to(axes[1][None, :, None], (nz, ny, nx))
    zz = np.broadcast_to(axes[2][:, None, None], (nz, ny, nx))
    return torch.FloatTensor(xx.copy()), torch.FloatTensor(yy.copy()), torch.FloatTensor(zz.copy())


# ----------------------------------------------------------------------------------------------
# operators
# ----------------------------------------------------------------------------------------------


def gauss_pt_eval(tensor, kernels, nsd=2, stride=1):
    """DiffNet/DiffNetFEM.py:7-18: one strided conv per Gauss point, concatenated on dim 1.
    `kernels`: sequence of (1,1,*nbf) tensors or a stacked (G,1,1,*nbf) tensor."""
    conv = {1: F.conv1d, 2: F.conv2d, 3: F.conv3d}[nsd]
    outs = [conv(tensor, kernels[i].reshape((1, 1) + tuple(kernels[i].shape[-nsd:])), stride=stride)
            for i in range(len(kernels))]
    return torch.cat(outs, 1)


class Oracle:
    """Bundles a FemSpec with its tables and the wrappers of DiffNet/DiffNetFEM.py:143-174."""

    def __init__(self, **kw):
        self.spec = FemSpec(**kw)
        self.t = build_tables(self.spec)
        self.nsd = self.spec.nsd
        self.stride = self.spec.nbf_1d - 1
        self.gpw = self.t["gpw"]

    def ev(self, x, name="N_gp"):
        return gauss_pt_eval(x, self.t[name], self.nsd, self.stride)

    def gp_coords(self):
        return tuple(self.ev(c[None, None]) for c in node_coords(self.spec))

    def _w(self, like, extra=0):
        w = self.gpw.to(like.dtype)
        return w.reshape((1,) * (1 + extra) + (-1,) + (1,) * self.nsd)

    # -- a9 + a10 ------------------------------------------------------------------------------
    def energy(self, u, nu=None, f=None, f_gp=None, dirichlet=(), c=1.0, jac=1.0):
        """L = mean_{b,e} sum_g gpw_g * jac * (c * nu_g * |grad u|_g^2 - u_g f_g).

        Composition of IBN_2D.py:116-134 (c=1, nu==1), 12_klsum.py:53-78 (c=1, nu field),
        solve_in_object_3d.py:75-102 (c=1/2), e8_2d_poisson_mms.py:152-180 (f given at Gauss
        points, Dirichlet field).  `dirichlet`: sequence of (mask, value) applied in order as
        u = where(mask > 0.5, value, u)."""
        for mask, val in dirichlet:
            u = torch.where(mask > 0.5, val + u * 0.0, u)
        u_gp = self.ev(u)
        grads = [self.ev(u, n) for n in ["dN_x_gp", "dN_y_gp", "dN_z_gp"][: self.nsd]]
        g2 = sum(gd ** 2 for gd in grads)
        if nu is not None:
            g2 = self.ev(nu) * g2
        if f_gp is None:
            f_gp = self.ev(f) if f is not None else torch.zeros_like(u_gp)
        dens = self._w(u) * jac * (c * g2 - u_gp * f_gp)
        return torch.mean(torch.sum(dens, 1))

    # -- a11 + a12 + a13 -------------------------------------------------------------------------
    def assemble_q1(self, r_split, out):
        """Q1 element->node scatter-add, e8_2d_poisson_mms.py:85-90 / e8_3d_poisson_mms.py:78-87."""
        if self.nsd == 2:
            a = 0
            for j in (0, 1):
                for i in (0, 1):
                    out[:, 0, j:out.shape[2] - 1 + j, i:out.shape[3] - 1 + i] += r_split[:, a]
                    a += 1
        else:
            a = 0
            for k in (0, 1):
                for j in (0, 1):
                    for i in (0, 1):
                        out[:, 0, k:out.shape[2] - 1 + k, j:out.shape[3] - 1 + j, i:out.shape[4] - 1 + i] += r_split[:, a]
                        a += 1
        return out

    def residual(self, u, nu=None, f=None, f_gp=None, dirichlet=(), jac=1.0, zero_masks=()):
        """Assembled weak-form residual R_a = sum_e sum_g JxW_g (nu_g grad N_a . grad u - N_a f_g)
        (12_klsum.py:80-126, e8_3d_poisson_mms.py:89-136), Q1 only as in the reference."""
        for mask, val in dirichlet:
            u = torch.where(mask > 0.5, val + u * 0.0, u)
        names = ["dN_x", "dN_y", "dN_z"][: self.nsd]
        lhs = 0
        for n in names:
            lhs = lhs + self.t[n + "_values"].to(u.dtype) * self.ev(u, n + "_gp").unsqueeze(1)
        if nu is not None:
            lhs = self.ev(nu).unsqueeze(1) * lhs
        if f_gp is None and f is not None:
            f_gp = self.ev(f)
        if f_gp is not None:
            lhs = lhs - self.t["Nvalues"].to(u.dtype) * f_gp.unsqueeze(1)
        jxw = self._w(u, extra=1) * jac
        r_split = torch.sum(lhs * jxw, 2)
        R = self.assemble_q1(r_split, torch.zeros_like(u))
        for mask in zero_masks:
            R = torch.where(mask > 0.5, R * 0.0, R)
        return R

    def resmin(self, *a, **k):
        return torch.sum(self.residual(*a, **k) ** 2)

    # -- assembly for any degree ------------------------------------------------------------------
    def assemble(self, r_split):
        """Element->node assembly for Q1/Q2/Q3.  The reference has the Q1 helper only (`assemble_q1` above); for higher
        degrees assembly is DEFINED as the adjoint of `gauss_pt_eval` with one-hot tables: with
        K_a = the nbf^nsd kernel that is 1 at local node a and 0 elsewhere, gauss_pt_eval(v, K)[b,a,e] = v at local node a
        of element e, and assemble(r) = d<r, gauss_pt_eval(v, K)>/dv, evaluated by torch autograd on the reference's conv
        formulation (DiffNetFEM.py:7-18).  For Q1 this equals `assemble_q1` (same scatter-add)."""
        nb, nsd = self.spec.nbf_1d, self.nsd
        onehot = torch.eye(nb ** nsd, dtype=r_split.dtype).reshape((nb ** nsd, 1, 1) + (nb,) * nsd)
        node_sp = [n * (nb - 1) + 1 for n in r_split.shape[2:]]
        v = torch.zeros((r_split.shape[0], 1, *node_sp), dtype=r_split.dtype, requires_grad=True)
        gathered = gauss_pt_eval(v, onehot, nsd, self.stride)
        (out,) = torch.autograd.grad(gathered, v, r_split, create_graph=r_split.requires_grad)
        return out

    def residual_any_degree(self, u, nu=None, f=None, dirichlet=(), jac=1.0, zero_masks=()):
        """`residual` with the degree-independent assembly: the broadcast weak form of 12_klsum.py:96-122 on the module's
        dense tables, then `assemble`."""
        for mask, val in dirichlet:
            u = torch.where(mask > 0.5, val + u * 0.0, u)
        lhs = 0
        for n in ["dN_x", "dN_y", "dN_z"][: self.nsd]:
            lhs = lhs + self.t[n + "_values"].to(u.dtype) * self.ev(u, n + "_gp").unsqueeze(1)
        if nu is not None:
            lhs = self.ev(nu).unsqueeze(1) * lhs
        if f is not None:
            lhs = lhs - self.t["Nvalues"].to(u.dtype) * self.ev(f).unsqueeze(1)
        R = self.assemble(torch.sum(lhs * (self._w(u, extra=1) * jac), 2))
        for mask in zero_masks:
            R = torch.where(mask > 0.5, R * 0.0, R)
        return R

    # -- 8(f) row 3 ------------------------------------------------------------------------------
    def l2_err(self, u_sol, exact_fn, u_exact_nodes=None):
        """calc_l2_err (DiffNet/DiffNetFEM.py:348-379 in 2-D, :560-591 in 3-D): Gauss-quadrature L2 norms of the error, the
        solution and the exact solution, plus the nodal vector norm; `u_sol` (B,1,*N), `exact_fn(xgp, ygp[, zgp])`."""
        u_gp = self.ev(u_sol)
        u_ex_gp = exact_fn(*self.gp_coords()).type_as(u_sol)
        jac = 1.0
        for h in self.spec.hs:
            jac = jac * (0.5 * h)
        jxw = (self.gpw.type_as(u_sol) * jac).reshape((1, -1) + (1,) * self.nsd)
        out = [torch.sqrt(torch.sum(torch.sum(v ** 2 * jxw, 1))) for v in (u_gp - u_ex_gp, u_gp, u_ex_gp)]
        if u_exact_nodes is not None:
            n_nodes = 1
            for s_ in self.spec.sizes:
                n_nodes *= s_
            out.append(torch.norm(u_exact_nodes - u_sol, 'fro') / np.sqrt(n_nodes))
        return out

    def l2_err_old(self, u_sol, exact_fn, u_exact_nodes):
        """calc_l2_err_old (DiffNet/DiffNetFEM.py:286-346, :482-558): the element / Gauss-point loops in float64 numpy on a unit
        domain with the exact 2-point rule and unit weights.  Plain Python loops as in the reference (small meshes only)."""
        n, nsd = self.spec.domain_size, self.nsd
        a = 0.577350269189626 if nsd == 2 else float(self.spec.gpx_1d[1])
        x = np.linspace(0, 1, n)
        J = (0.5 / (n - 1)) ** nsd
        tr = lambda lo, hi, g: (lo + hi) / 2. + (hi - lo) / 2. * g
        u = np.asarray(u_sol, dtype=np.float64)
        e2 = s2 = x2 = 0.0
        import itertools
        for el in itertools.product(range(n - 1), repeat=nsd):            # (j, i) or (k, j, i)
            local = u[tuple(slice(c, c + 2) for c in el)]
            for gp in itertools.product((-a, a), repeat=nsd):             # slowest axis first, like the local node order
                basis = np.ones((2,) * nsd)
                for ax, g in enumerate(gp):
                    shp = [1] * nsd
                    shp[ax] = 2
                    basis = basis * (0.5 * np.array([1 - g, 1 + g])).reshape(shp)
                u1 = float(np.sum(local * basis))
                pts = [tr(x[c], x[c + 1], g) for c, g in zip(el, gp)][::-1]      # (x, y[, z])
                u2 = float(exact_fn(*pts))
                e2 += (u1 - u2) ** 2 * J
                s2 += u1 ** 2 * J
                x2 += u2 ** 2 * J
        ue = np.asarray(u_exact_nodes, dtype=np.float64)
        vec = np.linalg.norm(ue.reshape(-1, 1) - u.reshape(-1, 1), 'fro') / (1. * n) ** (nsd / 2.)
        return np.sqrt(e2), np.sqrt(s2), np.sqrt(x2), vec

    # -- a14 -------------------------------------------------------------------------------------
    def fsdt_residuals(self, w, px, py, bc, E=1.0, v=0.25, th=0.1, Ks=1.0, q=1.0):
        """First-order shear-deformation plate residuals, e1_plate_bending_fsdt.py:128-228
        (Dirichlet value 0 for all three fields where bc >= 0.5)."""
        zero = torch.zeros_like(w)
        w, px, py = (torch.where(bc >= 0.5, zero, t_) for t_ in (w, px, py))
        D11 = E * th ** 3 / (12 * (1 - v ** 2)); D22 = D11
        D12 = E * v * th ** 3 / (12 * (1 - v ** 2))
        D66 = E * th ** 3 / (12 * (1 + v))
        A44 = E * th / (2 * (1 + v)); A55 = A44
        ev = self.ev
        Qx = Ks * A55 * (ev(px) + ev(w, "dN_x_gp"))
        Qy = Ks * A44 * (ev(py) + ev(w, "dN_y_gp"))
        pxx, pxy = ev(px, "dN_x_gp"), ev(px, "dN_y_gp")
        pyx, pyy = ev(py, "dN_x_gp"), ev(py, "dN_y_gp")
        Mxx = D11 * pxx + D12 * pyy
        Myy = D12 * pxx + D22 * pyy
        Mxy = D66 * (pxy + pyx)
        N_, Nx, Ny = (self.t[n].to(w.dtype) for n in ["Nvalues", "dN_x_values", "dN_y_values"])
        jac = (0.5 * self.spec.h) ** 2
        jxw = (self.gpw.to(w.dtype) * jac).reshape(1, 1, -1, 1, 1)
        Qx, Qy, Mxx, Myy, Mxy = (t_.unsqueeze(1) for t_ in (Qx, Qy, Mxx, Myy, Mxy))
        t1 = Nx * Qx + Ny * Qy - N_ * (q * torch.ones_like(Qx))
        t2 = Nx * Mxx + Ny * Mxy + N_ * Qx
        t3 = Nx * Mxy + Ny * Myy + N_ * Qy
        Rs = []
        for tt in (t1, t2, t3):
            rs = torch.sum(tt * jxw, 2)
            R = self.assemble_q1(rs, torch.zeros_like(w)) if self.spec.deg == 1 else self.assemble(rs)
            Rs.append(torch.where(bc >= 0.5, zero, R))
        return Rs


def winding_nodes(points, normals, nodes):
    """Vectorised restatement of compute_winding_nodes (IBN/poisson-2d/parametric/IBN_2D.py:89-104):
    points/normals (B,Npts,2), nodes (2,Ny,Nx) -> (B,1,Nx,Ny); L1 distance in the denominator as in the reference."""
    import math
    q = nodes.permute(2, 1, 0)[None, :, :, None, :]                 # (1, Nx, Ny, 1, 2)
    d = points[:, None, None, :, :] - q                             # (B, Nx, Ny, Npts, 2)
    num = (d * normals[:, None, None, :, :]).sum(-1)
    den = (4 * math.pi * d.abs().sum(-1)) ** 3
    return (num / den).sum(-1)[:, None]
