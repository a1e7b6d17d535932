"""FEM operator classes with the reference's surface (`DiffNet/DiffNetFEM.py`): `gauss_pt_eval`,
`DiffNetFEM`, `DiffNet2DFEM`, `DiffNet3DFEM` -- same constructor kwargs, attribute names/shapes and
`state_dict` keys -- backed by the hand-written HIP kernels of libdiffnet_hip.so.

Two levels of use:
  * operator level (drop-in): the 12 `gauss_pt_evaluation*` methods are autograd-aware single-launch
    HIP ops, so unmodified `loss()` bodies of the reference scripts run as they are;
  * fused level (new): `energy_loss`, `energy_loss_and_grad`, `residual`, `residual_loss`, `assemble`
    evaluate a whole Poisson loss body (masks -> Gauss-point evaluation -> integrand -> reduction ->
    gradient / assembly) in one pass over the nodal fields.
"""
import numpy as np
import torch
from torch import nn

from . import ops
from ._lib import DnMesh
from .base import PDE
from .cuboid_mesh import CuboidMesh
from .ops import gauss_pt_eval  # noqa: F401  (re-exported: reference import path DiffNet.DiffNetFEM.gauss_pt_eval)
from .tables import MIN_NGP, Basis1D, gauss_rule, nd_tables


class FemGeometry:
    """Plain-data description of one structured mesh + quadrature: what the C ABI's `dn_mesh` carries."""

    def __init__(self, nsd, sizes_xyz, hs_xyz, deg, ngp_1d, gpx_1d, gpw_1d):
        self.nsd, self.deg, self.ngp_1d = nsd, deg, ngp_1d
        self.sizes = tuple(int(s) for s in sizes_xyz)          # nodes  (x, y[, z])
        self.hs = tuple(float(h) for h in hs_xyz)
        self.nel = tuple((s - 1) // deg for s in self.sizes)   # elements (x, y[, z])
        self.node_shape = self.sizes[::-1]                     # memory order (z, y, x)
        self.elem_shape = self.nel[::-1]
        self.ngp_total = ngp_1d ** nsd
        self.nelem_total = int(np.prod(self.nel))
        self.nnode_total = int(np.prod(self.sizes))
        self.gpx_1d, self.gpw_1d = np.asarray(gpx_1d, float), np.asarray(gpw_1d, float)
        B, D, _ = Basis1D(deg).at_gauss(self.gpx_1d)
        self.basis, self.dbasis = B, D
        self._mesh = {}
        # everything a launch takes from the geometry, as a hashable value (ops.poisson_apply keys its cached prepared calls on it)
        self.key = (nsd, deg, ngp_1d, self.sizes, self.hs, tuple(float(x) for x in self.gpx_1d), tuple(float(x) for x in self.gpw_1d))

    def mesh_struct(self, batch):
        m = self._mesh.get(batch)
        if m is None:
            m = DnMesh()
            m.nsd, m.degree, m.ngp, m.batch = self.nsd, self.deg, self.ngp_1d, batch
            m.nx, m.ny = self.sizes[0], self.sizes[1]
            m.nz = self.sizes[2] if self.nsd == 3 else 1
            for d in range(3):
                m.scale[d] = 2.0 / self.hs[d] if d < self.nsd else 1.0
            for i in range(self.ngp_1d):
                m.gpw[i] = self.gpw_1d[i]
                for j in range(self.deg + 1):
                    m.basis[i][j] = self.basis[i, j]
                    m.dbasis[i][j] = self.dbasis[i, j]
            self._mesh[batch] = m
        return m


class DiffNetFEM(PDE):
    """Quadrature rule, 1-D bases and mesh sizes (DiffNet/DiffNetFEM.py:21-126) + the operator wrappers
    (:143-174) + the fused entry points."""

    def __init__(self, network, dataset=None, **kwargs):
        super().__init__(network, dataset, **kwargs)
        self.ngp_1d = kwargs.get('ngp_1d', 2)
        self.fem_basis_deg = deg = kwargs.get('fem_basis_deg', 1)
        if deg not in MIN_NGP:
            raise ValueError("fem_basis_deg must be 1, 2 or 3")
        self.ngp_1d = max(self.ngp_1d, MIN_NGP[deg])
        self.ngp_total = self.ngp_1d ** self.nsd
        self.gpx_1d, self.gpw_1d = self.gauss_guadrature_scheme(self.ngp_1d)
        if deg > 1:
            assert (self.domain_size - 1) % deg == 0
        self.nelemX = int((self.domain_sizeX - 1) / deg)
        self.nelemY = int((self.domain_sizeY - 1) / deg)
        self.hx = self.domain_lengthX / self.nelemX
        self.hy = self.domain_lengthY / self.nelemY
        if self.nsd == 3:
            self.nelemZ = int((self.domain_sizeZ - 1) / deg)
            self.hz = self.domain_lengthZ / self.nelemZ
        self.nelem = int((self.domain_size - 1) / deg)   # backward compatibility: X-direction value
        self.h = self.domain_length / self.nelem
        self.nbf_1d = deg + 1
        self.nbf_total = self.nbf_1d ** self.nsd
        b1 = Basis1D(deg)
        self.bf_1d, self.bf_1d_der, self.bf_1d_der2 = b1.val, b1.der, b1.der2
        if deg == 1:
            self.bf_1d_th = lambda x: torch.stack((0.5 * (1. - x), 0.5 * (1. + x)))
            self.bf_1d_der_th = lambda x: torch.stack((-0.5 * torch.ones_like(x), 0.5 * torch.ones_like(x)))
            self.bf_1d_der2_th = lambda x: torch.stack((torch.zeros_like(x), torch.zeros_like(x)))
        elif deg == 2:
            self.bf_1d_th = lambda x: torch.stack((0.5 * x * (x - 1.), 1. - x ** 2, 0.5 * x * (x + 1.)))
            self.bf_1d_der_th = lambda x: torch.stack((x - 0.5, -2. * x, x + 0.5))
            self.bf_1d_der2_th = lambda x: torch.stack((torch.ones_like(x), -2. * torch.ones_like(x), torch.ones_like(x)))
        self._table_cache = {}

    # -- construction helpers ----------------------------------------------------------------------
    def gauss_guadrature_scheme(self, ngp_1d):
        return gauss_rule(ngp_1d)

    def _install_tables(self, sizes_xyz, hs_xyz):
        self.geom = FemGeometry(self.nsd, sizes_xyz, hs_xyz, self.fem_basis_deg, self.ngp_1d, self.gpx_1d, self.gpw_1d)
        K, V, w = nd_tables(self.nsd, self.fem_basis_deg, self.gpx_1d, self.gpw_1d, hs_xyz)
        self.gpw = torch.from_numpy(w)
        for name, arr in K.items():
            plist = nn.ParameterList()
            for g in range(arr.shape[0]):
                plist.append(nn.Parameter(torch.from_numpy(arr[g].copy())[None, None], requires_grad=False))
            setattr(self, name, plist)
        for name, arr in V.items():
            setattr(self, name, torch.from_numpy(arr.copy()))
        self._K = K

    def _host_gp_values(self, nodal):
        """Gauss-point interpolation of a nodal numpy field on the host (constructor only: xgp/ygp/zgp)."""
        deg, nb = self.fem_basis_deg, self.nbf_1d
        Kn = self._K["N_gp"]
        out_sp = tuple((n - 1) // deg for n in nodal.shape)
        acc = np.zeros((Kn.shape[0],) + out_sp, dtype=np.float32)
        f32 = nodal.astype(np.float32)
        for idx in np.ndindex(*(nb,) * self.nsd):
            sl = tuple(slice(i, i + deg * (o - 1) + 1, deg) for i, o in zip(idx, out_sp))
            acc += Kn[(slice(None),) + idx].reshape((-1,) + (1,) * self.nsd) * f32[sl][None]
        return torch.from_numpy(acc)[None]

    # -- operator level (DiffNet/DiffNetFEM.py:143-174); the `stride` argument is ignored as in the reference
    def _stacked(self, name, device):
        plist = getattr(self, name)
        if torch.compiler.is_compiling():      # traced: stack in-graph (the version-keyed cache below is host logic dynamo cannot guard on)
            return torch.stack([p.detach().reshape(-1) for p in plist], 0).to(device=device, dtype=torch.float32)
        key = (name, str(device))
        ver = tuple(p._version for p in plist) + (plist[0].data_ptr(),)
        hit = self._table_cache.get(key)
        if hit is None or hit[0] != ver:
            hit = (ver, ops.stack_tables(plist, self.nsd).to(device))
            self._table_cache[key] = hit
        return hit[1]

    def _ev(self, tensor, name, nsd=None):
        nsd = self.nsd if nsd is None else nsd
        return ops._GaussPtEval.apply(tensor, self._stacked(name, tensor.device), nsd, self.nbf_1d, self.nbf_1d - 1)

    def gauss_pt_evaluation(self, tensor, stride=1):
        return self._ev(tensor, "N_gp")

    def gauss_pt_evaluation_surf(self, tensor, stride=1):
        return self._ev(tensor, "N_gp_surf", self.nsd - 1)

    def gauss_pt_evaluation_der_x(self, tensor, stride=1):
        return self._ev(tensor, "dN_x_gp")

    def gauss_pt_evaluation_der_y(self, tensor, stride=1):
        return self._ev(tensor, "dN_y_gp")

    def gauss_pt_evaluation_der_z(self, tensor, stride=1):
        return self._ev(tensor, "dN_z_gp")

    def gauss_pt_evaluation_der2_x(self, tensor, stride=1):
        return self._ev(tensor, "d2N_x_gp")

    def gauss_pt_evaluation_der2_y(self, tensor, stride=1):
        return self._ev(tensor, "d2N_y_gp")

    def gauss_pt_evaluation_der2_z(self, tensor, stride=1):
        return self._ev(tensor, "d2N_z_gp")

    def gauss_pt_evaluation_der2_xy(self, tensor, stride=1):
        return self._ev(tensor, "d2N_xy_gp")

    def gauss_pt_evaluation_der2_yz(self, tensor, stride=1):
        return self._ev(tensor, "d2N_yz_gp")

    def gauss_pt_evaluation_der2_zx(self, tensor, stride=1):
        return self._ev(tensor, "d2N_zx_gp")

    # -- fused level ---------------------------------------------------------------------------------
    def energy_loss(self, u, nu=None, f=None, f_gp=None, dirichlet=(), c=1.0, jac=1.0):
        """mean over (batch, elements) of sum_g gpw_g*jac*(c*nu_g*|grad u|^2 - u_g f_g) with Dirichlet
        conditions `dirichlet=[(mask, value), ...]` applied first -- one HIP pass, differentiable wrt u.
        Equals the loss bodies of IBN_2D.py:116-134 (c=1), solve_in_object_3d.py:75-102 (c=1/2), ..."""
        return ops.energy_loss(self.geom, u, nu, f, f_gp, dirichlet, c, jac)

    def energy_loss_and_grad(self, u, nu=None, f=None, f_gp=None, dirichlet=(), c=1.0, jac=1.0, out=None):
        return ops.energy_loss_and_grad(self.geom, u, nu, f, f_gp, dirichlet, c, jac, out=out)

    def residual(self, u, nu=None, f=None, f_gp=None, dirichlet=(), jac=1.0):
        """Assembled, Dirichlet-masked weak-form residual (12_klsum.py:80-126, e8_3d_poisson_mms.py:89-136)."""
        return ops.residual(self.geom, u, nu, f, f_gp, dirichlet, jac)

    def residual_loss(self, u, nu=None, f=None, f_gp=None, dirichlet=(), jac=1.0):
        """sum(R^2) of `residual` (12_klsum.py:128-131) with fused reduction and single-kernel backward."""
        return ops.residual_loss(self.geom, u, nu, f, f_gp, dirichlet, jac)

    def assemble(self, r_split, out=None):
        """Q1_2D/3D_vector_assembly of the reference scripts (any degree), deterministic."""
        return ops.assemble(r_split, self.nsd, self.nbf_1d, out)


class DiffNet2DFEM(DiffNetFEM):
    """2-D tables, node / Gauss-point coordinates and edge tables (DiffNet/DiffNetFEM.py:178-284)."""

    def __init__(self, network, dataset=None, **kwargs):
        super().__init__(network, dataset, **kwargs)
        assert self.nsd == 2
        self._install_tables((self.domain_sizeX, self.domain_sizeY), (self.hx, self.hy))
        x = np.linspace(0, self.domain_lengthX, self.domain_sizeX)
        y = np.linspace(0, self.domain_lengthY, self.domain_sizeY)
        xx, yy = np.meshgrid(x, y)
        self.xx, self.yy = torch.FloatTensor(xx), torch.FloatTensor(yy)
        self.xgp, self.ygp = self._host_gp_values(xx), self._host_gp_values(yy)
        ng = self.ngp_1d
        gx = torch.tensor(self.gpx_1d, dtype=torch.float32)
        self.xiigp = gx.repeat(ng).reshape(1, -1, 1, 1).expand_as(self.xgp).clone()
        self.etagp = gx.repeat_interleave(ng).reshape(1, -1, 1, 1).expand_as(self.ygp).clone()
        self.gpw_surf = torch.tensor(self.gpw_1d, dtype=torch.float32)

    def _l2_terms(self, u_sol):
        u_gp = self.gauss_pt_evaluation(u_sol)
        u_ex_gp = self.exact_solution(self.xgp, self.ygp).type_as(u_sol)
        jxw = (self.gpw.type_as(u_sol) * (0.5 * self.hx) * (0.5 * self.hy)).reshape(1, -1, 1, 1)
        return [torch.sqrt(torch.sum(v ** 2 * jxw)) for v in (u_gp - u_ex_gp, u_gp, u_ex_gp)]

    def calc_l2_err(self, u_sol):
        """L2 norms of error / solution / exact solution by Gauss quadrature (DiffNet/DiffNetFEM.py:348-379).
        Needs `self.exact_solution` and `self.u_exact` from the subclass; prints like the reference and
        returns (eL2, uL2, u_exL2, vector_norm) -- the reference returns None."""
        eL2, uL2, u_exL2 = self._l2_terms(u_sol)
        u_ex = torch.as_tensor(np.asarray(self.u_exact), dtype=torch.float32).to(u_sol.device)
        vec = torch.norm(u_ex - u_sol, 'fro') / np.sqrt(self.domain_sizeX * self.domain_sizeY)
        print("J = ", (0.5 * self.hx) * (0.5 * self.hy))
        print("usol.shape =", u_sol.shape)
        print("uex.shape =", u_ex.shape)
        print("||u_sol||, ||uex|| = ", uL2, u_exL2)
        print("||e||_{{L2}} = ", eL2)
        print("||e|| (vector-norm) = ", vec)
        return eL2, uL2, u_exL2, vec

    def calc_l2_err_old(self, u_sol):
        """The reference's earlier host routine (DiffNet/DiffNetFEM.py:286-346): Q1, the exact 2-point rule with unit weights,
        a unit square of `domain_size`^2 nodes, float64 numpy; `u_sol` a (N, N) numpy array, `self.u_exact` likewise.  Same
        sums, evaluated for all elements at once instead of in a Python double loop; prints like the reference and returns
        (eL2, uL2, u_exL2, vector_norm)."""
        u = np.asarray(u_sol, dtype=np.float64)
        n = self.domain_size
        J = (0.5 / (n - 1)) ** 2
        a = 0.577350269189626
        x = np.linspace(0, 1, n)
        lo, hi = x[:-1], x[1:]
        sums = np.zeros(3)
        for gx, gy in ((-a, -a), (a, -a), (-a, a), (a, a)):
            N = 0.25 * np.array([(1 - gx) * (1 - gy), (1 + gx) * (1 - gy), (1 - gx) * (1 + gy), (1 + gx) * (1 + gy)])
            u1 = N[0] * u[:-1, :-1] + N[1] * u[:-1, 1:] + N[2] * u[1:, :-1] + N[3] * u[1:, 1:]
            xp = ((lo + hi) / 2. + (hi - lo) / 2. * gx)[None, :]
            yp = ((lo + hi) / 2. + (hi - lo) / 2. * gy)[:, None]
            u2 = np.asarray(self.exact_solution(xp, yp), dtype=np.float64)
            sums += [np.sum((u1 - u2) ** 2) * J, np.sum(u1 ** 2) * J, np.sum(np.broadcast_to(u2, u1.shape) ** 2) * J]
        eL2, uL2, u_exL2 = np.sqrt(sums)
        u_ex = np.asarray(self.u_exact, dtype=np.float64)
        vec = np.linalg.norm(u_ex - u, 'fro') / n
        print("J = ", J)
        print("usol.shape =", u.shape)
        print("uex.shape =", u_ex.shape)
        print("||u_sol||, ||uex|| = ", uL2, u_exL2)
        print("||e||_{{L2}} = ", eL2)
        print("||e|| (vector-norm) = ", vec)
        return eL2, uL2, u_exL2, vec


class DiffNet3DFEM(DiffNetFEM):
    """3-D tables and coordinates (DiffNet/DiffNetFEM.py:382-465), including the reference's transposed
    second-derivative kernels and `d2N_z_gp` == `d2N_x_gp` (bug-compatible on purpose: state_dict parity)."""

    def __init__(self, network, dataset=None, **kwargs):
        super().__init__(network, dataset, **kwargs)
        assert self.nsd == 3
        self._install_tables((self.domain_sizeX, self.domain_sizeY, self.domain_sizeZ), (self.hx, self.hy, self.hz))
        x = np.linspace(0, self.domain_lengthX, self.domain_sizeX)
        y = np.linspace(0, self.domain_lengthY, self.domain_sizeY)
        z = np.linspace(0, self.domain_lengthZ, self.domain_sizeZ)
        xx, yy, zz = CuboidMesh.meshgrid_3d(x, y, z)
        self.xx, self.yy, self.zz = torch.FloatTensor(xx), torch.FloatTensor(yy), torch.FloatTensor(zz)
        self.xgp, self.ygp, self.zgp = (self._host_gp_values(a) for a in (xx, yy, zz))

    def _l2_terms(self, u_sol):
        u_gp = self.gauss_pt_evaluation(u_sol)
        u_ex_gp = self.exact_solution(self.xgp, self.ygp, self.zgp).type_as(u_sol)
        jxw = (self.gpw.type_as(u_sol) * (0.5 * self.hx) * (0.5 * self.hy) * (0.5 * self.hz)).reshape(1, -1, 1, 1, 1)
        return [torch.sqrt(torch.sum(v ** 2 * jxw)) for v in (u_gp - u_ex_gp, u_gp, u_ex_gp)]

    def calc_l2_err(self, u_sol):
        """DiffNet/DiffNetFEM.py:560-591; returns (eL2, uL2, u_exL2, vector_norm) -- the reference returns None."""
        eL2, uL2, u_exL2 = self._l2_terms(u_sol)
        u_ex = torch.as_tensor(np.asarray(self.u_exact), dtype=torch.float32).to(u_sol.device)
        vec = torch.norm(u_ex - u_sol, 'fro') / np.sqrt(self.domain_sizeX * self.domain_sizeY * self.domain_sizeZ)
        print("J = ", (0.5 * self.hx) * (0.5 * self.hy) * (0.5 * self.hz))
        print("usol.shape =", u_sol.shape)
        print("uex.shape =", u_ex.shape)
        print("||u_sol||, ||uex|| = ", uL2, u_exL2)
        print("||e||_{{L2}} = ", eL2)
        print("||e|| (vector-norm) = ", vec)
        return eL2, uL2, u_exL2, vec

    def calc_l2_err_old(self, u_sol):
        """DiffNet/DiffNetFEM.py:482-558: the host routine in 3-D (Q1, the module's own 2-point rule with unit weights, unit
        cube, float64 numpy); `u_sol` a (N, N, N) numpy array [k, j, i], `self.u_exact` a tensor or array."""
        u = np.asarray(u_sol, dtype=np.float64)
        n = self.domain_size
        J = (0.5 / (n - 1)) ** 3
        g1 = np.asarray(self.gpx_1d, dtype=np.float64)
        x = np.linspace(0, 1, n)
        lo, hi = x[:-1], x[1:]
        mid, half = (lo + hi) / 2., (hi - lo) / 2.
        sums = np.zeros(3)
        for gz in g1[:2]:
            for gy in g1[:2]:
                for gx in g1[:2]:
                    u1 = 0.0
                    for kk, sz in ((0, 1 - gz), (1, 1 + gz)):
                        for jj, sy in ((0, 1 - gy), (1, 1 + gy)):
                            for ii, sx in ((0, 1 - gx), (1, 1 + gx)):
                                u1 = u1 + 0.125 * sx * sy * sz * u[kk:n - 1 + kk, jj:n - 1 + jj, ii:n - 1 + ii]
                    xp, yp, zp = (mid + half * gx)[None, None, :], (mid + half * gy)[None, :, None], (mid + half * gz)[:, None, None]
                    u2 = np.asarray(self.exact_solution(xp, yp, zp), dtype=np.float64)
                    sums += [np.sum((u1 - u2) ** 2) * J, np.sum(u1 ** 2) * J, np.sum(np.broadcast_to(u2, u1.shape) ** 2) * J]
        eL2, uL2, u_exL2 = np.sqrt(sums)
        ue = self.u_exact
        u_ex = (ue.squeeze().detach().cpu().numpy() if isinstance(ue, torch.Tensor) else np.asarray(ue)).astype(np.float64)
        vec = np.linalg.norm(u_ex.reshape(-1, 1) - u.reshape(-1, 1), 'fro') / (1. * n) ** 1.5
        print("J = ", J)
        print("usol.shape =", u.shape)
        print("uex.shape =", u_ex.shape)
        print("||u_sol||, ||uex|| = ", uL2, u_exL2)
        print("||e||_{{L2}} = ", eL2)
        print("||e|| (vector-norm) = ", vec)
        return eL2, uL2, u_exL2, vec
