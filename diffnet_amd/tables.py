"""Host-side construction of quadrature rules, 1-D Lagrange bases and the tensor-product tables
that the reference exposes as module attributes (DiffNet/DiffNetFEM.py:54-141, 183-269, 387-465).

Vectorised numpy, float64 products rounded once to float32 -- in the reference's association
order, so the tables are bit-identical to the reference's (tests/test_tables.py pins that
against golden vectors).  The HIP kernels only consume the small 1-D tables (`Basis1D`); the nd
tables exist for API compatibility (`N_gp`, `dN_x_gp`, ..., `Nvalues`, `state_dict` keys) and for
`gauss_pt_eval` with user-supplied table lists.
"""
import numpy as np


def gauss_rule(ngp_1d):
    """1..4-point Gauss-Legendre rule with the reference's literals (3-pt abscissa truncated to 9
    digits, 4-pt rule to 6 digits: DiffNet/DiffNetFEM.py:128-141)."""
    rules = {
        1: ([0.0], [2.0]),
        2: ([-0.5773502691896258, 0.5773502691896258], [1.0, 1.0]),
        3: ([-0.774596669, 0.0, 0.774596669], [5.0 / 9.0, 8.0 / 9.0, 5.0 / 9.0]),
        4: ([-0.861136, -0.339981, 0.339981, 0.861136], [0.347855, 0.652145, 0.652145, 0.347855]),
    }
    if ngp_1d not in rules:
        raise ValueError(f"ngp_1d={ngp_1d} not supported (1..4)")
    x, w = rules[ngp_1d]
    return np.array(x), np.array(w)


MIN_NGP = {1: 2, 2: 3, 3: 3}


class Basis1D:
    """1-D Lagrange basis of degree 1..3 on [-1, 1] with equispaced nodes.

    `val/der/der2(x)` accept scalars or arrays and return an array of shape (deg+1, *x.shape),
    like the reference's `bf_1d`, `bf_1d_der`, `bf_1d_der2` lambdas."""

    def __init__(self, deg):
        if deg not in (1, 2, 3):
            raise ValueError("fem_basis_deg must be 1, 2 or 3")
        self.deg = deg

    def val(self, x):
        x = np.asarray(x, dtype=float)
        if self.deg == 1:
            return np.array([0.5 * (1.0 - x), 0.5 * (1.0 + x)])
        if self.deg == 2:
            return np.array([0.5 * x * (x - 1.0), (1.0 - x ** 2), 0.5 * x * (x + 1.0)], dtype=float)
        q, r, s, t = 9.0 / 16.0, 27.0 / 16.0, 1.0 / 9.0, 1.0 / 3.0
        return np.array([-q * (x ** 3 - x ** 2 - s * x + s), r * (x ** 3 - t * x ** 2 - x + t),
                         -r * (x ** 3 + t * x ** 2 - x - t), q * (x ** 3 + x ** 2 - s * x - s)], dtype=float)

    def der(self, x):
        x = np.asarray(x, dtype=float)
        one = np.ones_like(x)
        if self.deg == 1:
            return np.array([-0.5 * one, 0.5 * one])
        if self.deg == 2:
            return np.array([0.5 * (2.0 * x - 1.0), -2.0 * x, 0.5 * (2.0 * x + 1.0)], dtype=float)
        q, r, s = 9.0 / 16.0, 27.0 / 16.0, 1.0 / 9.0
        return np.array([-q * (3 * x ** 2 - 2 * x - s), r * (3 * x ** 2 - (2.0 / 3.0) * x - 1),
                         -r * (3 * x ** 2 + (2.0 / 3.0) * x - 1), q * (3 * x ** 2 + 2 * x - s)], dtype=float)

    def der2(self, x):
        x = np.asarray(x, dtype=float)
        one = np.ones_like(x)
        if self.deg == 1:
            return np.array([0.0 * one, 0.0 * one])
        if self.deg == 2:
            return np.array([one, -2.0 * one, one], dtype=float)
        q, r = 9.0 / 16.0, 27.0 / 16.0
        return np.array([-q * (6.0 * x - 2.0), r * (6.0 * x - (2.0 / 3.0)), -r * (6.0 * x + (2.0 / 3.0)),
                         q * (6.0 * x + 2.0)], dtype=float)

    def at_gauss(self, gpx):
        """(B, D, D2) each (ngp, nbf): row ig holds the basis (or derivative) values at gauss point ig."""
        return self.val(gpx).T.copy(), self.der(gpx).T.copy(), self.der2(gpx).T.copy()


def _outer(*vs):
    """Left-associated tensor product ((v0 x v1) x v2): result[i0,i1,i2] with v_k indexed by i_k."""
    out = vs[0]
    for k, v in enumerate(vs[1:], start=1):
        out = out[..., None] * v.reshape((1,) * k + (-1,))
    return out


def nd_tables(nsd, deg, gpx, gpw, hs):
    """All per-Gauss-point kernels of the reference as float32 arrays.

    Returns (kernels, values, gpw_nd): kernels[name] has shape (G, nbf, ..., nbf) with axes in
    (z, y, x) order; values[name] has shape (1, nbf_total, G, 1, ..., 1); gpw_nd (G,).
    Products follow the reference's association order (x factor, y factor, z factor, then the
    2/h scales) so that rounding to float32 is bit-identical."""
    b1 = Basis1D(deg)
    B, D, D2 = b1.at_gauss(gpx)           # (ngp, nbf)
    ng, nb = B.shape
    s = [2.0 / h for h in hs]
    K = {}
    if nsd == 2:
        def tp(fi, fj):                    # [jg, ig, jb, ib] = fi[ig, ib] * fj[jg, jb]
            return fi[None, :, None, :] * fj[:, None, :, None]
        K["N_gp"] = tp(B, B)
        K["dN_x_gp"] = tp(D, B) * s[0]
        K["dN_y_gp"] = tp(B, D) * s[1]
        K["d2N_x_gp"] = tp(D2, B) * s[0] ** 2
        K["d2N_y_gp"] = tp(B, D2) * s[1] ** 2
        K["d2N_xy_gp"] = tp(D, D) * s[0] * s[1]
        K = {k: v.reshape(ng * ng, nb, nb).astype(np.float32) for k, v in K.items()}
        w = (gpw[None, :] * gpw[:, None]).reshape(-1)
        # dense values: (1, a=(jb,ib), g, 1, 1)
        V = {}
        for kn, vn in (("N_gp", "Nvalues"), ("dN_x_gp", "dN_x_values"), ("dN_y_gp", "dN_y_values"),
                       ("d2N_x_gp", "d2N_x_values"), ("d2N_y_gp", "d2N_y_values"), ("d2N_xy_gp", "d2N_xy_values")):
            V[vn] = np.ascontiguousarray(K[kn].reshape(ng * ng, nb * nb).T).reshape(1, nb * nb, ng * ng, 1, 1)
        # 1-D edge tables (DiffNet/DiffNetFEM.py:244-269)
        K["N_gp_surf"] = B.astype(np.float32)
        K["dN_x_gp_surf"] = (D * s[0]).astype(np.float32)
        K["dN_y_gp_surf"] = (D * s[1]).astype(np.float32)
        for kn, vn in (("N_gp_surf", "Nvalues_surf"), ("dN_x_gp_surf", "dN_x_values_surf"), ("dN_y_gp_surf", "dN_y_values_surf")):
            V[vn] = np.ascontiguousarray(K[kn].T).reshape(1, nb, ng, 1)
        return K, V, w.astype(np.float32)

    def tp(fi, fj, fk):                    # [kg, jg, ig, kb, jb, ib] = (fi[ig,ib] * fj[jg,jb]) * fk[kg,kb]
        return (fi[None, None, :, None, None, :] * fj[None, :, None, None, :, None]) * fk[:, None, None, :, None, None]

    G = ng ** 3
    nat = {
        "N_gp": tp(B, B, B),
        "dN_x_gp": tp(D, B, B) * s[0],
        "dN_y_gp": tp(B, D, B) * s[1],
        "dN_z_gp": tp(B, B, D) * s[2],
        "d2N_x_gp": tp(D2, B, B) * s[0] ** 2,
        "d2N_y_gp": tp(B, D2, B) * s[1] ** 2,
        "d2N_z_gp": tp(B, B, D2) * s[2] ** 2,
        "d2N_xy_gp": tp(D, D, B) * s[0] * s[1],
        "d2N_yz_gp": tp(B, D, D) * s[1] * s[2],
        "d2N_zx_gp": tp(D, B, D) * s[2] * s[0],
    }
    for k, v in nat.items():
        v = v.reshape(G, nb, nb, nb).astype(np.float32)
        if k.startswith("d2"):
            # reference quirk (DiffNetFEM.py:430-435): second-derivative kernels are stored at [ib, jb, kb]
            v = np.ascontiguousarray(v.transpose(0, 3, 2, 1))
        K[k] = v
    V = {}
    for kn, vn in (("N_gp", "Nvalues"), ("dN_x_gp", "dN_x_values"), ("dN_y_gp", "dN_y_values"), ("dN_z_gp", "dN_z_values")):
        V[vn] = np.ascontiguousarray(K[kn].reshape(G, nb ** 3).T).reshape(1, nb ** 3, G, 1, 1, 1)
    # reference quirk (DiffNetFEM.py:440-442): d2*_values[a=(kb,jb,ib)] reads the transposed kernel at
    # [kb,jb,ib] while it is still being filled => entries with ib > kb are still zero when read.
    kb, jb, ib = np.meshgrid(np.arange(nb), np.arange(nb), np.arange(nb), indexing="ij")
    written = (ib <= kb).astype(np.float32)
    for kn, vn in (("d2N_x_gp", "d2N_x_values"), ("d2N_y_gp", "d2N_y_values"), ("d2N_z_gp", "d2N_z_values")):
        vv = K[kn] * written[None]
        V[vn] = np.ascontiguousarray(vv.reshape(G, nb ** 3).T).reshape(1, nb ** 3, G, 1, 1, 1)
    K["d2N_z_gp"] = K["d2N_x_gp"].copy()   # DiffNetFEM.py:450 appends d2N_x_gp under the d2N_z_gp name
    w = ((gpw[None, None, :] * gpw[None, :, None]) * gpw[:, None, None]).reshape(-1)
    return K, V, w.astype(np.float32)
