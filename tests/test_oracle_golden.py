"""Pin the CPU oracle (oracle/fem_oracle.py) against golden vectors produced by the imported
reference (tools/gen_golden.py).  CPU only."""
import glob
import math
import os

import numpy as np
import pytest
import torch

from oracle.fem_oracle import Oracle, gauss_pt_eval, node_coords

from conftest import GOLDEN

FEM_FILES = sorted(glob.glob(os.path.join(GOLDEN, "fem_*.npz")))


def spec_kwargs(z):
    """Rebuild constructor kwargs from the scalars stored in a fem_* fixture."""
    nsd = 3 if "scalar_nelemZ" in z.files else 2
    deg = int(z["scalar_fem_basis_deg"])
    if nsd == 2:
        ny, nx = z["attr_xx"].shape
        sizes = (nx, ny, ny)
        lens = (float(z["attr_xx"].max()), float(z["attr_yy"].max()), 1.0)
    else:
        nz, ny, nx = z["attr_xx"].shape
        sizes = (nx, ny, nz)
        lens = (float(z["attr_xx"].max()), float(z["attr_yy"].max()), float(z["attr_zz"].max()))
    return dict(nsd=nsd, fem_basis_deg=deg, ngp_1d=int(z["scalar_ngp_1d"]), domain_sizes=sizes, domain_lengths=lens,
                domain_size=sizes[0], domain_length=lens[0])


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def tt(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("path", FEM_FILES, ids=[os.path.basename(p)[4:-4] for p in FEM_FILES])
def test_tables_and_operators(path):
    z = np.load(path)
    o = Oracle(**spec_kwargs(z))
    s = o.spec
    assert s.ngp_total == int(z["scalar_ngp_total"]) and s.nbf_total == int(z["scalar_nbf_total"])
    assert math.isclose(s.hs[0], float(z["scalar_hx"]), rel_tol=1e-6)   # lengths were recovered from float32 coords
    for key in z.files:
        if key.startswith("tab_"):
            got = o.t[key[4:]].numpy()
            np.testing.assert_allclose(got, z[key], rtol=2e-6, atol=1e-6 * np.abs(z[key]).max(), err_msg=key)
        elif key.startswith("attr_") and key[5:] in o.t:
            got = o.t[key[5:]].numpy()
            np.testing.assert_allclose(got, z[key], rtol=2e-6, atol=1e-6 * np.abs(z[key]).max(), err_msg=key)
    u = tt(z["in_u"])
    ops = {"gauss_pt_evaluation": "N_gp", "gauss_pt_evaluation_der_x": "dN_x_gp", "gauss_pt_evaluation_der_y": "dN_y_gp",
           "gauss_pt_evaluation_der_z": "dN_z_gp", "gauss_pt_evaluation_der2_x": "d2N_x_gp",
           "gauss_pt_evaluation_der2_y": "d2N_y_gp", "gauss_pt_evaluation_der2_z": "d2N_z_gp",
           "gauss_pt_evaluation_der2_xy": "d2N_xy_gp", "gauss_pt_evaluation_der2_yz": "d2N_yz_gp",
           "gauss_pt_evaluation_der2_zx": "d2N_zx_gp"}
    for op, tab in ops.items():
        if "op_" + op not in z.files:
            continue
        ur = u.clone().requires_grad_(True)
        y = o.ev(ur, tab)
        ref = z["op_" + op]
        np.testing.assert_allclose(y.detach().numpy(), ref, rtol=1e-5, atol=1e-6 * max(1.0, np.abs(ref).max()), err_msg=op)
        (g,) = torch.autograd.grad(y, ur, tt(z["cot_" + op]))
        refg = z["vjp_" + op]
        np.testing.assert_allclose(g.numpy(), refg, rtol=1e-5, atol=1e-6 * max(1.0, np.abs(refg).max()), err_msg=op)
    custom = tt(z["custom_tables"])
    y = gauss_pt_eval(u, custom, nsd=s.nsd, stride=s.nbf_1d - 1)
    np.testing.assert_allclose(y.numpy(), z["op_custom"], rtol=1e-5, atol=1e-6)
    # node and gauss-point coordinates
    coords = node_coords(s)
    for c, n in zip(coords, ["xx", "yy", "zz"]):
        np.testing.assert_array_equal(c.numpy(), z["attr_" + n])
    for c, n in zip(o.gp_coords(), ["xgp", "ygp", "zgp"]):
        np.testing.assert_allclose(c.numpy(), z["attr_" + n], rtol=1e-6, atol=1e-6)
    if s.nsd == 2:
        e = tt(z["in_edge"])
        y = gauss_pt_eval(e, o.t["N_gp_surf"], nsd=1, stride=s.nbf_1d - 1)
        np.testing.assert_allclose(y.numpy(), z["op_gauss_pt_evaluation_surf"], rtol=1e-5, atol=1e-6)


def check(val, ref, grad=None, refg=None, rtol=1e-5, gtol=1e-4):
    np.testing.assert_allclose(float(val), float(ref), rtol=rtol)
    if grad is not None:
        np.testing.assert_allclose(grad.numpy(), refg, rtol=gtol, atol=gtol * np.abs(refg).max())


def vg(fn, u):
    ur = u.clone().requires_grad_(True)
    v = fn(ur)
    (g,) = torch.autograd.grad(v, ur)
    return v.detach(), g


@pytest.mark.parametrize("tag", ["n17_g2", "n17_g3", "n64_g3", "n33_g4"])
def test_ibn2d_energy(tag):
    z = load(f"loss_ibn2d_{tag}.npz")
    o = Oracle(**eval(str(z["kwargs"])))
    src, sink, f = tt(z["source"]), tt(z["sink"]), tt(z["f"])
    v, g = vg(lambda u: o.energy(u, None, f, dirichlet=[(src, 1.0), (sink, 0.0)], c=1.0), tt(z["u"]))
    check(v, z["loss"], g, z["grad_u"])


@pytest.mark.parametrize("tag", ["n17", "n33", "n17_g3"])
def test_klsum_energy_and_resmin(tag):
    z = load(f"loss_klsum_{tag}.npz")
    o = Oracle(**eval(str(z["kwargs"])))
    inp, f = tt(z["inputs"]), tt(z["f"])
    nu, bc1, bc2 = inp[:, 0:1], inp[:, 1:2], inp[:, 2:3]
    d = [(bc1, 1.0), (bc2, 0.0)]
    v, g = vg(lambda u: o.energy(u, nu, f, dirichlet=d, c=1.0), tt(z["u"]))
    check(v, z["energy"], g, z["energy_grad"])
    v, g = vg(lambda u: o.resmin(u, nu, f, dirichlet=d, jac=1.0, zero_masks=[bc1, bc2]), tt(z["u"]))
    check(v, z["resmin"], g, z["resmin_grad"])


@pytest.mark.parametrize("tag", ["n17", "n33_g3"])
def test_e8_2d_energy_fgp_and_dirichlet_field(tag):
    z = load(f"loss_e8_2d_{tag}.npz")
    o = Oracle(**eval(str(z["kwargs"])))
    inp = tt(z["inputs"])
    nu, bc2 = inp[:, 0:1], inp[:, 2:3]
    ubc = tt(z["u_bc"])[None, None]
    v, g = vg(lambda u: o.energy(u, nu, f_gp=tt(z["f_gp"]), dirichlet=[(bc2, ubc)], c=0.5), tt(z["u"]))
    check(v, z["energy"], g, z["energy_grad"])


def test_e8_3d_energy_quirk_and_resmin():
    z = load("loss_e8_3d_n9.npz")
    o = Oracle(**eval(str(z["kwargs"])))
    inp = tt(z["inputs"])
    nu, bc2 = inp[:, 0:1], inp[:, 2:3]
    ubc, fgp = tt(z["u_bc"])[None, None], tt(z["f_gp"])

    def energy_quirk(u):   # e8_3d_poisson_mms.py:165 sums u_y^2 twice and never uses u_z
        u = torch.where(bc2 > 0.5, ubc, u)
        ux, uy = o.ev(u, "dN_x_gp"), o.ev(u, "dN_y_gp")
        dens = o.gpw.reshape(1, -1, 1, 1, 1) * (0.5 * o.ev(nu) * (ux ** 2 + uy ** 2 + uy ** 2) - o.ev(u) * fgp)
        return torch.mean(torch.sum(dens, 1))

    v, g = vg(energy_quirk, tt(z["u"]))
    check(v, z["energy"], g, z["energy_grad"])
    jac = (0.5 * o.spec.h) ** 3
    v, g = vg(lambda u: o.resmin(u, nu, f_gp=fgp, dirichlet=[(bc2, ubc)], jac=jac, zero_masks=[bc2]), tt(z["u"]))
    check(v, z["resmin"], g, z["resmin_grad"])


@pytest.mark.parametrize("tag", ["n9", "n17", "box"])
def test_solve_in_object_3d(tag):
    z = load(f"loss_sio3d_{tag}.npz")
    o = Oracle(**eval(str(z["kwargs"])))
    inp, f = tt(z["inputs"]), tt(z["f"])
    v, g = vg(lambda u: o.energy(u, inp[:, 0:1], f, dirichlet=[(inp[:, 1:2], 0.0)], c=0.5), tt(z["u"]))
    check(v, z["loss"], g, z["grad_u"])


@pytest.mark.parametrize("tag", ["n9", "n9_g3"])
def test_ibn3d(tag):
    z = load(f"loss_ibn3d_{tag}.npz")
    o = Oracle(**eval(str(z["kwargs"])))
    src, sink, f = tt(z["source"]), tt(z["sink"]), tt(z["f"])
    srcb = (src > 0.5).float()
    sink_adj = torch.where(srcb == sink, torch.zeros_like(sink), sink)   # IBN_3D.py:120-121
    v, g = vg(lambda u: o.energy(u, None, f, dirichlet=[(src, 1.0), (sink_adj, 0.0)], c=1.0), tt(z["u"]))
    check(v, z["loss"], g, z["grad_u"])


@pytest.mark.parametrize("dom", [16, 64])
def test_reference_test2d(dom):
    z = load(f"loss_reftest2d_n{dom}.npz")
    o = Oracle(domain_size=dom + 2)
    k = tt(z["k"])
    pad = torch.nn.functional.pad

    def fn(u):   # tests/test.py:43-79
        kp = pad(k, (1, 1, 1, 1), "replicate")
        up = pad(u, (0, 0, 1, 1), "replicate")
        up = pad(up, (1, 0, 0, 0), "constant", value=1)
        up = pad(up, (0, 1, 0, 0), "constant", value=0)
        R = o.residual(up, nu=kp, jac=(0.5 * o.spec.h) ** 2)
        return torch.mean(torch.sum(R ** 2, (-1, -2, -3)))

    v, g = vg(fn, tt(z["u"]))
    check(v, z["loss"], g, z["grad_u"])


@pytest.mark.parametrize("dom", [8, 16])
def test_reference_test3d(dom):
    z = load(f"loss_reftest3d_n{dom}.npz")
    o = Oracle(domain_size=dom + 2, nsd=3)
    k = tt(z["k"])
    pad = torch.nn.functional.pad

    def fn(u):   # tests/test3D.py:47-87
        kp = pad(k, (1,) * 6, "replicate")
        up = pad(u, (0, 0, 1, 1, 1, 1), "replicate")
        up = pad(up, (1, 0, 0, 0, 0, 0), "constant", value=1)
        up = pad(up, (0, 1, 0, 0, 0, 0), "constant", value=0)
        R = o.residual(up, nu=kp, jac=(0.5 * o.spec.h) ** 3)
        return torch.mean(torch.sum(R ** 2, (-1, -2, -3, -4)))

    v, g = vg(fn, tt(z["u"]))
    check(v, z["loss"], g, z["grad_u"], rtol=2e-5)


@pytest.mark.parametrize("tag", ["n17", "n33"])
def test_fsdt_plate(tag):
    z = load(f"loss_fsdt_{tag}.npz")
    o = Oracle(**eval(str(z["kwargs"])))
    bc = tt(z["inputs"])[:, 3:4]
    fields = [tt(z[n]).requires_grad_(True) for n in ("w", "phi_x", "phi_y")]
    Rs = o.fsdt_residuals(*fields, bc)
    for i, R in enumerate(Rs):
        ref = z[f"R{i + 1}"]
        np.testing.assert_allclose(R.detach().numpy(), ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max())
        n = torch.norm(R, "fro")
        np.testing.assert_allclose(float(n), float(z["norms"][i]), rtol=1e-5)
        gs = torch.autograd.grad(n, fields, retain_graph=True)
        refg = z[f"grad_norm{i + 1}"]
        for gq, rq in zip(gs, refg):
            np.testing.assert_allclose(gq.numpy(), rq, rtol=1e-4, atol=1e-4 * np.abs(refg).max())


@pytest.mark.parametrize("tag", ["n17_p40", "n33_p100"])
def test_winding_number_oracle(tag):
    from oracle.fem_oracle import winding_nodes
    z = load(f"winding_{tag}.npz")
    w = winding_nodes(tt(z["points"]), tt(z["normals"]), tt(z["nodes"]))
    np.testing.assert_allclose(w.numpy(), z["winding"], rtol=2e-4, atol=1e-5 * np.abs(z["winding"]).max())


# ---------------------------------------------------------------------------------------------
# round 2: calc_l2_err and the degree-independent assembly
# ---------------------------------------------------------------------------------------------
def exact_sines(*xs):
    if torch.is_tensor(xs[0]):
        out = torch.sin(math.pi * xs[0])
        for x in xs[1:]:
            out = out * torch.sin(math.pi * x)
        return out
    out = np.sin(math.pi * xs[0])
    for x in xs[1:]:
        out = out * np.sin(math.pi * x)
    return out


L2_FILES = sorted(glob.glob(os.path.join(GOLDEN, "l2_*.npz")))


@pytest.mark.parametrize("path", L2_FILES, ids=[os.path.basename(p)[3:-4] for p in L2_FILES])
def test_l2_error_norms_vs_reference(path):
    """calc_l2_err / calc_l2_err_old numbers the reference printed (tools/gen_golden.py --l2) vs the oracle restatement and
    vs the product's host routine `calc_l2_err_old` (numpy, no GPU involved)."""
    z = np.load(path)
    kw = eval(str(z["kwargs"]))
    o = Oracle(**kw)
    u_sol, u_ex = tt(z["u_sol"]), tt(z["u_exact"])
    got = o.l2_err(u_sol[None, None], exact_sines, u_ex)
    np.testing.assert_allclose([float(v) for v in got], z["new"], rtol=2e-6)
    if "old" in z.files:
        old = o.l2_err_old(z["u_sol"].astype(np.float64), exact_sines, z["u_exact"])
        np.testing.assert_allclose(old, z["old"], rtol=1e-12)
        from diffnet_amd import DiffNet2DFEM, DiffNet3DFEM
        m = (DiffNet3DFEM if kw.get("nsd", 2) == 3 else DiffNet2DFEM)(None, **kw)
        m.exact_solution, m.u_exact = exact_sines, (u_ex if kw.get("nsd", 2) == 3 else z["u_exact"])
        mine = m.calc_l2_err_old(z["u_sol"].astype(np.float64))
        np.testing.assert_allclose(mine, z["old"], rtol=1e-12)


@pytest.mark.parametrize("nsd,deg,n", [(2, 1, 9), (3, 1, 5), (2, 2, 9), (2, 3, 10), (3, 2, 5)])
def test_assembly_as_adjoint_of_evaluation(nsd, deg, n):
    """`Oracle.assemble` (autograd of the conv formulation with one-hot tables) equals the reference's Q1 slicing helper
    bit for bit at degree 1, and is the exact adjoint of the one-hot evaluation at every degree."""
    o = Oracle(domain_size=n, nsd=nsd, fem_basis_deg=deg)
    g = torch.Generator().manual_seed(5)
    nel = (n - 1) // deg
    rs = torch.rand((2, (deg + 1) ** nsd) + (nel,) * nsd, generator=g) - 0.5
    a = o.assemble(rs)
    assert tuple(a.shape) == (2, 1) + (n,) * nsd
    if deg == 1:
        ref = o.assemble_q1(rs, torch.zeros_like(a))
        np.testing.assert_allclose(a.numpy(), ref.numpy(), rtol=0, atol=1e-6)
    v = torch.rand(a.shape, generator=g)
    nb = deg + 1
    onehot = torch.eye(nb ** nsd).reshape((nb ** nsd, 1, 1) + (nb,) * nsd)
    lhs = float((a.double() * v.double()).sum())
    rhs = float((rs.double() * gauss_pt_eval(v, onehot, nsd, deg).double()).sum())
    assert abs(lhs - rhs) < 1e-6 * abs(lhs)
    np.testing.assert_allclose(float(a.double().sum()), float(rs.double().sum()), rtol=1e-6)   # every entry lands exactly once
