#!/usr/bin/env python3
"""Generate golden vectors for the FEM hot path from the *imported* reference.

Runs ONLY in the build container (needs /root/reference).  It never copies
reference source: it imports the reference package and the reference example
scripts as modules (with stub modules for the third-party packages that are
absent here) and calls their own methods on seeded inputs.  Only data --
inputs and the reference's outputs -- is written, to tests/golden/*.npz.

Shims:
  1. every missing third-party package (pytorch_lightning, libconf, attrdict,
     skimage, trimesh, wandb, ...) is replaced by an auto-stub module whose
     attributes are inert classes; `LightningModule` is a torch.nn.Module with a
     no-op `log`.
  2. `np.float = float` (the reference's Q2/Q3 bases use the removed alias).
Constructors whose example-script `__init__` needs absent data files are
bypassed with `__new__` + the library constructor (DiffNet2DFEM/3DFEM), then the
script's own loss method is called unbound.

Usage: python tools/gen_golden.py  [--out tests/golden]
"""
import argparse
import importlib.abc
import importlib.machinery
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
STUB_ROOTS = {
    "pytorch_lightning", "lightning", "libconf", "attrdict", "skimage", "trimesh",
    "wandb", "tensorboard", "torchvision", "NURBSDiff", "seaborn", "tqdm_stub",
}


class _Inert:
    """Subclassable, callable, attribute-permissive placeholder."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Inert()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Inert()


class _LightningModule(torch.nn.Module):
    def log(self, *a, **k):
        pass

    def log_dict(self, *a, **k):
        pass


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name == "LightningModule":
            return _LightningModule
        if name == "seed_everything":
            return lambda *a, **k: None
        if name == "AttrDict":
            class AttrDict(dict):
                __getattr__ = dict.__getitem__
            return AttrDict
        full = self.__name__ + "." + name
        if full in sys.modules:
            return sys.modules[full]
        cls = type(name, (_Inert,), {})
        setattr(self, name, cls)
        return cls


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def install_shims():
    sys.meta_path.insert(0, _StubFinder())
    np.float = float  # noqa: NPY001  (shim 2)
    sys.path.insert(0, REF)
    import matplotlib
    matplotlib.use("Agg")


def load_script(relpath, modname):
    path = os.path.join(REF, relpath)
    d = os.path.dirname(path)
    if d not in sys.path:
        sys.path.insert(0, d)
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def T(x):
    return x.detach().cpu().numpy()


def rng(seed):
    g = torch.Generator()
    g.manual_seed(seed)
    return g


# --------------------------------------------------------------------------------------
# 1. tables + operator outputs
# --------------------------------------------------------------------------------------
TABLE_LISTS_2D = ["N_gp", "dN_x_gp", "dN_y_gp", "d2N_x_gp", "d2N_y_gp", "d2N_xy_gp",
                  "N_gp_surf", "dN_x_gp_surf", "dN_y_gp_surf"]
TABLE_LISTS_3D = ["N_gp", "dN_x_gp", "dN_y_gp", "dN_z_gp", "d2N_x_gp", "d2N_y_gp", "d2N_z_gp",
                  "d2N_xy_gp", "d2N_yz_gp", "d2N_zx_gp"]
PLAIN_2D = ["gpw", "Nvalues", "dN_x_values", "dN_y_values", "d2N_x_values", "d2N_y_values",
            "d2N_xy_values", "xx", "yy", "xgp", "ygp", "xiigp", "etagp", "gpw_surf",
            "Nvalues_surf", "dN_x_values_surf", "dN_y_values_surf"]
PLAIN_3D = ["gpw", "Nvalues", "dN_x_values", "dN_y_values", "dN_z_values", "d2N_x_values",
            "d2N_y_values", "d2N_z_values", "xx", "yy", "zz", "xgp", "ygp", "zgp"]
OPS_2D = ["gauss_pt_evaluation", "gauss_pt_evaluation_der_x", "gauss_pt_evaluation_der_y",
          "gauss_pt_evaluation_der2_x", "gauss_pt_evaluation_der2_y", "gauss_pt_evaluation_der2_xy"]
OPS_3D = ["gauss_pt_evaluation", "gauss_pt_evaluation_der_x", "gauss_pt_evaluation_der_y",
          "gauss_pt_evaluation_der_z", "gauss_pt_evaluation_der2_x", "gauss_pt_evaluation_der2_y",
          "gauss_pt_evaluation_der2_z", "gauss_pt_evaluation_der2_xy", "gauss_pt_evaluation_der2_yz",
          "gauss_pt_evaluation_der2_zx"]
SCALARS = ["ngp_1d", "fem_basis_deg", "ngp_total", "nelemX", "nelemY", "nelemZ", "nelem", "hx", "hy",
           "hz", "h", "nbf_1d", "nbf_total"]

FEM_CASES = [
    # name, nsd, kwargs
    ("2d_q1_g2_n9", 2, dict(domain_size=9)),
    ("2d_q1_g2_rect", 2, dict(domain_sizes=(12, 9, 9), domain_lengths=(1.5, 1.0, 1.0), domain_size=12,
                              domain_length=1.5)),
    ("2d_q1_g3_n17", 2, dict(domain_size=17, ngp_1d=3)),
    ("2d_q1_g4_n9", 2, dict(domain_size=9, ngp_1d=4)),
    ("2d_q2_g3_n9", 2, dict(domain_size=9, fem_basis_deg=2)),
    ("2d_q2_g3_n17", 2, dict(domain_size=17, fem_basis_deg=2, domain_length=2.0)),
    ("2d_q2_g4_n9", 2, dict(domain_size=9, fem_basis_deg=2, ngp_1d=4)),
    ("2d_q3_g3_n10", 2, dict(domain_size=10, fem_basis_deg=3)),
    ("2d_q3_g4_n10", 2, dict(domain_size=10, fem_basis_deg=3, ngp_1d=4)),
    ("3d_q1_g2_n9", 3, dict(domain_size=9, nsd=3)),
    ("3d_q1_g2_box", 3, dict(domain_sizes=(10, 8, 6), domain_lengths=(2.0, 1.0, 0.5), domain_size=10,
                             domain_length=2.0, nsd=3)),
    ("3d_q1_g3_n5", 3, dict(domain_size=5, nsd=3, ngp_1d=3)),
    ("3d_q2_g3_n5", 3, dict(domain_size=5, nsd=3, fem_basis_deg=2)),
    ("3d_q2_g3_n9", 3, dict(domain_size=9, nsd=3, fem_basis_deg=2)),
]


def gen_fem_case(name, nsd, kw, outdir):
    from DiffNet.DiffNetFEM import DiffNet2DFEM, DiffNet3DFEM
    cls = DiffNet2DFEM if nsd == 2 else DiffNet3DFEM
    m = cls(None, **kw)
    out = {}
    for s in SCALARS:
        if hasattr(m, s):
            out["scalar_" + s] = np.asarray(getattr(m, s))
    out["gpx_1d"] = np.asarray(m.gpx_1d)
    out["gpw_1d"] = np.asarray(m.gpw_1d)
    for tl in (TABLE_LISTS_2D if nsd == 2 else TABLE_LISTS_3D):
        out["tab_" + tl] = np.stack([T(p) for p in getattr(m, tl)], 0)
    for pl_ in (PLAIN_2D if nsd == 2 else PLAIN_3D):
        out["attr_" + pl_] = T(getattr(m, pl_))
    out["state_dict_keys"] = np.array(sorted(m.state_dict().keys()))
    # operators on a random field (B=2) and their vjp with a random cotangent
    if nsd == 2:
        shape = (2, 1, m.domain_sizeY, m.domain_sizeX)
    else:
        shape = (2, 1, m.domain_sizeZ, m.domain_sizeY, m.domain_sizeX)
    g = rng(1234)
    u = torch.rand(shape, generator=g)
    out["in_u"] = T(u)
    for op in (OPS_2D if nsd == 2 else OPS_3D):
        ur = u.clone().requires_grad_(True)
        y = getattr(m, op)(ur)
        cot = torch.rand(y.shape, generator=g)
        (gu,) = torch.autograd.grad(y, ur, cot)
        out["op_" + op] = T(y)
        out["cot_" + op] = T(cot)
        out["vjp_" + op] = T(gu)
    if nsd == 2:
        e = torch.rand((2, 1, m.domain_sizeX), generator=g)
        out["in_edge"] = T(e)
        out["op_gauss_pt_evaluation_surf"] = T(m.gauss_pt_evaluation_surf(e))
    # the free function with a non-default table list (arbitrary user tables)
    from DiffNet.DiffNetFEM import gauss_pt_eval
    custom = [torch.rand(p.shape, generator=g) for p in m.N_gp][: max(1, len(m.N_gp) // 2)]
    out["custom_tables"] = np.stack([T(c) for c in custom], 0)
    out["op_custom"] = T(gauss_pt_eval(u, custom, nsd=nsd, stride=m.nbf_1d - 1))
    np.savez_compressed(os.path.join(outdir, "fem_" + name + ".npz"), **out)
    print("wrote fem_" + name, {k: v.shape for k, v in out.items() if k.startswith("op_")})


# --------------------------------------------------------------------------------------
# 2. loss bodies of the example scripts (called unbound on library-constructed objects)
# --------------------------------------------------------------------------------------
def make(cls, base, **kw):
    obj = cls.__new__(cls)
    base.__init__(obj, None, **kw)
    return obj


def boundary_mask(shape):
    m = torch.zeros(shape)
    nd = len(shape) - 2
    for d in range(nd):
        idx = [slice(None)] * len(shape)
        idx[2 + d] = 0
        m[tuple(idx)] = 1
        idx[2 + d] = -1
        m[tuple(idx)] = 1
    return m


def blob_mask(shape, g, p=0.15):
    return (torch.rand(shape, generator=g) < p).float()


def loss_and_grad(fn, u, *args):
    ur = u.clone().requires_grad_(True)
    val = fn(ur, *args)
    (gu,) = torch.autograd.grad(val, ur)
    return T(val), T(gu)


def gen_losses(outdir):
    from DiffNet.DiffNetFEM import DiffNet2DFEM, DiffNet3DFEM
    ibn2d = load_script("IBN/poisson-2d/parametric/IBN_2D.py", "ref_ibn2d")
    ibn3d = load_script("IBN/poisson-3d/parametric/IBN_3D.py", "ref_ibn3d")
    sio3d = load_script("IBN/poisson-3d/non-parametric/solve_in_object_3d.py", "ref_sio3d")
    kl = load_script("examples/poisson/single_instance/12_klsum.py", "ref_klsum")
    e82 = load_script("examples/poisson/single_instance/e8_2d_poisson_mms.py", "ref_e82d")
    e83 = load_script("examples/poisson/single_instance/e8_3d_poisson_mms.py", "ref_e83d")
    el = load_script("examples/elasticity/single_instance/e1_plate_bending_fsdt.py", "ref_fsdt")
    t2 = load_script("tests/test.py", "ref_test2d")
    t3 = load_script("tests/test3D.py", "ref_test3d")

    # ---- IBN_2D.Poisson.loss : c=1, nu==1, source mask -> 1, sink mask -> 0 (IBN_2D.py:116-134)
    for tag, kw, B in [("n17_g2", dict(domain_size=17), 3), ("n17_g3", dict(domain_size=17, ngp_1d=3), 2),
                       ("n64_g3", dict(domain_size=64, ngp_1d=3), 2), ("n33_g4", dict(domain_size=33, ngp_1d=4), 1)]:
        g = rng(7)
        m = make(ibn2d.Poisson, DiffNet2DFEM, **kw)
        n = m.domain_size
        u = torch.rand((B, 1, n, n), generator=g)
        src = blob_mask((B, 1, n, n), g)
        f = torch.rand((B, 1, n, n), generator=g)
        sink = boundary_mask((B, 1, n, n))
        val, gu = loss_and_grad(lambda uu: ibn2d.Poisson.loss(m, uu, src, f, sink), u)
        np.savez_compressed(os.path.join(outdir, f"loss_ibn2d_{tag}.npz"), kwargs=repr(kw), u=T(u), source=T(src),
                            f=T(f), sink=T(sink), loss=val, grad_u=gu)
        print("ibn2d", tag, val)

    # ---- 12_klsum energy (c=1, nu field) + resmin with assembly (12_klsum.py:53-132)
    for tag, kw, B in [("n17", dict(domain_size=17), 2), ("n33", dict(domain_size=33), 1),
                       ("n17_g3", dict(domain_size=17, ngp_1d=3), 2)]:
        g = rng(11)
        m = make(kl.Poisson, DiffNet2DFEM, **kw)
        n = m.domain_size
        u = torch.rand((B, 1, n, n), generator=g)
        nu = 0.5 + torch.rand((B, 1, n, n), generator=g)
        bc1 = torch.zeros((B, 1, n, n)); bc1[..., :, 0] = 1
        bc2 = torch.zeros((B, 1, n, n)); bc2[..., :, -1] = 1
        inputs = torch.cat([nu, bc1, bc2], 1)
        f = torch.rand((B, 1, n, n), generator=g)
        ve, ge = loss_and_grad(lambda uu: kl.Poisson.loss_EnergyMin(m, uu, inputs, f), u)
        vr, gr = loss_and_grad(lambda uu: kl.Poisson.loss_ResMin(m, uu, inputs, f), u)
        np.savez_compressed(os.path.join(outdir, f"loss_klsum_{tag}.npz"), kwargs=repr(kw), u=T(u), inputs=T(inputs),
                            f=T(f), energy=ve, energy_grad=ge, resmin=vr, resmin_grad=gr)
        print("klsum", tag, ve, vr)

    # ---- e8_2d energy: Dirichlet field u_bc, forcing given AT gauss points (e8_2d_poisson_mms.py:152-180)
    for tag, kw in [("n17", dict(domain_size=17)), ("n33_g3", dict(domain_size=33, ngp_1d=3))]:
        g = rng(13)
        m = make(e82.Poisson, DiffNet2DFEM, **kw)
        n = m.domain_size
        m.u_exact = e82.Poisson.exact_solution(m, m.xx.numpy(), m.yy.numpy())
        m.f_gp = e82.Poisson.forcing_func(m, m.xgp, m.ygp)
        m.u_bc = torch.FloatTensor(m.u_exact)
        u = torch.rand((1, 1, n, n), generator=g)
        nu = torch.ones((1, 1, n, n))
        bc1 = torch.zeros((1, 1, n, n))
        bc2 = boundary_mask((1, 1, n, n))
        inputs = torch.cat([nu, bc1, bc2], 1)
        f = torch.zeros((1, 1, n, n))
        ve, ge = loss_and_grad(lambda uu: e82.Poisson.loss_EnergyMin(m, uu, inputs, f), u)
        np.savez_compressed(os.path.join(outdir, f"loss_e8_2d_{tag}.npz"), kwargs=repr(kw), u=T(u), inputs=T(inputs),
                            u_bc=T(m.u_bc), f_gp=T(m.f_gp), energy=ve, energy_grad=ge)
        print("e8_2d", tag, ve)

    # ---- e8_3d energy (u_y^2 twice quirk) + resmin (e8_3d_poisson_mms.py:89-168)
    for tag, kw in [("n9", dict(domain_size=9, nsd=3))]:
        g = rng(17)
        m = make(e83.Poisson, DiffNet3DFEM, **kw)
        n = m.domain_size
        m.u_exact = e83.Poisson.exact_solution(m, m.xx.numpy(), m.yy.numpy(), m.zz.numpy())
        m.f_gp = e83.Poisson.forcing_func(m, m.xgp, m.ygp, m.zgp)
        m.u_bc = torch.FloatTensor(m.u_exact)
        u = torch.rand((1, 1, n, n, n), generator=g)
        nu = 0.5 + torch.rand((1, 1, n, n, n), generator=g)
        bc1 = torch.zeros((1, 1, n, n, n))
        bc2 = boundary_mask((1, 1, n, n, n))
        inputs = torch.cat([nu, bc1, bc2], 1)
        f = torch.zeros((1, 1, n, n, n))
        ve, ge = loss_and_grad(lambda uu: e83.Poisson.loss_EnergyMin(m, uu, inputs, f), u)
        vr, gr = loss_and_grad(lambda uu: e83.Poisson.loss_ResMin(m, uu, inputs, f), u)
        np.savez_compressed(os.path.join(outdir, f"loss_e8_3d_{tag}.npz"), kwargs=repr(kw), u=T(u), inputs=T(inputs),
                            u_bc=T(m.u_bc), f_gp=T(m.f_gp), energy=ve, energy_grad=ge, resmin=vr, resmin_grad=gr)
        print("e8_3d", tag, ve, vr)

    # ---- solve_in_object_3d loss: c=1/2, nu field, bc1 -> 0 (solve_in_object_3d.py:75-102)
    for tag, kw, B in [("n9", dict(domain_size=9, nsd=3), 2), ("n17", dict(domain_size=17, nsd=3), 1),
                       ("box", dict(domain_sizes=(12, 9, 7), domain_lengths=(1.0, 0.75, 0.5), domain_size=12, nsd=3), 1)]:
        g = rng(19)
        m = make(sio3d.Poisson, DiffNet3DFEM, **kw)
        shape = (B, 1, m.domain_sizeZ, m.domain_sizeY, m.domain_sizeX)
        u = torch.rand(shape, generator=g)
        nu = 0.5 + torch.rand(shape, generator=g)
        bc1 = torch.maximum(boundary_mask(shape), blob_mask(shape, g, 0.05))
        bc2 = torch.zeros(shape)
        inputs = torch.cat([nu, bc1, bc2], 1)
        f = torch.rand(shape, generator=g)
        v, gu = loss_and_grad(lambda uu: sio3d.Poisson.loss(m, uu, inputs, f), u)
        np.savez_compressed(os.path.join(outdir, f"loss_sio3d_{tag}.npz"), kwargs=repr(kw), u=T(u), inputs=T(inputs),
                            f=T(f), loss=v, grad_u=gu)
        print("sio3d", tag, v)

    # ---- IBN_3D loss: c=1, no nu, source -> 1, adjusted sink -> 0 (IBN_3D.py:114-136)
    for tag, kw, B in [("n9", dict(domain_size=9, nsd=3), 2), ("n9_g3", dict(domain_size=9, nsd=3, ngp_1d=3), 1)]:
        g = rng(23)
        m = make(ibn3d.Poisson, DiffNet3DFEM, **kw)
        n = m.domain_size
        shape = (B, 1, n, n, n)
        u = torch.rand(shape, generator=g)
        src = blob_mask(shape, g, 0.1)
        sink = boundary_mask(shape)
        f = torch.rand(shape, generator=g)
        v, gu = loss_and_grad(lambda uu: ibn3d.Poisson.loss(m, uu, src, sink, f), u)
        np.savez_compressed(os.path.join(outdir, f"loss_ibn3d_{tag}.npz"), kwargs=repr(kw), u=T(u), source=T(src),
                            sink=T(sink), f=T(f), loss=v, grad_u=gu)
        print("ibn3d", tag, v)

    # ---- reference tests/test.py + tests/test3D.py residual norms on their own analytic fields
    for dom, B in [(16, 2), (64, 2)]:
        m = make(t2.Test, DiffNet2DFEM, domain_size=dom + 2)  # fields are padded by one node per side
        x = torch.linspace(0.0, 1.0, dom)
        xx, yy = torch.meshgrid(x, x, indexing="ij")
        seed = (torch.sin(math.pi * xx) * torch.sin(math.pi * yy))[None, None]
        u = torch.repeat_interleave(seed, B, 0)
        k = torch.ones((B, 1, dom, dom))
        v, gu = loss_and_grad(lambda uu: t2.Test.calc_residuals(m, uu, k), u)
        np.savez_compressed(os.path.join(outdir, f"loss_reftest2d_n{dom}.npz"), dom=dom, u=T(u), k=T(k), loss=v, grad_u=gu)
        print("reftest2d", dom, v)
    for dom, B in [(8, 1), (16, 1)]:
        m = make(t3.Test3D, DiffNet3DFEM, domain_size=dom + 2, nsd=3)
        x = torch.linspace(0.0, 1.0, dom)
        xx, yy, zz = torch.meshgrid(x, x, x, indexing="ij")
        xx = xx.permute(2, 1, 0)
        seed = ((1 - xx) ** 3)[None, None]
        u = torch.repeat_interleave(seed, B, 0).contiguous()
        k = torch.ones((B, 1, dom, dom, dom))
        v, gu = loss_and_grad(lambda uu: t3.Test3D.calc_residuals(m, uu, k), u)
        np.savez_compressed(os.path.join(outdir, f"loss_reftest3d_n{dom}.npz"), dom=dom, u=T(u), k=T(k), loss=v, grad_u=gu)
        print("reftest3d", dom, v)

    # ---- FSDT plate (3 fields) residuals (e1_plate_bending_fsdt.py:128-232)
    for tag, kw in [("n17", dict(domain_size=17)), ("n33", dict(domain_size=33))]:
        g = rng(29)
        m = make(el.Elastic_FSDT, DiffNet2DFEM, **kw)
        n = m.domain_size
        fx_gp, fy_gp = np.ones_like(m.xgp.numpy()), np.ones_like(m.xgp.numpy())
        m.fx_gp = torch.FloatTensor(fx_gp)
        m.fy_gp = torch.FloatTensor(fy_gp)
        m.w_bc = torch.zeros((n, n)); m.phi_x_bc = torch.zeros((n, n)); m.phi_y_bc = torch.zeros((n, n))
        w = torch.rand((1, 1, n, n), generator=g)
        px = torch.rand((1, 1, n, n), generator=g)
        py = torch.rand((1, 1, n, n), generator=g)
        inputs = torch.zeros((1, 5, n, n))
        inputs[:, 3:4] = boundary_mask((1, 1, n, n))
        f = torch.zeros((1, 1, n, n))
        wr, pxr, pyr = (t.clone().requires_grad_(True) for t in (w, px, py))
        R = el.Elastic_FSDT.calc_residuals(m, (wr, pxr, pyr), inputs, f)
        norms = el.Elastic_FSDT.loss(m, (wr, pxr, pyr), inputs, f)
        out = dict(kwargs=repr(kw), w=T(w), phi_x=T(px), phi_y=T(py), inputs=T(inputs),
                   R1=T(R[0]), R2=T(R[1]), R3=T(R[2]), norms=np.array([T(v) for v in norms]))
        for i, nv in enumerate(norms):
            gs = torch.autograd.grad(nv, (wr, pxr, pyr), retain_graph=True)
            out[f"grad_norm{i + 1}"] = np.stack([T(x) for x in gs], 0)
        np.savez_compressed(os.path.join(outdir, f"loss_fsdt_{tag}.npz"), **out)
        print("fsdt", tag, out["norms"])


# --------------------------------------------------------------------------------------
# 3. networks: seeded default init (weights are NOT stored: the rebuilt net must consume the
#    RNG identically; a weight checksum in the fixture catches any drift), eval mode.
# --------------------------------------------------------------------------------------
def gen_networks(outdir):
    from DiffNet.networks.unets import UNet
    from DiffNet.networks.autoencoders import AE
    from DiffNet.networks.wgan3d import GoodGenerator
    specs = [
        ("unet_2_1_n64", lambda: UNet(in_channels=2, out_channels=1), (2, 2, 64, 64)),
        ("unet_2_1_n96", lambda: UNet(in_channels=2, out_channels=1), (1, 2, 96, 96)),
        ("ae_1_1_d2_n32", lambda: AE(in_channels=1, out_channels=1, n_downsample=2), (2, 1, 32, 32)),
        ("goodgen3d_1_1_n32", lambda: GoodGenerator(in_channels=1, out_channels=1), (1, 1, 32, 32, 32)),
    ]
    for name, ctor, shape in specs:
        torch.manual_seed(2024)
        net = ctor().eval()
        sd = net.state_dict()
        nparam = sum(p.numel() for p in net.parameters())
        checksum = np.array([float(sum(v.double().sum() for v in sd.values())),
                             float(sum(v.double().abs().sum() for v in sd.values()))])
        g = rng(31)
        x = torch.rand(shape, generator=g).requires_grad_(True)
        y = net(x)
        cot = torch.rand(y.shape, generator=g)
        gx, = torch.autograd.grad(y, x, cot, retain_graph=True)
        first = list(net.parameters())[0]
        gw, = torch.autograd.grad(y, first, cot)
        np.savez_compressed(os.path.join(outdir, f"net_{name}.npz"), x=T(x), y=T(y), cot=T(cot), grad_x=T(gx),
                            grad_w0=T(gw), nparam=nparam, checksum=checksum,
                            keys=np.array(list(sd.keys())), torch_version=torch.__version__)
        print("net", name, nparam, checksum)


def param_stats(grads):
    """Per parameter tensor: [sum, l2 norm, dot with a seeded +-1 vector] of its gradient (float64): a small fixture that still pins
    every entry of every parameter gradient (the random projection sees all of them)."""
    out = []
    for i, g in enumerate(grads):
        gd = g.detach().double().reshape(-1).numpy()
        sign = np.random.default_rng(1000 + i).integers(0, 2, gd.size) * 2.0 - 1.0
        out.append([gd.sum(), np.sqrt((gd * gd).sum()), (gd * sign).sum()])
    return np.array(out)


def gen_network_grads(outdir):
    """Gradients wrt EVERY parameter (eval mode, same seeds as gen_networks) as per-parameter statistics, and a TRAIN-mode forward +
    backward (Dropout live; torch.manual_seed(777) right before the call fixes the masks of the CPU generator)."""
    from DiffNet.networks.unets import UNet
    from DiffNet.networks.autoencoders import AE
    from DiffNet.networks.wgan3d import GoodGenerator
    specs = [
        ("unet_2_1_n64", lambda: UNet(in_channels=2, out_channels=1), (2, 2, 64, 64)),
        ("ae_1_1_d2_n32", lambda: AE(in_channels=1, out_channels=1, n_downsample=2), (2, 1, 32, 32)),
        ("goodgen3d_1_1_n32", lambda: GoodGenerator(in_channels=1, out_channels=1), (1, 1, 32, 32, 32)),
    ]
    for name, ctor, shape in specs:
        torch.manual_seed(2024)
        net = ctor().eval()
        g = rng(31)
        x = torch.rand(shape, generator=g).requires_grad_(True)
        y = net(x)
        cot = torch.rand(y.shape, generator=g)
        params = list(net.parameters())
        grads = torch.autograd.grad(y, params, cot)
        net.train()
        torch.manual_seed(777)
        yt = net(x)
        gt = torch.autograd.grad(yt, [x] + params, cot)
        np.savez_compressed(os.path.join(outdir, f"netgrads_{name}.npz"), names=np.array([k for k, _ in net.named_parameters()]),
                            stats=param_stats(grads), y_train=T(yt), grad_x_train=T(gt[0]), stats_train=param_stats(gt[1:]),
                            torch_version=torch.__version__)
        print("netgrads", name, len(params), float(yt.mean()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    out = os.path.abspath(args.out)
    os.makedirs(out, exist_ok=True)
    install_shims()
    torch.set_num_threads(4)
    if args.only in ("", "fem"):
        for name, nsd, kw in FEM_CASES:
            gen_fem_case(name, nsd, kw, out)
    if args.only in ("", "loss"):
        gen_losses(out)
    if args.only in ("", "net"):
        gen_networks(out)
    with open(os.path.join(out, "PROVENANCE.txt"), "w") as fh:
        fh.write("generated by tools/gen_golden.py from the reference imported at /root/reference\n"
                 f"torch {torch.__version__} (CPU path), numpy {np.__version__}\n")


# --------------------------------------------------------------------------------------
# 4. "next" rows: winding-number inside/outside field of IBN_2D.py:89-104
# --------------------------------------------------------------------------------------
def gen_winding(outdir):
    ibn2d = sys.modules.get("ref_ibn2d") or load_script("IBN/poisson-2d/parametric/IBN_2D.py", "ref_ibn2d")
    from DiffNet.DiffNetFEM import DiffNet2DFEM
    for tag, n, npts, B in [("n17_p40", 17, 40, 2), ("n33_p100", 33, 100, 1)]:
        g = rng(37)
        m = DiffNet2DFEM(None, domain_size=n)
        th = torch.sort(torch.rand((B, npts), generator=g) * 2 * math.pi, dim=1).values
        r = 0.2 + 0.1 * torch.rand((B, 1), generator=g)
        pts = torch.stack([0.5 + r * torch.cos(th), 0.5 + r * torch.sin(th)], -1)          # (B, npts, 2) closed curve
        nrm = torch.stack([torch.cos(th), torch.sin(th)], -1)
        area = torch.rand((B, npts, 1), generator=g)
        nodes = torch.stack((m.xx, m.yy), 0)
        w = ibn2d.compute_winding_nodes(pts.unsqueeze(1), nrm.unsqueeze(1), area.unsqueeze(1), nodes)
        np.savez_compressed(os.path.join(outdir, f"winding_{tag}.npz"), n=n, points=T(pts), normals=T(nrm), area=T(area),
                            nodes=T(nodes), winding=T(w))
        print("winding", tag, tuple(w.shape), float(w.abs().max()))


def gen_fdm(outdir):
    import contextlib
    import io
    from DiffNet.DiffNetFDM import DiffNetFDM
    for n, B in [(16, 2), (33, 1)]:
        with contextlib.redirect_stdout(io.StringIO()):        # the reference prints its correction matrices
            m = DiffNetFDM(None, domain_size=n)
        g = rng(41)
        u = torch.rand((B, 1, n, n), generator=g)
        out = {"u": T(u), "keys": np.array(sorted(m.state_dict().keys()))}
        for name, pad in (("x", m.pad), ("y", m.pad), ("xx", m.pad_d2), ("yy", m.pad_d2)):
            ur = u.clone().requires_grad_(True)
            d = getattr(m, "derivative_" + name)(pad(ur))
            cot = torch.rand(d.shape, generator=g)
            (gu,) = torch.autograd.grad(d, ur, cot)
            out["d_" + name], out["cot_" + name], out["vjp_" + name] = T(d), T(cot), T(gu)
        for k in ("sobelx", "sobely", "sobelxx", "sobelyy", "h_corr", "v_corr", "h_corr_d2", "v_corr_d2"):
            out["par_" + k] = T(getattr(m, k))
        np.savez_compressed(os.path.join(outdir, f"fdm_n{n}.npz"), **out)
        print("fdm", n, float(np.abs(out["d_xx"]).max()))


def gen_datasets(outdir):
    """Datasets: first sample, length and auxiliary attributes of every reference dataset class; the
    file-based ones run on small synthetic files whose bytes are stored in the fixture so the test can recreate them."""
    import tempfile
    import PIL.Image
    from DiffNet.datasets.single_instances import rectangles, cuboids, circles, Lshaped, images as s_images, voxels, klsum as s_klsum
    from DiffNet.datasets.parametric import images as p_images, klsum as p_klsum
    from DiffNet import gen_input_calc
    out = {}

    def put(tag, ds, attrs=()):
        x, f = ds[min(1, len(ds) - 1)]
        out[tag + "/inputs"], out[tag + "/forcing"], out[tag + "/len"] = T(x), T(f), np.array(len(ds))
        for a in attrs:
            v = getattr(ds, a)
            out[tag + "/attr_" + a] = T(v) if isinstance(v, torch.Tensor) else np.asarray(v)

    n = 72          # larger than the hard-coded 64-grid offsets of the immersed shapes
    for name in ("Rectangle", "RectangleManufactured", "AdvDiff1dRectangle", "AdvDiff2dRectangle", "AllenCahnIceMeltRectangle",
                 "RectangleManufacturedNonZeroBC", "RectangleHelmholtzManufactured", "RectangleHelmholtzDeltaForce",
                 "RectangleManufacturedStokes", "RectangleIM", "RectangleIMBack"):
        attrs = {"AllenCahnIceMeltRectangle": ("u0", "initial_guess"), "RectangleManufacturedNonZeroBC": ("u_exact",)}.get(name, ())
        put("rect/" + name, getattr(rectangles, name)(domain_size=n), attrs)
    np.random.seed(1234)
    torch.manual_seed(1234)
    put("rect/SpaceTimeRectangleManufactured", rectangles.SpaceTimeRectangleManufactured(domain_size=24), ("u0", "initial_guess"))
    put("cuboid/Cuboid", cuboids.Cuboid(domain_size=12))
    put("cuboid/CuboidManufactured", cuboids.CuboidManufactured(domain_size=12))
    put("circle/CircleIMBack", circles.CircleIMBack(domain_size=n))
    put("lshaped/LShaped", Lshaped.LShaped(domain_size=n))
    g = np.random.RandomState(7)
    with tempfile.TemporaryDirectory() as tmp:
        # images: three small grey PNGs with blobs
        imgdir = os.path.join(tmp, "imgs")
        os.makedirs(imgdir)
        for k in range(3):
            a = np.zeros((20, 28), dtype=np.uint8)
            r0, c0 = g.randint(2, 8), g.randint(2, 12)
            a[r0:r0 + g.randint(3, 9), c0:c0 + g.randint(3, 12)] = g.randint(1, 255)
            path = os.path.join(imgdir, f"shape{k}.png")
            PIL.Image.fromarray(a).save(path)
            out[f"files/img{k}"] = np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
        put("img/single_ImageIMBack", s_images.ImageIMBack(os.path.join(imgdir, "shape1.png")))
        put("img/single_Disk", s_images.Disk(os.path.join(imgdir, "shape2.png")))
        for name in ("ImageIMBack", "ImageIMBackObject", "ImageIMBackNeumann"):
            put("img/param_" + name, getattr(p_images, name)(imgdir))
        # voxels: 9 x 7 x 5 object in a 48^3 background (offset 32 is hard-coded in the reference)
        vox = (g.rand(9, 7, 5) > 0.5).astype(np.uint8) * 254
        prefix = os.path.join(tmp, "obj_")
        vox.flatten(order="F").tofile(prefix + "inouts.raw")
        cfg = "voxel config\n0.0 0.0 0.0\n1.0 1.0 1.0\n9 7 5\n0.1 0.1 0.1\n100\n20\n"
        open(prefix + "VoxelConfig.txt", "w").write(cfg)
        out["files/vox_raw"] = np.frombuffer(open(prefix + "inouts.raw", "rb").read(), dtype=np.uint8)
        out["files/vox_cfg"] = np.frombuffer(cfg.encode(), dtype=np.uint8)
        put("vox/VoxelIMBackRAW", voxels.VoxelIMBackRAW(prefix, domain_size=48))
        # KL sums
        coeffs = g.randn(5, 6)
        np.save(os.path.join(tmp, "coeffs.npy"), coeffs)
        np.savetxt(os.path.join(tmp, "coeff.txt"), coeffs[3])
        out["files/kl_coeffs"] = coeffs
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            ks = p_klsum.KLSumStochastic(os.path.join(tmp, "coeffs.npy"), domain_size=20, kl_terms=6)
            ks4 = p_klsum.KLSumStochastic(os.path.join(tmp, "coeffs.npy"), domain_size=20, kl_terms=4)
        put("kl/param_KLSumStochastic", ks)
        out["kl/param_all"] = np.stack([T(ks[i][0]) for i in range(len(ks))])
        out["kl/param_terms4"] = np.stack([T(ks4[i][0]) for i in range(len(ks4))])
        put("kl/param_Dataset", p_klsum.Dataset(os.path.join(tmp, "coeff.txt"), domain_size=20))
        put("kl/single_Dataset", s_klsum.Dataset(os.path.join(tmp, "coeff.txt"), domain_size=20))
    # the PointClouds dataset of the flagship script (IBN_2D.py:35-84) on six synthetic closed curves
    sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
    from diffnet_amd.datasets.parametric.pointclouds import write_star_shapes
    ibn = load_script("IBN/poisson-2d/parametric/IBN_2D.py", "ref_ibn2d_ds")
    with tempfile.TemporaryDirectory() as tmp:
        raw, nrm = write_star_shapes(tmp + os.sep, n_shapes=6, n_points=40, seed=3)
        out["files/pc_points"], out["files/pc_normals"] = raw, nrm
        ds = ibn.PointClouds(tmp + os.sep, type='val', domain_size=24)
        x, f, snk = ds[4]
        out["pc/inputs"], out["pc/forcing"], out["pc/sink"], out["pc/len"] = T(x), T(f), T(snk), np.array(len(ds))
        out["pc/area"] = ds.area
    for eta in (0.1, 0.2, 0.5, 0.7, 1.0):
        out[f"kl/omega_{eta}"] = gen_input_calc.calculate_omega_based_on_eta(eta)
    out["kl/nu3d"] = gen_input_calc.generate_diffusivity_tensor(coeffs[0], output_size=6, nsd=3)
    np.savez_compressed(os.path.join(outdir, "datasets.npz"), **out)
    print("datasets", len(out), "arrays")


def exact_sines(*xs):
    """sin(pi x) sin(pi y) [sin(pi z)] for torch tensors (calc_l2_err) and numpy scalars (calc_l2_err_old) alike."""
    if torch.is_tensor(xs[0]):
        out = torch.sin(math.pi * xs[0])
        for x in xs[1:]:
            out = out * torch.sin(math.pi * x)
        return out
    out = np.sin(math.pi * xs[0])
    for x in xs[1:]:
        out = out * np.sin(math.pi * x)
    return out


def gen_l2(outdir):
    """`calc_l2_err` / `calc_l2_err_old` of the reference (DiffNetFEM.py:286-379, 482-591).  They only print, so the
    numbers are parsed from the captured stdout."""
    import contextlib
    import io
    import re
    from DiffNet.DiffNetFEM import DiffNet2DFEM, DiffNet3DFEM
    num = r"(?:tensor\()?([-+0-9.eE]+)"

    def parse(text):
        a = re.search(r"\|\|u_sol\|\|, \|\|uex\|\| =\s+" + num + r"\)?\s+" + num, text)
        e = re.search(r"\|\|e\|\|_\{\{L2\}\} =\s+" + num, text)   # the reference prints the doubled braces literally
        v = re.search(r"\(vector-norm\) =\s+" + num, text)
        return np.array([float(e.group(1)), float(a.group(1)), float(a.group(2)), float(v.group(1))])

    for tag, nsd, kw in [("2d_n17_g2", 2, dict(domain_size=17)), ("2d_n33_g3", 2, dict(domain_size=33, ngp_1d=3)),
                         ("2d_q2_n17", 2, dict(domain_size=17, fem_basis_deg=2)), ("3d_n9_g2", 3, dict(domain_size=9, nsd=3))]:
        m = (DiffNet2DFEM if nsd == 2 else DiffNet3DFEM)(None, **kw)
        m.exact_solution = exact_sines
        coords = [m.xx, m.yy] + ([m.zz] if nsd == 3 else [])
        u_ex = exact_sines(*coords)
        g = rng(53)
        u_sol = u_ex + 0.01 * (torch.rand(u_ex.shape, generator=g) - 0.5)
        out = dict(kwargs=repr(kw), u_sol=T(u_sol), u_exact=T(u_ex))
        torch.set_printoptions(precision=10)
        m.u_exact = u_ex.numpy()
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            m.calc_l2_err(u_sol[None, None])
        out["new"] = parse(buf.getvalue())                       # eL2, uL2, u_exL2, vector norm
        if kw.get("fem_basis_deg", 1) == 1 and kw.get("ngp_1d", 2) == 2:   # the old routine hard-codes Q1 / 2 points / unit square
            if nsd == 3:
                m.u_exact = u_ex                                 # the 3-D routine calls .squeeze().detach() on it
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                m.calc_l2_err_old(u_sol.numpy().astype(np.float64))
            out["old"] = parse(buf.getvalue())
        torch.set_printoptions(profile="default")
        np.savez_compressed(os.path.join(outdir, f"l2_{tag}.npz"), **out)
        print("l2", tag, out["new"], out.get("old"))


def gen_networks_large(outdir):
    """BASELINE configs[1]'s network at its mesh size: U-Net(2 -> 1) on one 512 x 512 sample, eval mode; outputs and
    gradients are stored on a strided subset (the full tensors would be megabytes of fixture)."""
    from DiffNet.networks.unets import UNet
    torch.manual_seed(2024)
    net = UNet(in_channels=2, out_channels=1).eval()
    sd = net.state_dict()
    checksum = np.array([float(sum(v.double().sum() for v in sd.values())), float(sum(v.double().abs().sum() for v in sd.values()))])
    n = 512
    yy, xx = torch.meshgrid(torch.linspace(0, 1, n), torch.linspace(0, 1, n), indexing="ij")
    x = torch.stack([0.5 + 0.4 * torch.sin(7 * xx + 3 * yy), (torch.cos(5 * xx * yy) > 0.3).float()], 0)[None].requires_grad_(True)
    y = net(x)
    cot = torch.cos(11 * xx - 4 * yy)[None, None]
    gx, = torch.autograd.grad(y, x, cot, retain_graph=True)
    first = list(net.parameters())[0]
    gw, = torch.autograd.grad(y, first, cot)
    st = 8
    np.savez_compressed(os.path.join(outdir, "net_unet_2_1_n512.npz"), n=n, stride=st, y=T(y)[..., ::st, ::st], grad_x=T(gx)[..., ::st, ::st],
                        grad_w0=T(gw), y_sum=float(y.double().sum()), gx_abs_sum=float(gx.double().abs().sum()),
                        checksum=checksum, torch_version=torch.__version__)
    print("net unet 512", float(y.mean()), float(gx.abs().max()))


if __name__ == "__main__":
    # default: tables, operators, loss bodies and networks; the other fixture families are selected by flag
    outdir = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    extra = {"--fdm": gen_fdm, "--winding": gen_winding, "--datasets": gen_datasets, "--l2": gen_l2, "--net512": gen_networks_large, "--netgrads": gen_network_grads}
    chosen = [fn for flag, fn in extra.items() if flag in sys.argv]
    if chosen:
        install_shims()
        for fn in chosen:
            fn(outdir)
    else:
        main()
