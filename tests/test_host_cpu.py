"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol declared in
include/diffnet_hip.h, host-built tables are bit-identical to the reference's, the module surface
(attributes, shapes, state_dict keys, kwargs handling) matches, and the ops refuse CPU tensors."""
import glob
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT
from test_oracle_golden import spec_kwargs

FEM_FILES = sorted(glob.glob(os.path.join(GOLDEN, "fem_*.npz")))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "diffnet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dn_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_the_whole_abi():
    from diffnet_amd import _lib, build
    build.build(verbose=False)
    h = _lib.lib()
    syms = declared_symbols()
    assert len(syms) >= 8
    assert sorted(_lib.SYMBOLS) == syms, "ctypes binding and header disagree"
    for s in syms:
        assert hasattr(h, s), s
    assert h.dn_abi_version() == _lib.ABI_VERSION
    assert b"gfx950" in h.dn_build_info()


def test_launch_planner_rejects_bad_meshes_without_a_gpu():
    import ctypes as C
    from diffnet_amd import _lib
    from diffnet_amd.fem import FemGeometry
    from diffnet_amd.tables import gauss_rule
    h = _lib.lib()
    gx, gw = gauss_rule(2)
    m = FemGeometry(2, (64, 64), (1 / 63, 1 / 63), 1, 2, gx, gw).mesh_struct(4)
    assert h.dn_poisson_workspace_bytes(C.byref(m)) > 0
    m.nx = 1
    assert h.dn_poisson_workspace_bytes(C.byref(m)) == -1
    # 3-D Q2 / Q3 (element vectors + gather assembly): header + partial sums of its two kernels + (P+1)^3 floats per element
    m.nsd, m.degree, m.ngp, m.batch, m.nx, m.ny, m.nz = 3, 2, 3, 2, 9, 7, 5
    nel, nnode = 4 * 3 * 2, 9 * 7 * 5
    assert h.dn_poisson_workspace_bytes(C.byref(m)) == 64 * 65 + 8 * (-(-nel * 2 // 64) + -(-nnode * 2 // 256)) + 4 * 27 * nel * 2
    m.nx = 10                                                # (nx - 1) not a multiple of the degree
    assert h.dn_poisson_workspace_bytes(C.byref(m)) == -1
    assert h.dn_poisson_apply(C.byref(m), None, None) == -1          # validation happens before any launch


@pytest.mark.parametrize("path", FEM_FILES, ids=[os.path.basename(p)[4:-4] for p in FEM_FILES])
def test_module_surface_matches_reference(path):
    from diffnet_amd import DiffNet2DFEM, DiffNet3DFEM
    z = np.load(path)
    kw = spec_kwargs(z)
    m = (DiffNet3DFEM if kw["nsd"] == 3 else DiffNet2DFEM)(None, **kw)
    assert sorted(m.state_dict().keys()) == list(z["state_dict_keys"])
    for key in z.files:
        if key.startswith("scalar_"):
            np.testing.assert_allclose(float(getattr(m, key[7:])), float(z[key]), rtol=1e-6, err_msg=key)
        elif key.startswith("tab_"):
            got = np.stack([p.detach().numpy() for p in getattr(m, key[4:])], 0)
            assert got.shape == z[key].shape, key
            assert np.array_equal(got, z[key]), key                     # bit identical
        elif key.startswith("attr_"):
            got = getattr(m, key[5:]).numpy()
            assert got.shape == z[key].shape, key
            if key[5:] in ("xgp", "ygp", "zgp"):
                np.testing.assert_allclose(got, z[key], rtol=1e-6, atol=1e-6)
            else:
                assert np.array_equal(got, z[key]), key
    np.testing.assert_array_equal(m.gpx_1d, z["gpx_1d"])
    np.testing.assert_array_equal(m.gpw_1d, z["gpw_1d"])
    for p in m.parameters():
        assert not p.requires_grad
    x = np.array([-0.3, 0.1, 0.9])
    assert m.bf_1d(x).shape == (m.nbf_1d, 3) and np.allclose(m.bf_1d(x).sum(0), 1.0)
    assert np.allclose(m.bf_1d_der(x).sum(0), 0.0, atol=1e-12)
    if m.fem_basis_deg <= 2:
        tx = torch.tensor(x)
        np.testing.assert_allclose(m.bf_1d_th(tx).numpy(), m.bf_1d(x), rtol=1e-12)
        np.testing.assert_allclose(m.bf_1d_der_th(tx).numpy(), m.bf_1d_der(x), rtol=1e-12)
        np.testing.assert_allclose(m.bf_1d_der2_th(tx).numpy(), m.bf_1d_der2(x), rtol=1e-12)


def test_pde_kwargs_and_legacy_constructor():
    from diffnet_amd import PDE, DiffNet2DFEM, DiffNet3DFEM
    net = torch.nn.Linear(2, 2)
    p = PDE(net)
    assert (p.nsd, p.batch_size, p.n_workers, p.learning_rate, p.domain_size, p.domain_length) == (2, 64, 1, 3e-4, 64, 1.0)
    p.log("loss", 1.0)
    opts, sch = p.configure_optimizers()
    assert isinstance(opts[0], torch.optim.Adam) and sch == []
    with pytest.raises(NotImplementedError):
        p.loss(None, None, None)
    ds = object()
    m = DiffNet2DFEM(net, ds, domain_size=9)          # legacy 2-positional form (tests/test.py:33 of the reference)
    assert m.dataset is ds and m.network is net
    m3 = DiffNet3DFEM(None, nsd=3, domain_sizes=(10, 8, 6), domain_lengths=(2.0, 1.0, 0.5), domain_size=10)
    assert (m3.nelemX, m3.nelemY, m3.nelemZ) == (9, 7, 5) and m3.xx.shape == (6, 8, 10)
    assert m3.geom.node_shape == (6, 8, 10) and m3.geom.elem_shape == (5, 7, 9)
    with pytest.raises(AssertionError):
        DiffNet2DFEM(None, nsd=3, domain_size=9)
    with pytest.raises(AssertionError):
        DiffNet2DFEM(None, domain_size=10, fem_basis_deg=2)
    assert DiffNet2DFEM(None, domain_size=9, fem_basis_deg=2, ngp_1d=2).ngp_1d == 3     # raised to the degree's minimum


def test_ops_refuse_cpu_tensors():
    from diffnet_amd import DiffNet2DFEM, gauss_pt_eval
    from diffnet_amd._lib import DiffNetHipError
    m = DiffNet2DFEM(None, domain_size=9)
    u = torch.rand(1, 1, 9, 9)
    with pytest.raises(DiffNetHipError):
        m.gauss_pt_evaluation(u)
    with pytest.raises(DiffNetHipError):
        m.energy_loss(u)
    with pytest.raises(DiffNetHipError):
        gauss_pt_eval(u, m.N_gp, nsd=2, stride=1)
    with pytest.raises(UnboundLocalError):
        gauss_pt_eval(u, m.N_gp, nsd=4)


def test_reference_import_paths():
    import DiffNet
    from DiffNet.base import PDE
    from DiffNet.cuboid_mesh import CuboidMesh
    from DiffNet.DiffNetFEM import DiffNet2DFEM, DiffNet3DFEM, DiffNetFEM, gauss_pt_eval
    assert issubclass(DiffNet2DFEM, DiffNetFEM) and issubclass(DiffNetFEM, PDE)
    x, y, zz = CuboidMesh.meshgrid_3d(np.arange(4.0), np.arange(3.0), np.arange(2.0))
    assert x.shape == (2, 3, 4) and x[1, 2, 3] == 3 and y[1, 2, 3] == 2 and zz[1, 2, 3] == 1


@pytest.mark.parametrize("n", [16, 33])
def test_fdm_module_surface_matches_reference(n):
    from DiffNet.DiffNetFDM import DiffNetFDM
    z = np.load(os.path.join(GOLDEN, f"fdm_n{n}.npz"))
    m = DiffNetFDM(None, domain_size=n)
    assert sorted(m.state_dict().keys()) == list(z["keys"])
    for k in ("sobelx", "sobely", "sobelxx", "sobelyy", "h_corr", "v_corr", "h_corr_d2", "v_corr_d2"):
        assert np.array_equal(getattr(m, k).numpy(), z["par_" + k]), k
    assert isinstance(m.pad, torch.nn.ReplicationPad2d) and m.nsd == 2 and m.stencil_len == 3
    with pytest.raises(AttributeError):
        m.calc_laplacian(None)


def test_c_abi_rejects_bad_arguments_before_touching_the_gpu():
    """Every entry point validates its arguments on the host and returns DN_E_* without launching anything (so this runs
    without a GPU): null pointers, non-positive sizes, unsupported combinations, missing workspace."""
    import ctypes as C
    from diffnet_amd import _lib
    L = _lib.lib()
    null = None
    i3 = _lib.I32x3(8, 8, 1)
    BADARG, UNSUPPORTED, WORKSPACE = -1, -2, -3
    assert L.dn_gauss_pt_eval_fwd(null, null, null, 1, 2, i3, 2, 1, 4, null) == BADARG
    assert L.dn_gauss_pt_eval_bwd(null, null, null, 1, 2, i3, 2, 1, 4, null) == BADARG
    assert L.dn_gauss_pt_eval_fwd(null, null, null, 1, 2, i3, 7, 1, 4, null) == BADARG          # nbf out of range
    assert L.dn_assemble(null, null, 1, 2, i3, 2, 1, 0, null) == BADARG
    assert L.dn_assemble_bwd(null, null, 1, 2, i3, 2, 1, null) == BADARG
    assert L.dn_winding_nodes(null, null, null, null, 1, 10, 8, 8, null) == BADARG
    assert L.dn_fdm_stencil_fwd(null, null, 1, 8, 8, null, 3, 1.0, 1.0, null) != 0
    assert L.dn_instnorm_act_fwd(null, null, null, null, 4, 16, 1e-5, 0.2, null, 0, null) == BADARG
    assert L.dn_instnorm_act_bwd(null, null, null, null, null, 4, 16, 0.2, 4, 0, null, 0, null) == BADARG
    assert L.dn_instnorm_workspace_bytes(0, 16) == BADARG and L.dn_instnorm_workspace_bytes(4096, 64) == 0
    assert L.dn_instnorm_workspace_bytes(4, 1 << 20) > 0                                      # few large instances: sliced path
    assert L.dn_upconv_out_workspace_bytes(0, 64, 8, 8) == BADARG and L.dn_upconv_out_workspace_bytes(2, 64, 8, 8) > 0
    assert L.dn_upconv_out_fwd(null, null, null, null, 2, 64, 8, 8, 1, null, 0, null) == WORKSPACE
    assert L.dn_upconv_out_bwd(null, null, null, null, null, null, null, 2, 64, 8, 8, 1, null, 0, null) == BADARG
    assert L.dn_upconv3d_out_workspace_bytes(1, 32, 0, 8, 8) == BADARG and L.dn_upconv3d_out_workspace_bytes(1, 32, 8, 8, 8) > 0
    assert L.dn_upconv3d_out_fwd(null, null, null, null, 1, 32, 8, 8, 8, 1, null, 0, null) == BADARG
    assert L.dn_upconv3d_out_bwd(null, null, null, null, null, null, null, 1, 32, 8, 8, 8, 1, null, 0, null) == BADARG
    assert L.dn_conv3d_k4s2_wrw_workspace_bytes(1, 16, 200, 8, 8, 8) == UNSUPPORTED               # more than 128 coarse channels
    assert L.dn_conv3d_k4s2_wrw_workspace_bytes(1, 16, 32, 2, 2, 2) == 0                         # single workgroup: no partials
    assert L.dn_conv3d_k4s2_wrw(null, null, null, 1, 16, 32, 8, 8, 8, null, 0, null) == BADARG
    m = _lib.DnMesh()
    m.nsd, m.degree, m.ngp, m.batch, m.nx, m.ny, m.nz = 2, 1, 2, 1, 9, 9, 1
    assert L.dn_poisson_workspace_bytes(C.byref(m)) > 0 and L.dn_fsdt_workspace_bytes(C.byref(m)) > 0
    a = _lib.DnPoissonArgs()
    assert L.dn_poisson_apply(C.byref(m), C.byref(a), null) == BADARG                          # no u
    fa = _lib.DnFsdtArgs()
    assert L.dn_fsdt_apply(C.byref(m), C.byref(fa), null) == BADARG
    m.nsd = 3
    m.nz = 9
    assert L.dn_fsdt_workspace_bytes(C.byref(m)) == BADARG                                     # plate kernel is 2-D
    m.degree = 2
    a.u = 1; a.out = 1
    assert L.dn_poisson_apply(C.byref(m), C.byref(a), null) == UNSUPPORTED                     # fused 3-D is Q1 only


def test_torch_library_registration_and_fake_shapes():
    """The hot-path operators are registered in the `diffnet_mi` namespace; their fake implementations propagate
    shapes on fake GPU tensors without any device, which is what torch.compile / torch.export tracing relies on."""
    import diffnet_amd.torch_ops  # noqa: F401
    from torch._subclasses.fake_tensor import FakeTensorMode
    from diffnet_amd import DiffNet3DFEM
    for name in ("gauss_pt_eval_fwd", "gauss_pt_eval_bwd", "assemble", "assemble_bwd", "poisson_apply"):
        assert hasattr(torch.ops.diffnet_mi, name), name
    schema = str(torch.ops.diffnet_mi.poisson_apply.default._schema)
    assert "Tensor? nu" in schema and "-> (Tensor, Tensor, Tensor)" in schema
    m = DiffNet3DFEM(None, domain_size=9, nsd=3)
    with FakeTensorMode():
        u = torch.empty(2, 1, 9, 9, 9, device="cuda")
        t = torch.empty(8, 8, device="cuda")
        y = torch.ops.diffnet_mi.gauss_pt_eval_fwd(u, t, 3, 2, 1)
        assert tuple(y.shape) == (2, 8, 8, 8, 8) and y.device.type == "cuda"
        back = torch.ops.diffnet_mi.gauss_pt_eval_bwd(y, t, [2, 1, 9, 9, 9], 3, 2, 1)
        assert tuple(back.shape) == (2, 1, 9, 9, 9)
        a = torch.ops.diffnet_mi.assemble(y, 3, 2)
        assert tuple(a.shape) == (2, 1, 9, 9, 9)
        from diffnet_amd import torch_ops
        out, sums, loss = torch.ops.diffnet_mi.poisson_apply(u, None, None, None, None, None, 0.0, None, None, 0.0,
                                                             *torch_ops.geometry_args(m.geom), 1.0, 1.0, 0.5, 1.0, 1.0, 1.0)
        assert out.shape == u.shape and tuple(sums.shape) == (2,) and sums.dtype == torch.float64 and loss.dim() == 0


def test_closed_form_3d_kernel_lds_waits_and_prototype():
    """The closed-form 3-D Q1 kernel (csrc/poisson3d_q1_cf.hip) reads its node pairs from LDS with inline asm and writes the `s_waitcnt lgkmcnt`
    for them by hand; tools/check_lds_waits.py disassembles every instantiation (hipcc cross-compiles without a GPU) and verifies that no
    register of an LDS read is used before a covering wait.  tools/q1cf3d_proto.py is the float64 restatement of the kernel's formulas
    (monomial stages, closed-form z integration, mass-stencil forcing) checked against the oracle -- DiffNetFEM.py:7-18 + IBN_3D.py:114-136."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_lds_waits.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "violations: 0" in r.stdout and "instantiations: 48" in r.stdout
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "q1cf3d_proto.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
