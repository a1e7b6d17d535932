"""GPU tests of the prepared launch (ops.PoissonPlan) and of the two ways the 3-D kernel sums the stiffness energy."""
import numpy as np
import pytest
import torch

from test_gpu_parity import boundary_mask, close, cu, dev, module, seeded

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kw,B", [(dict(domain_size=64, ngp_1d=3), 3), (dict(domain_size=33, nsd=3), 2)])
def test_prepared_launch_equals_the_one_shot_call_and_follows_its_buffers(kw, B):
    from diffnet_amd import ops
    m = module(kw)
    shape = (B, 1, *m.geom.node_shape)
    u, nu, f = cu(seeded(shape, 1)), cu(seeded(shape, 2) + 0.5), cu(seeded(shape, 3))
    bc = boundary_mask((1,) + shape[1:]).to(torch.uint8).to(dev())
    scale = 1.0 / (B * m.geom.nelem_total)
    kwargs = dict(alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=scale, want_out=True, want_sums=True, loss_scale=scale)
    ref = ops.poisson_apply(m.geom, u, nu, f, None, [(bc, 0.0)], **kwargs)
    plan = ops.PoissonPlan(m.geom, u, nu, f, None, [(bc, 0.0)], **kwargs)
    got = plan.launch()
    for a, b in zip(got, ref):
        assert torch.equal(a, b)
    # the plan reads its buffers at launch time: new values in place -> new results, same output tensors
    u.mul_(0.5)
    ref2 = ops.poisson_apply(m.geom, u, nu, f, None, [(bc, 0.0)], **kwargs)
    got2 = plan.launch()
    assert got2[0].data_ptr() == got[0].data_ptr()
    for a, b in zip(got2, ref2):
        assert torch.equal(a, b)
    # a plan with a reduction workspace belongs to the stream it was prepared on
    side = torch.cuda.Stream(dev())
    with torch.cuda.stream(side):
        with pytest.raises(Exception, match="another stream"):
            plan.launch()


def test_3d_energy_from_nodal_values_equals_the_gauss_point_sum():
    """sum_a u_a out_a = alpha * sum W nu |grad u|^2 - beta * sum W f u: the default 3-D kernel takes the stiffness energy from the
    finished nodal values; dn_config_set("Q1_3D_E1SUM") sums it Gauss point by Gauss point.  Same loss to fp32 rounding, same gradient."""
    from diffnet_amd import _lib
    m = module(dict(domain_size=65, nsd=3))
    shape = (2, 1, 65, 65, 65)
    u, nu, f = cu(seeded(shape, 11)), cu(seeded(shape, 12) + 0.5), cu(seeded(shape, 13))
    bc = boundary_mask((1,) + shape[1:]).to(torch.uint8).to(dev())
    src = (seeded(shape, 14) < 0.03).to(torch.uint8).to(dev())
    for c in (1.0, 0.5):
        la, ga = m.energy_loss_and_grad(u, nu, f, dirichlet=[(src, 1.0), (bc, 0.0)], c=c)
        _lib.config_set("Q1_3D_E1SUM", "1")
        try:
            lb, gb = m.energy_loss_and_grad(u, nu, f, dirichlet=[(src, 1.0), (bc, 0.0)], c=c)
        finally:
            _lib.config_set("Q1_3D_E1SUM", "")
        assert torch.equal(ga, gb)
        np.testing.assert_allclose(float(la), float(lb), rtol=3e-6)


@pytest.mark.parametrize("sizes,ngp,B", [((17, 23, 29), 2, 2), ((33, 18, 16), 3, 1), ((16, 16, 40), 2, 3), ((5, 4, 3), 2, 1)])
def test_3d_q1_marching_adjoint_equals_the_tiled_adjoint(sizes, ngp, B):
    """gauss_pt_eval VJP in 3-D Q1: the marching kernel (every Gauss-point value read once) against the tiled LDS kernel
    (dn_config_set("GPE_TILED")) on ragged, multi-tile, multi-strip meshes; both against autograd of the oracle's conv formulation."""
    from diffnet_amd import _lib
    from oracle.fem_oracle import Oracle
    kw = dict(nsd=3, domain_sizes=sizes, domain_lengths=(1.0, 1.3, 0.7), domain_size=sizes[0], ngp_1d=ngp)
    m, o = module(kw), Oracle(**kw)
    u = seeded((B, 1, sizes[2], sizes[1], sizes[0]), 5)
    for name in ("N_gp", "dN_x_gp", "dN_z_gp"):
        ur = u.clone().requires_grad_(True)
        y = o.ev(ur, name)
        cot = seeded(tuple(y.shape), 6)
        (ref,) = torch.autograd.grad(y, ur, cot)
        fn = {"N_gp": m.gauss_pt_evaluation, "dN_x_gp": m.gauss_pt_evaluation_der_x, "dN_z_gp": m.gauss_pt_evaluation_der_z}[name]
        got = {}
        for tiled in ("", "1"):
            _lib.config_set("GPE_TILED", tiled)
            try:
                ug = cu(u).requires_grad_(True)
                (g,) = torch.autograd.grad(fn(ug), ug, cu(cot))
            finally:
                _lib.config_set("GPE_TILED", "")
            got[tiled] = g
            close(g, ref.numpy(), rtol=1e-5, arel=2e-6, msg=f"{name} tiled={tiled!r}")
        close(got[""], got["1"].cpu().numpy(), rtol=2e-6, arel=2e-6)


@pytest.mark.parametrize("n,ngp,B", [(2, 2, 1), (3, 3, 2), (5, 2, 3), (18, 3, 2), (33, 2, 1), (64, 3, 2)])
def test_2d_paired_strips_any_strip_height(n, ngp, B):
    """The closed-form 2-D kernel runs neighbouring strips in opposite directions (odd strips in the mirrored row coordinate).  Which
    rows and layers a strip owns must not depend on its direction: every strip height -- one layer per strip, heights that do not
    divide the mesh, one strip for the whole mesh, odd and even strip counts -- gives the per-point kernel's loss and gradient."""
    from diffnet_amd import BoxFaces, _lib, ops
    m = module(dict(domain_size=n, ngp_1d=ngp))
    shape = (B, 1, n, n)
    u, nu, f = cu(seeded(shape, 21)), cu(seeded(shape, 22) + 0.5), cu(seeded(shape, 23))
    bc = boundary_mask((1,) + shape[1:]).to(torch.uint8).to(dev())
    src = (seeded(shape, 24) < 0.05).to(torch.uint8).to(dev())
    conds = {"none": [], "box": [(BoxFaces("all"), 0.0)], "u8 x2": [(src, 1.0), (bc, 0.0)]}
    ref = {}
    _lib.config_set("Q1_RULE_KERNEL", "1")
    try:
        for name, d in conds.items():
            ref[name] = m.energy_loss_and_grad(u, nu, f, dirichlet=d, c=0.7)
    finally:
        _lib.config_set("Q1_RULE_KERNEL", "")
    try:
        for R in (1, 2, 3, 5, 7, 16, 64):
            for E in (2, 4) if n % 4 == 0 else (2,):
                _lib.config_set("PLAN2D", f"64,{E},{R}")
                ops._POISSON_WS_BYTES.clear()                      # the workspace size depends on the launch plan
                for name, d in conds.items():
                    loss, grad = m.energy_loss_and_grad(u, nu, f, dirichlet=d, c=0.7)
                    l0, g0 = ref[name]
                    np.testing.assert_allclose(float(loss), float(l0), rtol=3e-6, err_msg=f"R={R} E={E} {name}")
                    scale = float(g0.abs().max()) + 1e-30
                    assert float((grad - g0).abs().max()) <= 1e-5 * scale, f"R={R} E={E} {name}"
    finally:
        _lib.config_set("PLAN2D", "")
        ops._POISSON_WS_BYTES.clear()


@pytest.mark.parametrize("sizes,ngp,B", [((8, 8), 2, 1), ((64, 64), 3, 2), ((64, 40), 2, 3), ((516, 37), 3, 1), ((512, 512), 3, 2)])
def test_2d_chained_strips_equal_single_strips(sizes, ngp, B):
    """Launches that fill the chip chain several neighbouring strips per workgroup (poisson2d_q1_cf.hip, W > 1): a strip takes the node row it
    shares with its neighbour, and the neighbour's contributions to it, through LDS instead of re-reading the row and recomputing the
    seam layer.  Forced here on small meshes for every strip height class -- strips of 2 rows, heights that do not divide the mesh, a
    last workgroup with fewer strips than the chain, one strip only, two x-chunks -- with every coefficient / condition combination the
    kernel is instantiated for; against one strip per workgroup (same kernel) and the per-point kernel."""
    from diffnet_amd import BoxFaces, PackedMask, _lib, ops
    kw = dict(nsd=2, domain_sizes=sizes, domain_lengths=(1.0, 0.8), domain_size=sizes[0], ngp_1d=ngp)
    m = module(kw)
    shape = (B, 1, sizes[1], sizes[0])
    u, nu, f = cu(seeded(shape, 31)), cu(seeded(shape, 32) + 0.5), cu(seeded(shape, 33))
    bc = boundary_mask((1,) + shape[1:]).to(torch.uint8).to(dev())
    src = (seeded(shape, 34) < 0.05).to(torch.uint8).to(dev())
    cases = {"none": (nu, f, []), "box": (nu, f, [(BoxFaces("all"), 0.0)]), "bits+box": (nu, f, [(PackedMask.pack(src), 1.0), (BoxFaces("all"), 0.0)]),
             "u8 x2": (nu, f, [(src, 1.0), (bc, 0.0)]), "f32": (nu, f, [(bc.float(), 0.0)]), "no nu": (None, f, [(bc, 0.0)]),
             "no f": (nu, None, [(PackedMask.pack(bc.expand(shape).contiguous()), 0.0)]), "bare": (None, None, [(BoxFaces("all"), 0.0)])}
    ref = {}
    _lib.config_set("Q1_RULE_KERNEL", "1")
    try:
        for name, (a, b, d) in cases.items():
            if not any(isinstance(x[0], (BoxFaces, PackedMask)) for x in d):
                ref[name] = m.energy_loss_and_grad(u, a, b, dirichlet=d, c=0.7)
    finally:
        _lib.config_set("Q1_RULE_KERNEL", "")
    try:
        for R in (0, 2, 3, 5, 9, 16, 40, 61):        # 0: the plan the library picks by itself (chained strips for launches that do not fill the chip) against 16-row strips
            res = {}
            for W in (1, 2):
                _lib.config_set("PLAN2D", f"128,4,{R},{W}" if R else ("128,4,16,1" if W == 1 else ""))
                ops._POISSON_WS_BYTES.clear()
                for name, (a, b, d) in cases.items():
                    res[name, W] = m.energy_loss_and_grad(u, a, b, dirichlet=d, c=0.7)
            for name in cases:
                (l1, g1), (l2, g2) = res[name, 1], res[name, 2]
                scale = float(g1.abs().max()) + 1e-30
                np.testing.assert_allclose(float(l2), float(l1), rtol=2e-6, err_msg=f"R={R} {name}")
                assert float((g2 - g1).abs().max()) <= 2e-6 * scale, f"R={R} {name}"
                if name in ref:
                    l0, g0 = ref[name]
                    np.testing.assert_allclose(float(l2), float(l0), rtol=3e-6, err_msg=f"R={R} {name} vs per-point kernel")
                    assert float((g2 - g0).abs().max()) <= 1e-5 * scale, f"R={R} {name} vs per-point kernel"
    finally:
        _lib.config_set("PLAN2D", "")
        ops._POISSON_WS_BYTES.clear()


@pytest.mark.parametrize("sizes,B", [((321, 9), 3), ((513, 40), 2), ((385, 17), 1), ((1025, 5), 2), ((325, 33), 2)])
def test_2d_rows_of_4k_plus_1_nodes_on_the_vector_kernel(sizes, B):
    """Rows of 4 k + 1 nodes (>= 321): the closed-form kernel runs them four elements per thread with 16-byte accesses on 4-byte aligned rows, and the
    mesh's last node column is finished by the last full thread column (CF_UA).  Against the per-point marching kernels
    (dn_config_set("Q1_RULE_KERNEL"), two elements per thread, scalar accesses): energy, gradient, residual and its sum of squares, every mask format,
    one and two conditions, value fields, launch-plan overrides (several chunks per row, strips of 1 .. 32 rows)."""
    from diffnet_amd import BoxFaces, PackedMask, _lib
    m = module(dict(nsd=2, domain_sizes=sizes, domain_lengths=(1.0, 0.6), domain_size=sizes[0], ngp_1d=3))
    shape = (B, 1, sizes[1], sizes[0])
    u, nu, f = cu(seeded(shape, 21)), cu(seeded(shape, 22) + 0.5), cu(seeded(shape, 23))
    bc = boundary_mask(shape).to(torch.uint8).to(dev())
    bc[:, 0, sizes[1] // 2, -7:] = 1                      # (Dirichlet nodes in the last columns, the last one included)
    src = (seeded(shape, 24) < 0.1).to(torch.uint8).to(dev())
    src[..., -1] = (seeded(shape[:-1], 25) < 0.5).to(torch.uint8).to(dev())
    field = cu(seeded(shape, 26))
    conds = {"none": [], "bits": [(PackedMask.pack(bc), 0.25)], "u8 x2": [(src, 1.0), (bc, 0.0)], "f32": [(bc.float(), -0.5)], "box": [(BoxFaces("all"), 0.0)],
             "value field": [(bc, field)], "bits x2": [(PackedMask.pack(src), 1.0), (PackedMask.pack(bc), 0.0)]}
    try:
        for name, d in conds.items():
            _lib.config_set("Q1_RULE_KERNEL", "1")
            l0, g0 = m.energy_loss_and_grad(u, nu, f, dirichlet=d, c=0.7)
            R0 = m.residual(u, nu, f, dirichlet=d)
            r0 = m.residual_loss(u, nu, f, dirichlet=d)
            _lib.config_set("Q1_RULE_KERNEL", "")
            for plan in ("", "64,4,1", "128,4,3", "256,4,32", "192,4,5"):
                _lib.config_set("PLAN2D", plan)
                l1, g1 = m.energy_loss_and_grad(u, nu, f, dirichlet=d, c=0.7)
                R1 = m.residual(u, nu, f, dirichlet=d)
                r1 = m.residual_loss(u, nu, f, dirichlet=d)
                assert float((g1 - g0).abs().max()) <= 3e-6 * float(g0.abs().max()), f"{name} plan {plan!r}"
                assert float((R1 - R0).abs().max()) <= 3e-6 * float(R0.abs().max()), f"{name} plan {plan!r}"
                np.testing.assert_allclose(float(l1), float(l0), rtol=3e-6, err_msg=f"{name} plan {plan!r}")
                np.testing.assert_allclose(float(r1), float(r0), rtol=3e-6, err_msg=f"{name} plan {plan!r} residual loss")
    finally:
        _lib.config_set("PLAN2D", "")
        _lib.config_set("Q1_RULE_KERNEL", "")
