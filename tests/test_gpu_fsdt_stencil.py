"""GPU tests of the assembled-stencil form of the fused FSDT plate residuals (diffnet_amd/csrc/fsdt_st.hip, the default of dn_fsdt_apply
since round 4): against the oracle's reference formulation (oracle/fem_oracle.py, pinned to the reference's fixtures by
tests/test_oracle_golden.py -- examples/elasticity/single_instance/e1_plate_bending_fsdt.py:128-232) and against the element form of
rounds 1-3 (dn_config_set("FSDT_FORM", "elem")), on the shapes where the two differ in structure: element-column counts around the wave
chunk of 62 owner lanes, one to four waves per workgroup, strips of every height, Q1 / Q2 / Q3, every rule, every Dirichlet form."""
import numpy as np
import pytest
import torch

from test_gpu_parity import boundary_mask, close, cu, dev, module, seeded

pytestmark = pytest.mark.gpu

CONSTS = dict(D11=1.3, D12=0.4, D22=1.1, D66=0.6, A44=0.8, A55=0.9, q=1.2, wscale=0.3)


def cfg(key, value):
    """A launch-plan switch changes the number of workgroups: prepared calls and workspace sizes of the earlier plan are dropped."""
    from diffnet_amd import _lib, ops
    _lib.config_set(key, value)
    ops.call_cache_clear()


def both_forms(fn):
    """(element form, stencil form) -- the stencil form asked for explicitly: Q3 meshes run the element form by default"""
    cfg("FSDT_FORM", "elem")
    try:
        ref = fn()
        cfg("FSDT_FORM", "stencil")
        got = fn()
    finally:
        cfg("FSDT_FORM", "")
    return ref, got


# element columns per row: 61, 62, 63 (one chunk / the closing column on a ghost lane / two chunks), 124, 125 (two / three chunks), 187
@pytest.mark.parametrize("deg,ngp,nelx,nely,B", [(2, 3, 61, 5, 2), (2, 3, 62, 3, 1), (2, 3, 63, 4, 2), (2, 2, 124, 2, 1), (2, 4, 125, 3, 1), (2, 3, 187, 7, 1),
                                                 (1, 2, 62, 9, 2), (1, 3, 63, 6, 1), (1, 4, 130, 5, 1), (3, 3, 62, 3, 1), (3, 4, 63, 4, 2), (3, 3, 125, 2, 1),
                                                 (2, 3, 1, 1, 1), (1, 2, 1, 1, 2), (3, 4, 2, 1, 1), (2, 3, 3, 40, 1)])
def test_stencil_form_equals_element_form(deg, ngp, nelx, nely, B):
    from diffnet_amd import _lib, ops
    sizes = (deg * nelx + 1, deg * nely + 1)
    m = module(dict(nsd=2, domain_sizes=sizes, domain_lengths=(1.0, 0.8), domain_size=sizes[0], fem_basis_deg=deg, ngp_1d=ngp))
    shape = (B, 1, sizes[1], sizes[0])
    flds = [cu(seeded(shape, 40 + i)) for i in range(3)]
    bcm = boundary_mask(shape).to(dev())
    if sizes[0] > 12 and sizes[1] > 4:
        bcm[0, 0, sizes[1] // 2, 3:9] = 1.0
    wbc = cu(seeded(shape, 50))
    try:
        for mask in (None, bcm, bcm.to(torch.uint8)):
            for plan in ("", "64,1", "128,2", "192,3", "256,5", "64,64"):
                cfg("PLAN_FSDT", plan)
                (ref, rsums, rnorms), (got, sums, norms) = both_forms(
                    lambda: ops.fsdt_apply(m.geom, *flds, mask, (0.1, -0.2, 0.3), want_norms=True, **CONSTS))
                for k in range(3):
                    scale = float(ref[k].abs().max())
                    assert float((got[k] - ref[k]).abs().max()) <= 4e-6 * scale, f"plan {plan!r} field {k} mask {None if mask is None else mask.dtype}"
                np.testing.assert_allclose(sums.cpu().numpy(), rsums.cpu().numpy(), rtol=2e-6)
                np.testing.assert_allclose(norms.cpu().numpy(), np.sqrt(rsums.cpu().numpy()), rtol=2e-6)
            if mask is not None:
                cfg("PLAN_FSDT", "")
                # Dirichlet values as fields (one batched, one broadcast over the batch) next to a constant
                bcs = (wbc, -0.2, wbc[:1] * 0.5)
                (ref, _), (got, _) = both_forms(lambda: ops.fsdt_apply(m.geom, *flds, mask, bcs, **CONSTS))
                for k in range(3):
                    assert float((got[k] - ref[k]).abs().max()) <= 4e-6 * float(ref[k].abs().max()), f"value fields, field {k}"
    finally:
        cfg("PLAN_FSDT", "")
        cfg("FSDT_FORM", "")


@pytest.mark.parametrize("deg,ngp,n", [(1, 2, 40), (2, 3, 129), (2, 2, 33), (3, 4, 190)])
def test_stencil_form_vs_oracle(deg, ngp, n):
    """The stencil form against the oracle's per-Gauss-point reference formulation and its autograd VJP (two chunks / four at Q3)."""
    from diffnet_amd.elasticity import fsdt_loss, fsdt_residuals
    from oracle.fem_oracle import Oracle
    cfg("FSDT_FORM", "stencil")
    try:
        _stencil_vs_oracle(deg, ngp, n)
    finally:
        cfg("FSDT_FORM", "")


def _stencil_vs_oracle(deg, ngp, n):
    from diffnet_amd.elasticity import fsdt_loss, fsdt_residuals
    from oracle.fem_oracle import Oracle
    kw = dict(domain_size=n, fem_basis_deg=deg, ngp_1d=ngp)
    m, o = module(kw), Oracle(**kw)
    shape = (1, 1, n, n)
    flds = [seeded(shape, 20 + i) for i in range(3)]
    bc = boundary_mask(shape)
    par = dict(E=2.0, v=0.3, q=1.5)
    ref_in = [t.clone().requires_grad_(True) for t in flds]
    Rref = o.fsdt_residuals(*ref_in, bc, th=0.2, Ks=5.0 / 6.0, **par)
    gpu_in = [t.to(dev()).requires_grad_(True) for t in flds]
    R = fsdt_residuals(m, *gpu_in, bc.to(dev()), h=0.2, K_s=5.0 / 6.0, **par)
    for a, b in zip(R, Rref):
        close(a, b.detach().numpy(), rtol=1e-4, arel=2e-5)
    cots = [seeded(shape, 30 + i) for i in range(3)]
    gref = torch.autograd.grad(Rref, ref_in, cots)
    g = torch.autograd.grad(R, gpu_in, [c.to(dev()) for c in cots])
    for a, b in zip(g, gref):
        close(a, b.numpy(), rtol=1e-4, arel=1e-4)
    norms = fsdt_loss(m, *gpu_in, bc.to(dev()), h=0.2, K_s=5.0 / 6.0, **par)
    for nv, b in zip(norms, Rref):
        np.testing.assert_allclose(float(nv), float(torch.linalg.vector_norm(b.double())), rtol=2e-5)


def test_stencil_form_full_size_properties():
    """configs[4] of BASELINE.json at its full size (1025 x 1025 nodes = 512 x 512 Q2 elements, 3 x 3 points): bitwise repeatable; Dirichlet rows carry
    the boundary values; the homogeneous operator is symmetric (<a, K b> == <K a, b>: it is the Hessian of the plate energy); the same numbers as the
    element form and as a differently partitioned launch."""
    from diffnet_amd import _lib, ops
    n = 1025
    m = module(dict(domain_size=n, fem_basis_deg=2, ngp_1d=3))
    shape = (1, 1, n, n)
    a = [cu(seeded(shape, 1 + i)) for i in range(3)]
    b = [cu(seeded(shape, 11 + i)) for i in range(3)]
    bc = boundary_mask(shape).to(torch.uint8).to(dev())
    consts = dict(CONSTS, q=0.0)
    Ka, _ = ops.fsdt_apply(m.geom, *a, bc, (0.0, 0.0, 0.0), **consts)
    Ka2, _ = ops.fsdt_apply(m.geom, *a, bc, (0.0, 0.0, 0.0), **consts)
    assert all(torch.equal(x, y) for x, y in zip(Ka, Ka2))
    Kb, _ = ops.fsdt_apply(m.geom, *b, bc, (0.0, 0.0, 0.0), **consts)
    free = (bc == 0).double()
    lhs = sum(float((x.double() * y.double() * free).sum()) for x, y in zip(a, Kb))
    rhs = sum(float((x.double() * y.double() * free).sum()) for x, y in zip(Ka, b))
    # (the masked rows / columns: boundary values 0 make K act on the free nodes only)
    assert abs(lhs - rhs) <= 2e-5 * max(abs(lhs), abs(rhs)), (lhs, rhs)
    for k in range(3):
        assert float((Ka[k] * bc).abs().max()) == 0.0
    try:
        cfg("PLAN_FSDT", "128,7")
        Kp, _ = ops.fsdt_apply(m.geom, *a, bc, (0.0, 0.0, 0.0), **consts)
        cfg("PLAN_FSDT", "")
        cfg("FSDT_FORM", "elem")
        Ke, _ = ops.fsdt_apply(m.geom, *a, bc, (0.0, 0.0, 0.0), **consts)
    finally:
        cfg("PLAN_FSDT", "")
        cfg("FSDT_FORM", "")
    for k in range(3):
        scale = float(Ke[k].abs().max())
        assert float((Ka[k] - Ke[k]).abs().max()) <= 4e-6 * scale
        assert torch.equal(Ka[k], Kp[k])          # strips of another height: the seam layer is recomputed exactly


@pytest.mark.parametrize("deg,ngp,n,B,form", [(2, 3, 129, 2, ""), (2, 3, 129, 2, "elem"), (1, 2, 70, 1, ""), (3, 4, 64, 1, ""), (2, 3, 1025, 1, "")])
def test_deferred_norms_equal_in_kernel_norms(deg, ngp, n, B, form):
    """dn_fsdt_args.defer_sums / den_workspace (round 4): the residual launch leaves per-workgroup partials, the VJP launch forms the norms from
    them.  Same partials, same order of additions as the in-kernel reduction: norms and gradient are BITWISE those of the pair with in-kernel norms
    (what rounds 2-3 ran); both kernel forms; through ops.fsdt_apply, elasticity.fsdt_loss_and_grad and the prepared FsdtPlan."""
    from diffnet_amd import ops
    from diffnet_amd.elasticity import fsdt_loss_and_grad
    m = module(dict(domain_size=n, fem_basis_deg=deg, ngp_1d=ngp))
    shape = (B, 1, n, n)
    flds = [cu(seeded(shape, 5 + i)) for i in range(3)]
    bc = boundary_mask(shape).to(dev())
    wts = torch.tensor([1.0, 2.0, 0.5], device=dev())
    cfg("FSDT_FORM", form or "stencil")
    try:
        R, _, norms = ops.fsdt_apply(m.geom, *flds, bc, (0.1, -0.2, 0.3), want_sums=False, want_norms=True, **CONSTS)
        vconsts = dict(CONSTS, q=0.0)
        g, _ = ops.fsdt_apply(m.geom, *R, bc, (0.0, 0.0, 0.0), want_sums=False, in_num=wts, in_den=norms, **vconsts)
        R2, _, h = ops.fsdt_apply(m.geom, *flds, bc, (0.1, -0.2, 0.3), want_sums=False, defer_norms=True, **CONSTS)
        g2, _, norms2 = ops.fsdt_apply(m.geom, *R2, bc, (0.0, 0.0, 0.0), want_sums=False, want_norms=True, in_num=wts, norms_from=h, **vconsts)
        assert torch.equal(norms, norms2)
        for a, b in zip(R + g, R2 + g2):
            assert torch.equal(a, b)
        # the public pair and the prepared plan
        n3, g3 = fsdt_loss_and_grad(m, *flds, bc, w_bc=0.1, phi_x_bc=-0.2, phi_y_bc=0.3, weights=wts)
        plan = ops.FsdtPlan(m.geom, *flds, bc, (0.1, -0.2, 0.3), weights=(1.0, 2.0, 0.5), **CONSTS)
        n4, g4 = plan.launch()
        n4b, g4b = plan.launch()
        assert torch.equal(n4, norms) and all(torch.equal(a, b) for a, b in zip(g4, g))
        assert bool(torch.isfinite(n3).all()) and all(bool(torch.isfinite(t).all()) for t in g3)
    finally:
        cfg("FSDT_FORM", "")
    # a consumer whose partials are gone is never silent: another reducing launch in between clears the ticket, a newer deferring launch replaces it
    vconsts = dict(CONSTS, q=0.0)
    R3, _, h1 = ops.fsdt_apply(m.geom, *flds, bc, (0.1, -0.2, 0.3), want_sums=False, defer_norms=True, **CONSTS)
    ops.fsdt_apply(m.geom, *flds, bc, (0.1, -0.2, 0.3), want_out=False, want_sums=True, **CONSTS)                 # reduces in the kernel, same workspace
    g5, _, n5 = ops.fsdt_apply(m.geom, *R3, bc, (0.0, 0.0, 0.0), want_sums=False, want_norms=True, in_num=wts, norms_from=h1, **vconsts)
    assert bool(torch.isnan(n5).all()) and bool(torch.isnan(g5[0][0, 0, 1, 1]))
    R4, _, h2 = ops.fsdt_apply(m.geom, *flds, bc, (0.1, -0.2, 0.3), want_sums=False, defer_norms=True, **CONSTS)
    R5, _, h3 = ops.fsdt_apply(m.geom, *flds, bc, (0.1, -0.2, 0.3), want_sums=False, defer_norms=True, **CONSTS)
    _, _, n6 = ops.fsdt_apply(m.geom, *R4, bc, (0.0, 0.0, 0.0), want_sums=False, want_norms=True, in_num=wts, norms_from=h2, **vconsts)   # stale ticket
    _, _, n7 = ops.fsdt_apply(m.geom, *R5, bc, (0.0, 0.0, 0.0), want_sums=False, want_norms=True, in_num=wts, norms_from=h3, **vconsts)   # the live one
    assert bool(torch.isnan(n6).all()) and torch.allclose(n7, norms, rtol=1e-6)          # (default form here; `norms` may come from the element form)
    with pytest.raises(ValueError):
        ops.fsdt_apply(m.geom, *flds, bc, in_num=wts, in_den=norms, norms_from=h)
    with pytest.raises(ValueError):
        ops.fsdt_apply(m.geom, *flds, bc, defer_norms=True, want_sums=True)


@pytest.mark.parametrize("mode,degree", [("reference", 1), ("total", 2), ("plan", 2)])
def test_plate_bending_example_converges(mode, degree, monkeypatch, capsys):
    """examples/plate_bending_fsdt.py -- the flow of the reference's e1_plate_bending_fsdt.py (fields as parameters, clamped plate under a uniform load):
    the reference's one-optimiser-per-field scheme on the three fused norms, the single-objective form behind autograd, and the prepared two-launch
    plan with deferred sums; the residual norms fall and the plate deflects towards the load."""
    import importlib.util
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("ex_plate", os.path.join(here, "..", "examples", "plate_bending_fsdt.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    monkeypatch.setattr(sys, "argv", ["plate_bending_fsdt.py", "--size", "17", "--degree", str(degree), "--epochs", "120", "--mode", mode])
    model = ex.main()
    out = capsys.readouterr().out.strip().splitlines()
    first, last = out[0], out[-2]
    num = lambda line: sum(float(t) for t in line.replace("+", " ").split() if "e" in t and t[0].isdigit())
    assert num(last) < 0.8 * num(first), (first, last)
    w = model.net_w[0].detach()
    assert bool(torch.isfinite(w).all()) and float(w.abs().max()) > 0.0
