"""GPU tests added in round 3: cached prepared calls behind the public API, strict prepared launches, the slab path with compact
Dirichlet forms (in-process emulation of the ranks), the registered operator's mask validation, the measurement probes."""
import ctypes

import numpy as np
import pytest
import torch

from test_gpu_parity import boundary_mask, close, cu, dev, module, seeded

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kw,B", [(dict(domain_size=64, ngp_1d=2), 2), (dict(domain_size=33, nsd=3), 1), (dict(domain_size=65, ngp_1d=3, fem_basis_deg=2), 2)])
def test_cached_calls_are_bitwise_equal_and_follow_their_buffers(kw, B):
    """energy_loss_and_grad / residual_loss keep the prepared launch of a call in a small LRU keyed on pointers, shapes, condition
    forms and coefficients (ops.poisson_apply): a repeat must give bitwise the results of the uncached preparation, write into FRESH
    output tensors (an earlier result is never overwritten), see in-place updates of its inputs, miss on new buffers, and leave inputs
    that need a conversion copy (non-contiguous, bool masks) to the uncached path."""
    from diffnet_amd import ops
    m = module(kw)
    shape = (B, 1, *m.geom.node_shape)
    u, nu, f = cu(seeded(shape, 1)), cu(seeded(shape, 2) + 0.5), cu(seeded(shape, 3))
    bc = boundary_mask((1,) + shape[1:]).to(torch.uint8).to(dev())
    ops.call_cache_clear()
    ops._CALL_STATS.update(hit=0, miss=0, uncached=0)
    scale = 1.0 / (B * m.geom.nelem_total)
    kwargs = dict(alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=scale, want_out=True, want_sums=True, loss_scale=scale)
    ref = ops.PoissonPlan(m.geom, u, nu, f, None, [(bc, 0.0)], **kwargs).launch()
    ref = [t.clone() for t in ref]
    l1, g1 = m.energy_loss_and_grad(u, nu, f, dirichlet=[(bc, 0.0)], c=1.0)
    l2, g2 = m.energy_loss_and_grad(u, nu, f, dirichlet=[(bc, 0.0)], c=1.0)
    assert ops._CALL_STATS["miss"] == 1 and ops._CALL_STATS["hit"] == 1
    assert g1.data_ptr() != g2.data_ptr() and l1.data_ptr() != l2.data_ptr()          # fresh outputs per call
    for g, l in ((g1, l1), (g2, l2)):
        assert torch.equal(g, ref[0]) and torch.equal(l, ref[2])
    # in-place update of an input: same key, new values
    u.mul_(0.5)
    l3, g3 = m.energy_loss_and_grad(u, nu, f, dirichlet=[(bc, 0.0)], c=1.0)
    ref3 = ops.PoissonPlan(m.geom, u, nu, f, None, [(bc, 0.0)], **kwargs).launch()
    assert ops._CALL_STATS["hit"] == 2 and torch.equal(g3, ref3[0]) and torch.equal(l3, ref3[2])
    assert torch.equal(g1, ref[0])                                                     # the earlier result is untouched
    # new buffers: a miss, then a hit of the new entry
    u2 = u.clone()
    l4, g4 = m.energy_loss_and_grad(u2, nu, f, dirichlet=[(bc, 0.0)], c=1.0)
    assert ops._CALL_STATS["miss"] == 2 and torch.equal(g4, g3)
    # another coefficient: another entry
    l5, _ = m.energy_loss_and_grad(u2, nu, f, dirichlet=[(bc, 0.0)], c=0.5)
    assert ops._CALL_STATS["miss"] == 3 and float(l5) != float(l4)
    # inputs that need a conversion are never cached: a non-contiguous field, a bool mask
    wide = torch.zeros(shape[:-1] + (2 * shape[-1],), device=dev())
    un = wide[..., ::2]
    un.copy_(u2)
    assert not un.is_contiguous()
    l6, g6 = m.energy_loss_and_grad(un, nu, f, dirichlet=[(bc.bool(), 0.0)], c=1.0)
    assert ops._CALL_STATS["uncached"] == 1 and torch.equal(g6, g4) and torch.equal(l6, l4)
    # the residual form shares the cache
    if m.geom.deg == 1:
        r1 = m.residual_loss(u2, nu, f, dirichlet=[(bc, 0.0)])
        r2 = m.residual_loss(u2, nu, f, dirichlet=[(bc, 0.0)])
        assert torch.equal(r1, r2)
    with pytest.raises(Exception, match="not contiguous"):
        ops.PoissonPlan(m.geom, un, nu, f, None, [(bc, 0.0)], **kwargs)
    with pytest.raises(Exception, match="bool or non-contiguous"):
        ops.PoissonPlan(m.geom, u2, nu, f, None, [(bc.bool(), 0.0)], **kwargs)


@pytest.mark.parametrize("nsd,sizes,world,B", [(2, (40, 37), 2, 2), (2, (64, 64), 3, 1), (3, (17, 19, 23), 2, 2), (3, (16, 16, 23), 4, 1)])
def test_slab_path_takes_packed_masks_and_box_faces(nsd, sizes, world, B):
    """SlabPoisson with the compact Dirichlet forms: a PackedMask is unpacked for the slab launches, BoxFaces name faces of
    the GLOBAL box -- the faces across the decomposed axis exist on the outermost ranks only.  The ranks are emulated in-process (the
    prepared launches of every rank run on this GPU; no collective): the two launches of a rank's evaluation (face strips, then the
    interior) leave the interface layers final after the first; the shares of the loss add up to the global loss, interior layers of
    the slab gradients equal the global gradient, interface layers are the sum of the two neighbours' parts.  Uneven slabs
    (22 element layers over 4 ranks: 6 / 6 / 5 / 5) included."""
    from diffnet_amd import BoxFaces, PackedMask
    from diffnet_amd.slab import SlabPoisson
    lengths = (1.0, 0.8, 1.3)[:nsd]
    kw = dict(nsd=nsd, domain_sizes=sizes + (1,) * (3 - nsd), domain_lengths=lengths + (1.0,) * (3 - nsd), domain_size=sizes[0])
    m = module(kw)
    shape = (B, 1, *m.geom.node_shape)
    u, nu, f = cu(seeded(shape, 41)), cu(seeded(shape, 42) + 0.5), cu(seeded(shape, 43))
    src = (seeded(shape, 44) < 0.04).to(torch.uint8).to(dev())
    box = BoxFaces("all")
    lref, gref = m.energy_loss_and_grad(u, nu, f, dirichlet=[(src, 1.0), (box, 0.0)], c=0.7)
    total, gsum = 0.0, torch.zeros_like(u)
    for r in range(world):
        sp = SlabPoisson(nsd, sizes, lengths, r, world, ngp_1d=2, device=dev(), overlap=True)
        dec = sp.dec
        ul, nul, fl = dec.take(u), dec.take(nu), dec.take(f)
        cond = [(PackedMask.pack(dec.take(src)), 1.0), (box, 0.0)]
        scale = 1.0 / (B * dec.nel_global)
        first, rest, local = sp._plans(ul, nul, fl, cond, 0.7, 1.0, scale)
        assert all(isinstance(d.mask, torch.Tensor) for d in local) and rest is not None
        grad, _, loss = first.launch()
        # after the first launch (the strips next to the faces) this rank's parts of its interface layers are final ...
        face_lo, face_hi = grad[:, :, 0].clone(), grad[:, :, -1].clone()
        g2, _, loss2 = rest.launch()
        assert g2.data_ptr() == grad.data_ptr() and loss2.data_ptr() == loss.data_ptr()      # ... the second launch fills in the rest
        assert torch.equal(grad[:, :, 0], face_lo) and torch.equal(grad[:, :, -1], face_hi)
        total += float(loss)
        gsum[:, :, dec.n0:dec.n1 + 1] += grad
        # overlap="auto" (the default, round 4): a slab this small is evaluated by ONE launch -- same numbers
        sp1 = SlabPoisson(nsd, sizes, lengths, r, world, ngp_1d=2, device=dev())
        one, none, _ = sp1._plans(ul, nul, fl, cond, 0.7, 1.0, scale)
        assert none is None and sp1.overlap == "auto"
        g1, _, l1 = one.launch()
        assert torch.equal(g1, grad)
        np.testing.assert_allclose(float(l1), float(loss), rtol=1e-6)
        with pytest.raises(Exception, match="contiguous"):
            sp._plans(ul[..., ::2], nul, fl, cond, 0.7, 1.0, scale)
    np.testing.assert_allclose(total, float(lref), rtol=2e-6)
    close(gsum, gref.cpu().numpy(), rtol=1e-5, arel=2e-6)


def test_registered_operator_validates_int32_masks():
    """int32 mask tensors are reserved for bit-packed masks (ops.PackedMask.bits): an int32 IMAGE is rejected instead of being
    reinterpreted as bit rows, and a bit tensor of the wrong shape is rejected by the operator."""
    from diffnet_amd import ops, torch_ops
    m = module(dict(domain_size=32, nsd=3))
    shape = (1, 1, 32, 32, 32)
    u = cu(seeded(shape, 1))
    img = boundary_mask(shape).to(torch.int32).to(dev())
    with pytest.raises(TypeError, match="int32"):
        m.energy_loss(u, dirichlet=[(img, 0.0)])
    bad_bits = torch.zeros((1, 32 * 32, 2), dtype=torch.int32, device=dev())        # row_words 2 for nx = 32: not a packed mask of this mesh
    with pytest.raises(Exception, match="bit-packed"):
        torch_ops.poisson_apply(u, None, None, None, bad_bits, None, 0.0, None, None, 0.0, *torch_ops.geometry_args(m.geom), 2.0, 1.0, 1.0, 1.0, 1.0, 1.0)
    # a packed mask in 3-D travels as its cached uint8 image (nothing is unpacked per call)
    pm = ops.PackedMask.pack(boundary_mask(shape).to(torch.uint8).to(dev()))
    l1 = m.energy_loss(u, dirichlet=[(pm, 0.0)])
    img8 = pm._u8
    l2 = m.energy_loss(u, dirichlet=[(pm, 0.0)])
    assert pm._u8 is img8 and torch.equal(l1, l2)
    l3 = m.energy_loss(u, dirichlet=[(boundary_mask(shape).to(dev()), 0.0)])
    assert torch.equal(l1, l3)


def test_measurement_probes_compute_what_they_say():
    """dn_probe_stream / dn_probe_march are measurement kernels, but their results are defined (out = a * b + c; the marching probe writes
    every owned row): a probe that skipped work would report a rate that means nothing."""
    from diffnet_amd import _lib
    L = _lib.lib()
    B, n = 2, 512
    a, b, c = (cu(seeded((B, n, n), s)) for s in (1, 2, 3))
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for mode in range(12):
        out = torch.full_like(a, float("nan"))
        assert L.dn_probe_stream(a.data_ptr(), b.data_ptr(), c.data_ptr(), out.data_ptr(), a.numel(), mode, stream) == 0
        assert torch.allclose(out, a * b + c, rtol=1e-6, atol=1e-7), mode          # (the kernel contracts a * b + c into one fma)
    for flags in (4, 5, 7, 12, 21, 29, 36, 44):
        for R, D in ((16, 1), (16, 2), (8, 3), (32, 4)):
            out = torch.full_like(a, float("nan"))
            assert L.dn_probe_march(a.data_ptr(), b.data_ptr(), c.data_ptr(), out.data_ptr(), B, n, R, D, flags, stream) == 0
            assert bool(torch.isfinite(out).all()), (flags, R, D)
    assert L.dn_probe_stream(a.data_ptr(), b.data_ptr(), c.data_ptr(), a.data_ptr(), 6, 0, stream) == -1
    assert L.dn_probe_march(a.data_ptr(), b.data_ptr(), c.data_ptr(), a.data_ptr(), B, n, 16, 9, 0, stream) == -1


@pytest.mark.parametrize("deg,ngp,sizes,B", [(2, 3, (65, 65), 1), (2, 3, (129, 41), 2), (1, 2, (70, 37), 3), (3, 3, (64, 46), 1), (2, 4, (257, 33), 1)])
def test_fsdt_chained_strips_equal_single_strips(deg, ngp, sizes, B):
    """dn_fsdt_apply with chained sub-strips (PLAN_FSDT "64,R,W": W one-wave sub-strips per workgroup, rows shared through LDS, one
    recomputed seam layer per workgroup instead of one per strip) against the un-chained launch ("T,R") for every chain length class:
    W = 2 .. 12, strips of 1 .. 5 element rows, a last workgroup with fewer sub-strips than the chain, more than one chunk of 63
    element columns, Q1 / Q2 / Q3, no mask / uint8 / fp32 masks, the sums and the norms written by the launch; and the launch plan the
    library picks by itself against both."""
    from diffnet_amd import _lib, ops
    kw = dict(nsd=2, domain_sizes=sizes, domain_lengths=(1.0, 0.9), domain_size=sizes[0], fem_basis_deg=deg, ngp_1d=ngp)
    m = module(kw)
    shape = (B, 1, sizes[1], sizes[0])
    flds = [cu(seeded(shape, 90 + i)) for i in range(3)]
    bcf = boundary_mask(shape).to(dev())
    bcf[0, 0, sizes[1] // 2, 3:9] = 1.0
    consts = dict(D11=1.3, D12=0.4, D22=1.1, D66=0.6, A44=0.8, A55=0.9, q=1.2, wscale=0.3)
    try:
        for mask in (None, bcf, bcf.to(torch.uint8)):
            _lib.config_set("PLAN_FSDT", "192,3")
            ref, rsums, rnorms = ops.fsdt_apply(m.geom, *flds, mask, (0.1, -0.2, 0.3), want_norms=True, **consts)
            cases = ["64,1,2", "64,2,3", "64,1,12", "64,3,5", "64,5,4", "64,2,9", "64,4,7", ""]
            for plan in cases:
                _lib.config_set("PLAN_FSDT", plan)
                got, sums, norms = ops.fsdt_apply(m.geom, *flds, mask, (0.1, -0.2, 0.3), want_norms=True, **consts)
                for k in range(3):
                    scale = float(ref[k].abs().max())
                    assert float((got[k] - ref[k]).abs().max()) <= 3e-6 * scale, f"plan {plan!r} field {k} mask {None if mask is None else mask.dtype}"
                np.testing.assert_allclose(sums.cpu().numpy(), rsums.cpu().numpy(), rtol=1e-6)
                np.testing.assert_allclose(norms.cpu().numpy(), np.sqrt(rsums.cpu().numpy()), rtol=1e-6)
    finally:
        _lib.config_set("PLAN_FSDT", "")


def test_fsdt_q2_middle_point_form_equals_the_generic_form():
    """Q2 with the symmetric 3-point rule runs an element routine in which the contractions at the middle Gauss point (basis (0, 1, 0),
    derivative (-d, 0, d)) are spelled as copies and differences; dn_config_set("FSDT_GENERIC") runs the table-driven routine.  Same
    numbers to rounding (d (F2 - F0) against d F2 + (-d) F0 + 0 F1: one rounding less)."""
    from diffnet_amd import _lib, ops
    m = module(dict(domain_size=129, fem_basis_deg=2, ngp_1d=3))
    shape = (2, 1, 129, 129)
    flds = [cu(seeded(shape, 60 + i)) for i in range(3)]
    bc = boundary_mask(shape).to(dev())
    consts = dict(D11=1.3, D12=0.4, D22=1.1, D66=0.6, A44=0.8, A55=0.9, q=1.2, wscale=0.3)
    try:
        for plan in ("192,4", "64,3,4"):
            _lib.config_set("PLAN_FSDT", plan)
            _lib.config_set("FSDT_FORM", "elem")          # (round 4: the default is the assembled-stencil form, tests/test_gpu_fsdt_stencil.py)
            a, sa = ops.fsdt_apply(m.geom, *flds, bc, (0.0, 0.1, 0.0), **consts)
            _lib.config_set("FSDT_FORM", "")
            _lib.config_set("FSDT_GENERIC", "1")
            b, sb = ops.fsdt_apply(m.geom, *flds, bc, (0.0, 0.1, 0.0), **consts)
            _lib.config_set("FSDT_GENERIC", "")
            for x, y in zip(a, b):
                assert float((x - y).abs().max()) <= 2e-6 * float(y.abs().max()), plan
            np.testing.assert_allclose(sa.cpu().numpy(), sb.cpu().numpy(), rtol=1e-6)
    finally:
        _lib.config_set("PLAN_FSDT", "")
        _lib.config_set("FSDT_GENERIC", "")
        _lib.config_set("FSDT_FORM", "")


@pytest.mark.parametrize("kw,B,cfg", [(dict(domain_size=512, ngp_1d=3), 3, None), (dict(domain_size=96, ngp_1d=2), 2, ("Q1_RULE_KERNEL", "1")),
                                      (dict(domain_size=129, ngp_1d=3, fem_basis_deg=2), 1, None), (dict(domain_size=65, nsd=3), 2, None),
                                      (dict(domain_size=65, nsd=3), 1, ("Q1_3D_T16", "1")), (dict(domain_size=33, nsd=3, ngp_1d=3), 1, None),
                                      (dict(domain_size=9, nsd=3), 1, None)])
def test_split_evaluation_equals_one_launch(kw, B, cfg):
    """dn_poisson_args.strip_select: the launch over the first + last strip of the marched axis followed by the launch over the other
    strips (sums accumulated) is the one-launch evaluation -- the gradient bitwise (the two launches write disjoint strips, every strip
    is computed exactly as in the full launch), the loss to the rounding of one more addition.  Every kernel family that marches strips:
    closed-form and per-point 2-D Q1, generic Q2, the two 3-D forms, a 3-point rule, and a mesh with fewer than three strips."""
    from diffnet_amd import _lib, ops
    m = module(kw)
    shape = (B, 1, *m.geom.node_shape)
    u, nu, f = cu(seeded(shape, 51)), cu(seeded(shape, 52) + 0.5), cu(seeded(shape, 53))
    bc = boundary_mask((1,) + shape[1:]).to(torch.uint8).to(dev())
    scale = 1.0 / (B * m.geom.nelem_total)
    kwargs = dict(alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=scale, want_out=True, want_sums=True, loss_scale=scale)
    if cfg:
        _lib.config_set(*cfg)
    try:
        ref = [t.clone() for t in ops.PoissonPlan(m.geom, u, nu, f, None, [(bc, 0.0)], **kwargs).launch()]
        first = ops.PoissonPlan(m.geom, u, nu, f, None, [(bc, 0.0)], strip_select=1, **kwargs)
        rest = ops.PoissonPlan(m.geom, u, nu, f, None, [(bc, 0.0)], strip_select=2, continues=first, **kwargs)
        first.result[0].fill_(float("nan"))
        first.launch()
        part = first.result[0].clone()
        out, sums, loss = rest.launch()
        # (the one-launch evaluation of a SMALL 2-D Q1 launch runs chained strips, whose seam rows are summed in another order: rounding)
        chained = m.geom.nsd == 2 and m.geom.deg == 1 and cfg is None
        same = (lambda a, b: float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())) if chained else torch.equal
        assert same(out, ref[0])
        done = ~torch.isnan(part)                   # what the first launch wrote is final (meshes of one or two strips: everything)
        assert same(part[done], ref[0][done]) and bool(done[:, :, 0].all()) and bool(done[:, :, -1].all())
        np.testing.assert_allclose(sums.cpu().numpy(), ref[1].cpu().numpy(), rtol=1e-6 if chained else 1e-12)
        np.testing.assert_allclose(float(loss), float(ref[2]), rtol=2e-6 if chained else 2e-7)
        with pytest.raises(Exception):
            bad = ops.PoissonPlan(m.geom, u, nu, f, None, [(bc, 0.0)], strip_select=3, **kwargs)
            bad.launch()
    finally:
        if cfg:
            _lib.config_set(cfg[0], "")


@pytest.mark.parametrize("kw,B", [(dict(domain_size=512, ngp_1d=3), 4), (dict(domain_size=65, nsd=3), 2), (dict(domain_size=64, ngp_1d=2), 1)])
def test_deferred_sums_on_a_side_stream_equal_the_in_kernel_reduction(kw, B):
    """PoissonPlan(async_sums=True): the launch leaves per-workgroup partial sums, dn_poisson_finish_sums forms the scalars on a side
    stream.  Same gradient bitwise, same sums to the rounding of another (fixed) order of additions, repeatable bitwise; the results
    follow in-place updates of the inputs; many launches back to back (the next launch may not overwrite partials the side stream has
    not consumed yet) give the same numbers as one."""
    from diffnet_amd import ops
    m = module(kw)
    shape = (B, 1, *m.geom.node_shape)
    u, nu, f = cu(seeded(shape, 61)), cu(seeded(shape, 62) + 0.5), cu(seeded(shape, 63))
    bc = boundary_mask((1,) + shape[1:]).to(torch.uint8).to(dev())
    scale = 1.0 / (B * m.geom.nelem_total)
    kwargs = dict(alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=scale, want_out=True, want_sums=True, loss_scale=scale)
    ref = [t.clone() for t in ops.PoissonPlan(m.geom, u, nu, f, None, [(bc, 0.0)], **kwargs).launch()]
    plan = ops.PoissonPlan(m.geom, u, nu, f, None, [(bc, 0.0)], async_sums=True, **kwargs)
    out, sums, loss = plan.launch()
    plan.wait_sums()
    assert torch.equal(out, ref[0])
    np.testing.assert_allclose(sums.cpu().numpy(), ref[1].cpu().numpy(), rtol=1e-13)
    np.testing.assert_allclose(float(loss), float(ref[2]), rtol=2e-7)
    first = (sums.clone(), loss.clone())
    for _ in range(50):
        plan.launch()
    plan.wait_sums()
    assert torch.equal(sums, first[0]) and torch.equal(loss, first[1]) and torch.equal(out, ref[0])
    u.mul_(0.5)
    ref2 = [t.clone() for t in ops.PoissonPlan(m.geom, u, nu, f, None, [(bc, 0.0)], **kwargs).launch()]
    plan.launch()
    torch.cuda.synchronize()
    assert torch.equal(out, ref2[0])
    np.testing.assert_allclose(sums.cpu().numpy(), ref2[1].cpu().numpy(), rtol=1e-13)


@pytest.mark.parametrize("sizes,B", [((16, 16, 40), 3), ((18, 23, 29), 2), ((34, 17, 9), 1), ((64, 64, 64), 1), ((2, 2, 2), 2), ((62, 31, 33), 1), ((128, 48, 20), 1)])
def test_3d_two_elements_per_thread_equals_one(sizes, B):
    """The 3-D node-owner kernel with two elements per thread (poisson3d_q1n2_kernel: even nx, exact 2-point rule) against the one-element
    form (dn_config_set("Q1_3D_E1")) and, on the small meshes, against the oracle: ragged tiles (nx = 18, 34, 62: several chunks with partly
    empty last threads), more than one tile in y, several strips, every coefficient / condition combination the kernel is instantiated for
    (uint8 and fp32 masks, one and two conditions, with and without nu / f), both ways of summing the stiffness energy, the residual form."""
    from diffnet_amd import _lib
    kw = dict(nsd=3, domain_sizes=sizes, domain_lengths=(1.0, 1.3, 0.7), domain_size=sizes[0], ngp_1d=2)
    m = module(kw)
    shape = (B, 1, sizes[2], sizes[1], sizes[0])
    u, nu, f = cu(seeded(shape, 71)), cu(seeded(shape, 72) + 0.5), cu(seeded(shape, 73))
    bc = boundary_mask((1,) + shape[1:]).to(torch.uint8).to(dev())
    src = (seeded(shape, 74) < 0.05).to(torch.uint8).to(dev())
    cases = {"none": (nu, f, []), "u8": (nu, f, [(bc, 0.0)]), "u8 x2": (nu, f, [(src, 1.0), (bc, 0.0)]), "f32": (nu, f, [(bc.float(), 0.0)]),
             "f32 x2": (nu, f, [(src.float(), 1.0), (bc.float(), 0.0)]), "no nu": (None, f, [(bc, 0.0)]), "no f": (nu, None, [(bc, 0.0)]),
             "bare": (None, None, [(bc, 0.0)])}
    for name, (a, b, d) in cases.items():
        for esum in ("", "1"):
            _lib.config_set("Q1_3D_E1SUM", esum)
            try:
                l2, g2 = m.energy_loss_and_grad(u, a, b, dirichlet=d, c=0.7)
                r2 = m.residual_loss(u, a, b, dirichlet=d)
                _lib.config_set("Q1_3D_E1", "1")
                l1, g1 = m.energy_loss_and_grad(u, a, b, dirichlet=d, c=0.7)
                r1 = m.residual_loss(u, a, b, dirichlet=d)
            finally:
                _lib.config_set("Q1_3D_E1", "")
                _lib.config_set("Q1_3D_E1SUM", "")
            scale = float(g1.abs().max()) + 1e-30
            assert float((g2 - g1).abs().max()) <= 3e-6 * scale, f"{name} esum={esum!r}"
            np.testing.assert_allclose(float(l2), float(l1), rtol=5e-6, err_msg=f"{name} esum={esum!r}")
            np.testing.assert_allclose(float(r2), float(r1), rtol=5e-6, err_msg=f"{name} residual")
    if sizes[0] * sizes[1] * sizes[2] <= 20000:
        from oracle.fem_oracle import Oracle
        o = Oracle(**kw)
        ur = u.cpu().clone().requires_grad_(True)
        ref = o.energy(ur, nu.cpu(), f.cpu(), dirichlet=[(src.cpu().float(), 1.0), (bc.cpu().float(), 0.0)], c=0.7)
        (gref,) = torch.autograd.grad(ref, ur)
        l2, g2 = m.energy_loss_and_grad(u, nu, f, dirichlet=[(src, 1.0), (bc, 0.0)], c=0.7)
        np.testing.assert_allclose(float(l2), float(ref), rtol=1e-5)
        close(g2, gref.numpy(), rtol=1e-4, arel=1e-4)


@pytest.mark.parametrize("deg,ngp,n,B", [(1, 2, 48, 2), (2, 3, 65, 1), (2, 3, 129, 2)])
def test_fsdt_loss_and_grad_and_plan_equal_the_autograd_path(deg, ngp, n, B):
    """fsdt_loss_and_grad (two launches, no graph), fsdt_total_loss (one autograd node) and ops.FsdtPlan (prepared launches on fixed
    buffers) against sum(fsdt_loss(...)).backward(): the same two kernels on the same data, so the same bits; weighted norms; the plan
    follows in-place updates of its input buffers and refuses non-contiguous fields and a foreign stream."""
    from diffnet_amd import ops
    from diffnet_amd.elasticity import _constants, fsdt_loss, fsdt_loss_and_grad, fsdt_total_loss
    m = module(dict(domain_size=n, fem_basis_deg=deg, ngp_1d=ngp))
    shape = (B, 1, n, n)
    flds = [cu(seeded(shape, 30 + i)).requires_grad_(True) for i in range(3)]
    bc = boundary_mask(shape).to(dev())
    norms_ref = fsdt_loss(m, *flds, bc, w_bc=0.0, phi_x_bc=0.1, phi_y_bc=-0.1)
    grads_ref = torch.autograd.grad(sum(norms_ref), flds)
    norms, grads = fsdt_loss_and_grad(m, *flds, bc, w_bc=0.0, phi_x_bc=0.1, phi_y_bc=-0.1)
    assert torch.equal(norms, torch.stack(norms_ref))
    for g, r in zip(grads, grads_ref):
        assert torch.equal(g, r)
    total = fsdt_total_loss(m, *flds, bc, w_bc=0.0, phi_x_bc=0.1, phi_y_bc=-0.1)
    np.testing.assert_allclose(float(total), float(sum(norms_ref)), rtol=1e-6)
    for g, r in zip(torch.autograd.grad(total, flds), grads_ref):
        assert torch.equal(g, r)
    wts = torch.tensor([0.5, 2.0, -1.5], device=dev())
    _, gw = fsdt_loss_and_grad(m, *flds, bc, w_bc=0.0, phi_x_bc=0.1, phi_y_bc=-0.1, weights=wts)
    gw_ref = torch.autograd.grad(sum(w * x for w, x in zip(wts, fsdt_loss(m, *flds, bc, w_bc=0.0, phi_x_bc=0.1, phi_y_bc=-0.1))), flds)
    for g, r in zip(gw, gw_ref):
        close(g, r.cpu().numpy(), rtol=1e-5, arel=1e-6)
    # prepared launches: fixed buffers, two ctypes calls
    bufs = [f.detach().clone() for f in flds]
    consts = _constants(1.0, 0.25, 0.1, 1.0)
    plan = ops.FsdtPlan(m.geom, *bufs, bc, (0.0, 0.1, -0.1), q=1.0, wscale=(0.5 * m.h) * (0.5 * m.h), **consts)
    pn, pg = plan.launch()
    assert torch.equal(pn, norms)
    for g, r in zip(pg, grads_ref):
        assert torch.equal(g, r)
    with torch.no_grad():
        bufs[0].mul_(1.5)
        bufs[2].add_(0.25)
    pn, pg = plan.launch()
    n2, g2 = fsdt_loss_and_grad(m, *bufs, bc, w_bc=0.0, phi_x_bc=0.1, phi_y_bc=-0.1)
    assert torch.equal(pn, n2)
    for g, r in zip(pg, g2):
        assert torch.equal(g, r)
    only = ops.FsdtPlan(m.geom, *bufs, bc, (0.0, 0.1, -0.1), want_grad=False, q=1.0, wscale=(0.5 * m.h) * (0.5 * m.h), **consts)
    on, og = only.launch()
    assert og is None and torch.equal(on, n2)
    with pytest.raises(ops.DiffNetHipError):
        ops.FsdtPlan(m.geom, bufs[0].transpose(2, 3), bufs[1], bufs[2], bc)
    with pytest.raises(ops.DiffNetHipError):
        with torch.cuda.stream(torch.cuda.Stream()):
            plan.launch()
    torch.cuda.synchronize()


def test_3d_two_elements_per_thread_edge_shapes():
    """poisson3d_q1n2_kernel at the edges of its tiling: one or two elements along an axis, nx / ny around the 30-element chunk and 15-row tile
    widths (tiles overlap by one thread), strips of 1 .. 5 layers through "PLAN3D" "16,16,2,R" (seam layers, a single layer, a last strip with one
    layer), several samples with different masks, one condition present in either slot -- against the one-element kernel and the oracle."""
    from diffnet_amd import _lib
    from oracle.fem_oracle import Oracle
    shapes = [(2, 2, 2), (4, 2, 3), (2, 17, 2), (30, 15, 4), (32, 16, 5), (34, 17, 3), (60, 31, 2), (62, 32, 6), (64, 2, 9), (6, 46, 7), (92, 3, 4)]
    for si, sizes in enumerate(shapes):
        kw = dict(nsd=3, domain_sizes=sizes, domain_lengths=(1.0, 0.7, 1.4), domain_size=sizes[0], ngp_1d=2)
        m = module(kw)
        B = 1 + si % 3
        shape = (B, 1, sizes[2], sizes[1], sizes[0])
        u, nu, f = cu(seeded(shape, 500 + si)), cu(seeded(shape, 600 + si) + 0.5), cu(seeded(shape, 700 + si))
        bc = (seeded(shape, 800 + si) < 0.15).to(torch.uint8).to(dev())
        bc[..., 0] = 1
        src = (seeded(shape, 900 + si) < 0.05).to(torch.uint8).to(dev())
        conds = [[(bc, 0.3)], [(src, 1.0), (bc, 0.0)], [(bc.float(), -0.2)]]
        nelz = sizes[2] - 1
        try:
            for plan in [""] + [f"16,16,2,{R}" for R in (1, 2, 3, 5) if R <= max(nelz, 1)]:
                for d in conds:
                    _lib.config_set("PLAN3D", plan)
                    l2, g2 = m.energy_loss_and_grad(u, nu, f, dirichlet=d, c=0.5)
                    _lib.config_set("PLAN3D", "")
                    _lib.config_set("Q1_3D_E1", "1")
                    l1, g1 = m.energy_loss_and_grad(u, nu, f, dirichlet=d, c=0.5)
                    _lib.config_set("Q1_3D_E1", "")
                    scale = float(g1.abs().max()) + 1e-30
                    assert float((g2 - g1).abs().max()) <= 4e-6 * scale, f"{sizes} plan {plan!r}"
                    np.testing.assert_allclose(float(l2), float(l1), rtol=1e-5, atol=1e-7, err_msg=f"{sizes} plan {plan!r}")
        finally:
            _lib.config_set("PLAN3D", "")
            _lib.config_set("Q1_3D_E1", "")
        if sizes[0] * sizes[1] * sizes[2] <= 6000:
            o = Oracle(**kw)
            ur = u.cpu().clone().requires_grad_(True)
            ref = o.energy(ur, nu.cpu(), f.cpu(), dirichlet=[(src.cpu().float(), 1.0), (bc.cpu().float(), 0.0)], c=0.5)
            (gref,) = torch.autograd.grad(ref, ur)
            l2, g2 = m.energy_loss_and_grad(u, nu, f, dirichlet=[(src, 1.0), (bc, 0.0)], c=0.5)
            np.testing.assert_allclose(float(l2), float(ref), rtol=2e-5, atol=1e-7)
            close(g2, gref.numpy(), rtol=1e-4, arel=1e-4)


def test_plan_writes_its_loss_into_a_callers_slot():
    """PoissonPlan(loss_out=...): the float32 loss of every launch lands in the caller's one-element tensor -- e.g. a slot of a buffer that ONE
    collective reduces for several steps (bench.py, N > 1) -- and nowhere else; wrong tensors are refused."""
    from diffnet_amd import ops
    m = module(dict(domain_size=96, ngp_1d=3))
    shape = (3, 1, 96, 96)
    u, nu, f = cu(seeded(shape, 11)), cu(seeded(shape, 12) + 0.5), cu(seeded(shape, 13))
    bc = boundary_mask(shape).to(torch.uint8).to(dev())
    kw = dict(alpha=2.0, beta=1.0, c=1.0, wscale=1.0, out_scale=0.5, want_out=True, want_sums=True, loss_scale=0.25)
    ref = ops.PoissonPlan(m.geom, u, nu, f, None, [(bc, 0.0)], **kw).launch()
    buf = torch.full((4,), -7.0, device=dev())
    plan = ops.PoissonPlan(m.geom, u, nu, f, None, [(bc, 0.0)], loss_out=buf[2:3], **kw)
    out, sums, loss = plan.launch()
    assert loss.data_ptr() == buf[2:3].data_ptr()
    assert torch.equal(buf[2], ref[2]) and torch.equal(out, ref[0]) and torch.equal(sums, ref[1])
    assert buf[0] == -7.0 and buf[1] == -7.0 and buf[3] == -7.0
    with pytest.raises(ops.DiffNetHipError):
        ops.PoissonPlan(m.geom, u, nu, f, None, [(bc, 0.0)], loss_out=torch.zeros(1), **kw)
    with pytest.raises(ops.DiffNetHipError):
        ops.PoissonPlan(m.geom, u, nu, f, None, [(bc, 0.0)], loss_out=buf[0:2], **kw)
    with pytest.raises(ValueError):
        ops.PoissonPlan(m.geom, u, nu, f, None, [(bc, 0.0)], loss_out=buf[0:1], **{**kw, "loss_scale": None})


@pytest.mark.parametrize("kw,B", [(dict(nsd=3, domain_sizes=(9, 7, 5), domain_lengths=(1.0, 0.8, 0.6), domain_size=9, fem_basis_deg=2), 2),
                                  (dict(nsd=3, domain_sizes=(13, 9, 11), domain_lengths=(1.0, 0.7, 0.9), domain_size=13, fem_basis_deg=2, ngp_1d=4), 1),
                                  (dict(nsd=3, domain_sizes=(7, 10, 4), domain_lengths=(1.0, 1.2, 0.5), domain_size=7, fem_basis_deg=3), 2),
                                  (dict(nsd=3, domain_sizes=(10, 7, 13), domain_lengths=(1.0, 0.6, 1.1), domain_size=10, fem_basis_deg=3, ngp_1d=4), 1),
                                  (dict(nsd=3, domain_size=17, fem_basis_deg=2), 3)])
def test_fused_3d_q2_q3_vs_oracle(kw, B):
    """dn_poisson_apply on 3-D meshes of Q2 / Q3 elements (poisson3d_gen.hip: element vectors + fixed-order gather assembly) against the
    oracle: energy loss and its gradient (nodal forcing, forcing at the Gauss points, no forcing / no nu), weak-form residual and the
    residual loss with its gradient, Dirichlet conditions with constant values and value fields, uint8 and fp32 masks, per sample; runs
    of the same call are bitwise equal (no atomics)."""
    from oracle.fem_oracle import Oracle
    m, o = module(kw), Oracle(**kw)
    shape = (B, 1, *m.geom.node_shape)
    u, nu, f = seeded(shape, 5), seeded(shape, 6, 0.5), seeded(shape, 7)
    bc = boundary_mask(shape)
    bc[0, 0, shape[2] // 2, 1:3, 2:4] = 1.0
    ubc = seeded(shape[2:], 8)
    fgp = seeded((B, m.geom.ngp_total, *m.geom.elem_shape), 9)
    ud, nud, fd = u.to(dev()), nu.to(dev()), f.to(dev())
    for name, (nn, ff, fg, dl) in {"nodal f, value field": (nu, f, None, [(bc, ubc[None, None])]), "f at the Gauss points": (nu, None, fgp, [(bc, 0.3)]),
                                   "bare": (None, None, None, [(bc, 0.0)]), "two conditions": (nu, f, None, [((seeded(shape, 10) < 0.1).float(), 1.0), (bc, 0.0)])}.items():
        ur = u.clone().requires_grad_(True)
        ref = o.energy(ur, nn, ff, f_gp=fg, dirichlet=dl, c=0.5, jac=0.7)
        (gref,) = torch.autograd.grad(ref, ur)
        cuo = lambda t: None if t is None else t.to(dev())
        for u8 in (False, True):
            dd = [((mk.to(torch.uint8) if u8 else mk).to(dev()), v.to(dev()) if isinstance(v, torch.Tensor) else v) for mk, v in dl]
            loss, g = m.energy_loss_and_grad(ud, cuo(nn), cuo(ff), f_gp=cuo(fg), dirichlet=dd, c=0.5, jac=0.7)
            np.testing.assert_allclose(float(loss), float(ref), rtol=2e-5, atol=1e-7, err_msg=name)
            close(g, gref.numpy(), rtol=1e-4, arel=1e-4)
            loss2, g2 = m.energy_loss_and_grad(ud, cuo(nn), cuo(ff), f_gp=cuo(fg), dirichlet=dd, c=0.5, jac=0.7)
            assert torch.equal(g, g2) and torch.equal(loss, loss2), name
    ur = u.clone().requires_grad_(True)
    Rref = o.residual_any_degree(ur, nu, f, dirichlet=[(bc, ubc[None, None])], jac=0.25, zero_masks=[bc])
    (gref,) = torch.autograd.grad(torch.sum(Rref ** 2), ur)
    d = [(bc.to(dev()), ubc.to(dev()))]
    R = m.residual(ud, nud, fd, dirichlet=d, jac=0.25)
    close(R, Rref.detach().numpy(), rtol=1e-4, arel=2e-5)
    ug = ud.clone().requires_grad_(True)
    v = m.residual_loss(ug, nud, fd, dirichlet=d, jac=0.25)
    (g,) = torch.autograd.grad(v, ug)
    np.testing.assert_allclose(float(v), float(torch.sum(Rref ** 2)), rtol=2e-5)
    close(g, gref.numpy(), rtol=1e-4, arel=1e-4)
