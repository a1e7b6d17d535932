"""GPU parity tests of the round-4 default 3-D Q1 kernel (diffnet_amd/csrc/poisson3d_q1_cf.hip: monomial in-plane stages, closed-form z
integration, forcing through the z mass stencil) -- the loss bodies of IBN/poisson-3d/parametric/IBN_3D.py:114-136 and
IBN/poisson-3d/non-parametric/solve_in_object_3d.py:75-102 on the fused operator.  Checked against the CPU oracle (DiffNetFEM.py:7-18 restated,
pinned to the reference's fixtures by tests/test_oracle_golden.py) and against the round-3 per-Gauss-point kernel (dn_config_set("Q1_3D_N2")).
Tolerances: loss rtol 1e-5, gradient 1e-4 of its largest entry against the oracle; 5e-6 between the two kernels."""
import numpy as np
import pytest
import torch

from test_gpu_parity import boundary_mask, close, cu, dev, module, seeded

pytestmark = pytest.mark.gpu


def _oracle(kw):
    from oracle.fem_oracle import Oracle
    return Oracle(**kw)


@pytest.mark.parametrize("sizes,lengths,B,plan", [
    ((6, 5, 4), (1.0, 1.0, 1.0), 2, ""), ((18, 23, 29), (1.0, 1.3, 0.7), 2, ""), ((34, 17, 9), (2.0, 1.0, 0.5), 1, ""),
    ((2, 2, 2), (1.0, 1.0, 1.0), 2, ""), ((62, 31, 33), (1.0, 1.0, 1.0), 1, "16,16,2,5"), ((16, 16, 40), (1.0, 0.5, 2.0), 1, "16,16,2,4"),
    ((32, 32, 3), (1.0, 1.0, 1.0), 1, ""), ((48, 40, 24), (1.0, 1.0, 1.0), 1, "16,16,2,7")])
def test_closed_form_kernel_vs_oracle(sizes, lengths, B, plan):
    """energy + gradient and the residual loss against the oracle: ragged tiles, several chunks / tiles / strips (plan overrides with strip
    heights that do not divide the mesh), anisotropic elements, two conditions with non-zero values, batch."""
    from diffnet_amd import _lib
    kw = dict(nsd=3, domain_sizes=sizes, domain_lengths=lengths, domain_size=sizes[0], ngp_1d=2)
    m = module(kw)
    o = _oracle(kw)
    shape = (B, 1, sizes[2], sizes[1], sizes[0])
    u, nu, f = seeded(shape, 11), seeded(shape, 12) + 0.5, seeded(shape, 13) - 0.3
    bc = boundary_mask((1,) + shape[1:])
    src = (seeded(shape, 14) < 0.05).float()
    conds = [(src, 0.8), (bc, 0.1)]
    ur = u.clone().requires_grad_(True)
    ref = o.energy(ur, nu, f, dirichlet=conds, c=0.7, jac=0.3)
    (gref,) = torch.autograd.grad(ref, ur)
    _lib.config_set("PLAN3D", plan)
    try:
        for fmt in (torch.uint8, torch.float32):
            d = [(cu(mk).to(fmt), v) for mk, v in conds]
            l, g = m.energy_loss_and_grad(cu(u), cu(nu), cu(f), dirichlet=d, c=0.7, jac=0.3)
            np.testing.assert_allclose(float(l), float(ref), rtol=1e-5)
            close(g, gref.numpy(), rtol=1e-4, arel=1e-4)
    finally:
        _lib.config_set("PLAN3D", "")


@pytest.mark.parametrize("sizes,B", [((16, 16, 40), 3), ((18, 23, 29), 2), ((34, 17, 9), 1), ((64, 64, 64), 1), ((2, 2, 2), 2), ((62, 31, 33), 1),
                                      ((128, 48, 20), 1)])
def test_closed_form_kernel_equals_the_per_point_kernel(sizes, B):
    """Every instantiation (with / without nu and f, uint8 / fp32 masks, one / two conditions, box faces, load vectors), energy and residual
    forms, against the round-3 kernel on the same inputs."""
    from diffnet_amd import BoxFaces, LoadVector, _lib
    kw = dict(nsd=3, domain_sizes=sizes, domain_lengths=(1.0, 1.3, 0.7), domain_size=sizes[0], ngp_1d=2)
    m = module(kw)
    shape = (B, 1, sizes[2], sizes[1], sizes[0])
    u, nu, f = cu(seeded(shape, 71)), cu(seeded(shape, 72) + 0.5), cu(seeded(shape, 73))
    bc = boundary_mask((1,) + shape[1:]).to(torch.uint8).to(dev())
    src = (seeded(shape, 74) < 0.05).to(torch.uint8).to(dev())
    lv = LoadVector.assemble(m.geom, f)
    cases = {"none": (nu, f, []), "u8": (nu, f, [(bc, 0.0)]), "u8 x2": (nu, f, [(src, 1.0), (bc, 0.0)]), "f32": (nu, f, [(bc.float(), 0.0)]),
             "f32 x2": (nu, f, [(src.float(), 1.0), (bc.float(), 0.0)]), "no nu": (None, f, [(bc, 0.0)]), "no f": (nu, None, [(bc, 0.0)]),
             "bare": (None, None, [(bc, 0.0)]), "box": (nu, f, [(BoxFaces(), 0.25)]), "box + u8": (nu, f, [(src, 1.0), (BoxFaces(), 0.0)]),
             "load": (nu, lv, [(bc, 0.0)]), "load no nu": (None, lv, [(src, 1.0), (bc, 0.0)])}
    for name, (a, b, d) in cases.items():
        l2, g2 = m.energy_loss_and_grad(u, a, b, dirichlet=d, c=0.7)
        r2 = m.residual_loss(u, a, b, dirichlet=d)
        _lib.config_set("Q1_3D_N2", "1")
        try:
            l1, g1 = m.energy_loss_and_grad(u, a, b, dirichlet=d, c=0.7)
            r1 = m.residual_loss(u, a, b, dirichlet=d)
        finally:
            _lib.config_set("Q1_3D_N2", "")
        scale = float(g1.abs().max()) + 1e-30
        assert float((g2 - g1).abs().max()) <= 5e-6 * scale, name
        np.testing.assert_allclose(float(l2), float(l1), rtol=5e-6, err_msg=name)
        np.testing.assert_allclose(float(r2), float(r1), rtol=1e-5, err_msg=f"{name} residual")


def test_closed_form_kernel_autograd_paths():
    """energy_loss.backward() with a non-unit upstream gradient and residual_loss.backward() (the symmetric operator applied to 2 R: the
    launch without forcing and without sums) against the oracle's autograd."""
    kw = dict(nsd=3, domain_sizes=(20, 18, 14), domain_lengths=(1.0, 0.9, 0.7), domain_size=20, ngp_1d=2)
    m = module(kw)
    o = _oracle(kw)
    shape = (2, 1, 14, 18, 20)
    u, nu, f = seeded(shape, 21), seeded(shape, 22) + 0.5, seeded(shape, 23)
    bc = boundary_mask((1,) + shape[1:])
    ud = cu(u).requires_grad_(True)
    (3.0 * m.energy_loss(ud, cu(nu), cu(f), dirichlet=[(cu(bc), 0.0)], c=0.5)).backward()
    ur = u.clone().requires_grad_(True)
    (3.0 * o.energy(ur, nu, f, dirichlet=[(bc, 0.0)], c=0.5)).backward()
    close(ud.grad, ur.grad.numpy(), rtol=1e-4, arel=1e-4)
    ud2 = cu(u).requires_grad_(True)
    r = m.residual_loss(ud2, cu(nu), cu(f), dirichlet=[(cu(bc), 0.0)])
    r.backward()
    ur2 = u.clone().requires_grad_(True)
    rr = o.resmin(ur2, nu, f, dirichlet=[(bc, 0.0)], zero_masks=[bc])
    rr.backward()
    np.testing.assert_allclose(float(r), float(rr), rtol=2e-5)
    close(ud2.grad, ur2.grad.numpy(), rtol=2e-4, arel=2e-4)


def test_closed_form_kernel_at_the_baseline_meshes():
    """configs[2] (128^3, c = 1/2, solve_in_object_3d.py:98) directly against the oracle; configs[3] (256^3, c = 1, IBN_3D.py:132): a 6-plane
    window of the gradient against the oracle run on the matching slab of the same fields (interior planes of the window only depend on
    the window's nodes), and the loss against the per-point kernel."""
    from diffnet_amd import _lib
    kw = dict(nsd=3, domain_size=128, ngp_1d=2)
    m = module(kw)
    o = _oracle(kw)
    shape = (1, 1, 128, 128, 128)
    u, nu, f = seeded(shape, 31), seeded(shape, 32) + 0.5, seeded(shape, 33)
    bc = boundary_mask(shape)
    l, g = m.energy_loss_and_grad(cu(u), cu(nu), cu(f), dirichlet=[(cu(bc).to(torch.uint8), 0.0)], c=0.5)
    ur = u.clone().requires_grad_(True)
    ref = o.energy(ur, nu, f, dirichlet=[(bc, 0.0)], c=0.5)
    (gref,) = torch.autograd.grad(ref, ur)
    np.testing.assert_allclose(float(l), float(ref), rtol=1e-5)
    close(g, gref.numpy(), rtol=1e-4, arel=1e-4)
    del g, gref, ur
    n = 256
    m2 = module(dict(nsd=3, domain_size=n, ngp_1d=2))
    shape = (1, 1, n, n, n)
    u, nu, f = seeded(shape, 41), seeded(shape, 42) + 0.5, seeded(shape, 43)
    bc = boundary_mask(shape)
    ud, nd, fd, bd = cu(u), cu(nu), cu(f), cu(bc).to(torch.uint8)
    l, g = m2.energy_loss_and_grad(ud, nd, fd, dirichlet=[(bd, 0.0)], c=1.0)
    _lib.config_set("Q1_3D_N2", "1")
    try:
        l1, g1 = m2.energy_loss_and_grad(ud, nd, fd, dirichlet=[(bd, 0.0)], c=1.0)
    finally:
        _lib.config_set("Q1_3D_N2", "")
    np.testing.assert_allclose(float(l), float(l1), rtol=5e-6)
    assert float((g - g1).abs().max()) <= 5e-6 * float(g1.abs().max())
    # window of planes z0 .. z0 + 5 against the oracle on that slab (same hz: lengths scaled with the node count)
    z0 = 100
    kws = dict(nsd=3, domain_sizes=(n, n, 6), domain_lengths=(1.0, 1.0, 5.0 / (n - 1)), domain_size=n, ngp_1d=2)
    os_ = _oracle(kws)
    us = u[:, :, z0:z0 + 6].clone().requires_grad_(True)
    refs = os_.energy(us, nu[:, :, z0:z0 + 6], f[:, :, z0:z0 + 6], dirichlet=[(bc[:, :, z0:z0 + 6], 0.0)], c=1.0)
    (gs,) = torch.autograd.grad(refs, us)
    # mean over the slab's 5 layers of elements vs. the mesh's n - 1 layers
    gs = gs * (5.0 / (n - 1))
    close(g[:, :, z0 + 1:z0 + 5], gs[:, :, 1:5].numpy(), rtol=1e-4, arel=1e-4)
