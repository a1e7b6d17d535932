"""GPU parity of the compact Dirichlet forms (C ABI DN_MASK_BITS / DN_MASK_BOX, include/diffnet_hip.h): bit-packed masks and
geometry-derived box faces must give EXACTLY the numbers of the reference's mask images (`torch.where(mask > 0.5, value, u)`,
IBN/poisson-2d/parametric/IBN_2D.py:119-121; the sink mask of IBN_2D.py:69-73 is the box boundary) -- only the source of the
condition bits differs, the arithmetic is the same kernel.  The oracle comparison pins the compact forms to the reference too."""
import numpy as np
import pytest
import torch

from test_gpu_parity import boundary_mask, close, cu, dev, module, seeded

pytestmark = pytest.mark.gpu


def rand_mask(shape, seed, p=0.05):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) < p)


@pytest.mark.parametrize("nx,B,batched", [(512, 3, True), (64, 2, False), (65, 2, True), (100, 1, True), (31, 2, True), (257, 2, False)])
def test_pack_unpack_roundtrip_and_layout(nx, B, batched):
    from diffnet_amd import PackedMask
    ny = 37
    img = rand_mask((B if batched else 1, 1, ny, nx), 5, 0.3)
    for src in (img.to(torch.uint8), img.to(torch.float32) * 0.9 + 0.05, img):          # u8, fp32 compared with 0.5 (0.95 / 0.05), bool
        pm = PackedMask.pack(src.to(dev()))
        assert pm.bits.shape == (img.shape[0], ny, (nx + 31) // 32) and pm.bits.dtype == torch.int32
        assert torch.equal(pm.image().cpu(), img.to(torch.uint8))
        # documented layout: node x of a row is bit (x & 31) of word (x >> 5)
        words = pm.bits.cpu().numpy().astype(np.uint32)
        ref = np.zeros_like(words)
        a = img.numpy()[:, 0]
        for x in range(nx):
            ref[:, :, x >> 5] |= a[:, :, x].astype(np.uint32) << np.uint32(x & 31)
        np.testing.assert_array_equal(words, ref)


CASES = [dict(domain_size=512, ngp_1d=3), dict(domain_size=64, ngp_1d=2), dict(domain_size=65, ngp_1d=3), dict(domain_size=130, ngp_1d=4),
         dict(domain_size=1024, ngp_1d=2), dict(domain_size=33, ngp_1d=2)]


@pytest.mark.parametrize("kw", CASES, ids=lambda k: f"n{k['domain_size']}_g{k['ngp_1d']}")
@pytest.mark.parametrize("with_nu_f", [True, False])
def test_compact_conditions_equal_the_mask_images_bitwise(kw, with_nu_f):
    from diffnet_amd import BoxFaces, PackedMask
    m = module(kw)
    n = kw["domain_size"]
    B = 3
    u = cu(seeded((B, 1, n, n), 1))
    nu = cu(seeded((B, 1, n, n), 2) + 0.5) if with_nu_f else None
    f = cu(seeded((B, 1, n, n), 3)) if with_nu_f else None
    src = rand_mask((B, 1, n, n), 7).to(torch.uint8).to(dev())        # immersed "source", per sample (IBN_2D.py:119)
    box = boundary_mask((1, 1, n, n)).to(torch.uint8).to(dev())        # exterior "sink" = the box boundary (IBN_2D.py:69-73)
    # (uint8 images: the fp32-image form of the kernel sums the forcing term per element instead of per row -- equal to rounding, not bitwise)
    ref_loss, ref_grad = m.energy_loss_and_grad(u, nu, f, dirichlet=[(src, 1.0), (box, 0.0)], c=1.0)
    forms = {"bits+box": [(PackedMask.pack(src), 1.0), (BoxFaces("all"), 0.0)],
             "bits+bits": [(PackedMask.pack(src), 1.0), (PackedMask.pack(box), 0.0)],
             "bits fp32 source": [(PackedMask.pack(src.float()), 1.0), (BoxFaces(["xlo", "xhi", "ylo", "yhi"]), 0.0)]}
    for name, d in forms.items():
        loss, grad = m.energy_loss_and_grad(u, nu, f, dirichlet=d, c=1.0)
        assert torch.equal(grad, ref_grad), name
        assert torch.equal(loss, ref_loss), name
    # one condition only, partial faces, against the image of those faces
    part = torch.zeros((1, 1, n, n), dtype=torch.uint8, device=dev())
    part[..., 0] = 1; part[..., -1, :] = 1                            # x = 0 and y = ny - 1
    r1 = m.energy_loss_and_grad(u, nu, f, dirichlet=[(part, 0.25)], c=0.5)
    g1 = m.energy_loss_and_grad(u, nu, f, dirichlet=[(BoxFaces(["xlo", "yhi"]), 0.25)], c=0.5)
    assert torch.equal(r1[1], g1[1]) and torch.equal(r1[0], g1[0])
    # residual form (alpha = beta = 1) on the same conditions
    assert torch.equal(m.residual(u, nu, f, dirichlet=[(src, 1.0), (box, 0.0)]), m.residual(u, nu, f, dirichlet=forms["bits+box"]))


def test_compact_conditions_vs_oracle_and_autograd_route():
    """The reference formulation (oracle) with mask images against the fused kernel fed with bits + box faces, through autograd."""
    from diffnet_amd import BoxFaces, PackedMask
    from oracle.fem_oracle import Oracle
    kw = dict(domain_size=48, ngp_1d=3)
    m, o = module(kw), Oracle(**kw)
    B, n = 2, 48
    u, nu, f = seeded((B, 1, n, n), 11), seeded((B, 1, n, n), 12) + 0.5, seeded((B, 1, n, n), 13)
    src = rand_mask((B, 1, n, n), 17).float()
    box = boundary_mask((1, 1, n, n)).float()
    ur = u.clone().requires_grad_(True)
    ub = torch.where(box > 0.5, torch.zeros_like(ur), torch.where(src > 0.5, torch.ones_like(ur), ur))
    ref = o.energy(ub, nu, f, c=1.0)
    ref.backward()
    ug = cu(u).requires_grad_(True)
    loss = m.energy_loss(ug, cu(nu), cu(f), dirichlet=[(PackedMask.pack(cu(src)), 1.0), (BoxFaces(), 0.0)], c=1.0)
    loss.backward()
    np.testing.assert_allclose(float(loss), float(ref), rtol=1e-5)
    close(ug.grad, ur.grad.numpy(), rtol=1e-4, arel=1e-4)


def test_compact_conditions_fall_back_to_images_where_the_kernel_has_no_compact_form():
    """3-D, Q2 and Gauss-point forcing have no bit / box path in the kernels: the host layer expands the condition to its image."""
    from diffnet_amd import BoxFaces, PackedMask
    m = module(dict(domain_size=17, nsd=3))
    u, nu = cu(seeded((2, 1, 17, 17, 17), 1)), cu(seeded((2, 1, 17, 17, 17), 2) + 0.5)
    box = boundary_mask((1, 1, 17, 17, 17)).to(dev())
    a = m.energy_loss_and_grad(u, nu, None, dirichlet=[(box, 0.0)])
    b = m.energy_loss_and_grad(u, nu, None, dirichlet=[(BoxFaces(), 0.0)])
    c = m.energy_loss_and_grad(u, nu, None, dirichlet=[(PackedMask.pack(box), 0.0)])
    assert torch.equal(a[1], b[1]) and torch.equal(a[1], c[1]) and torch.equal(a[0], b[0])
    m2 = module(dict(domain_size=33, fem_basis_deg=2, ngp_1d=3))
    u2 = cu(seeded((2, 1, 33, 33), 3))
    box2 = boundary_mask((1, 1, 33, 33)).to(dev())
    assert torch.equal(m2.residual(u2, dirichlet=[(box2, 0.0)]), m2.residual(u2, dirichlet=[(BoxFaces(), 0.0)]))


def test_c_abi_rejects_compact_conditions_it_cannot_run():
    import ctypes as C
    from diffnet_amd import _lib
    from diffnet_amd.fem import FemGeometry
    m = module(dict(domain_size=9, nsd=3))
    mesh = m.geom.mesh_struct(1)
    u = cu(seeded((1, 1, 9, 9, 9), 1))
    out = torch.empty_like(u)
    args = _lib.DnPoissonArgs()
    args.u, args.out = u.data_ptr(), out.data_ptr()
    args.alpha = args.beta = args.c = args.wscale = args.out_scale = 1.0
    args.bc[0].mask_kind, args.bc[0].box_faces = _lib.MASK_BOX, 63
    assert _lib.lib().dn_poisson_apply(C.byref(mesh), C.byref(args), None) == -2          # DN_E_UNSUPPORTED
    args.bc[0].mask_kind = 7
    assert _lib.lib().dn_poisson_apply(C.byref(mesh), C.byref(args), None) == -1          # DN_E_BADARG
