"""The plain-C float64 restatement (oracle/fem_oracle.c) agrees with the torch oracle (which is pinned against the
reference's golden vectors) -- two independent restatements of the same algorithm."""
import numpy as np
import pytest
import torch

from oracle import c_oracle
from oracle.fem_oracle import Oracle


@pytest.mark.parametrize("kw,B", [
    (dict(domain_size=17), 2), (dict(domain_size=17, ngp_1d=3), 1), (dict(domain_size=13, fem_basis_deg=2), 2),
    (dict(domain_size=13, fem_basis_deg=3, ngp_1d=4), 1), (dict(domain_size=9, nsd=3), 2),
    (dict(domain_sizes=(8, 6, 5), domain_lengths=(2.0, 1.0, 0.5), domain_size=8, domain_length=2.0, nsd=3, ngp_1d=3), 1),
    (dict(domain_size=5, nsd=3, fem_basis_deg=2), 1),
])
def test_c_oracle_matches_torch_oracle(kw, B):
    o = Oracle(**kw)
    g = torch.Generator().manual_seed(5)
    shape = (B, 1, *o.spec.sizes[::-1])
    u, nu, f = torch.rand(shape, generator=g), 0.5 + torch.rand(shape, generator=g), torch.rand(shape, generator=g)
    mask = (torch.rand(shape, generator=g) < 0.2).float()
    ur = u.double().requires_grad_(True)
    od = Oracle(**kw)
    od.t = {k: v.double() for k, v in od.t.items()}       # float64 tables rounded from float32: compare like with like
    od.gpw = od.t["gpw"]
    ref = od.energy(ur, nu.double(), f.double(), dirichlet=[(mask.double(), 1.0)], c=0.5, jac=0.7)
    (gref,) = torch.autograd.grad(ref, ur)
    loss, grad = c_oracle.energy(o.spec, u.numpy(), nu.numpy(), f.numpy(), mask.numpy(), 1.0, c=0.5, jac=0.7)
    np.testing.assert_allclose(loss, float(ref), rtol=2e-6)            # float32-rounded tables vs exact products
    np.testing.assert_allclose(grad, gref.numpy(), rtol=1e-5, atol=2e-6 * float(gref.abs().max()))
