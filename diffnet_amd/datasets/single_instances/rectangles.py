"""Single-instance problems on the unit square (reference: DiffNet/datasets/single_instances/rectangles.py:7-425).
Channel order [domain (= nu), bc1 (u = 1), bc2 (u = 0)]; array axis 0 is y, axis 1 is x."""
import math

import numpy as np
import torch

from .. import GridDataset, faces, unit_grid


class _Square(GridDataset):
    def _setup(self, n, bc1=(), bc2=(), n_samples=100):
        self.domain = np.ones((n, n))
        self.bc1 = faces((n, n), *bc1)
        self.bc2 = faces((n, n), *bc2)
        self.n_samples = n_samples


class Rectangle(_Square):
    """Source on the first row, sink on the last (rectangles.py:7-31)."""

    def __init__(self, domain_size=64):
        self._setup(domain_size, bc1=[(0, 0)], bc2=[(0, -1)], n_samples=6000)


class RectangleManufactured(_Square):
    """u = sin(pi x) sin(pi y): f = 2 pi^2 u, homogeneous Dirichlet everywhere (rectangles.py:35-64)."""

    def __init__(self, domain_size=64):
        self._setup(domain_size, bc2=("all",))
        self.xx, self.yy = unit_grid(domain_size)
        self.forcing = 2. * math.pi ** 2 * np.sin(math.pi * self.xx) * np.sin(math.pi * self.yy)


class SpaceTimeRectangleManufactured(_Square):
    """Heat equation in space-time, y = time (rectangles.py:66-101).  Draws the same random numbers as the reference:
    one numpy normal field (the `domain` channel), then one torch uniform field (initial-guess noise)."""

    def __init__(self, domain_size=64):
        n = domain_size
        self._setup(n, bc1=[(0, 0)], bc2=[(1, 0), (1, -1)])
        xx, yy = unit_grid(n)
        self.decay_rt = 0.5
        self.u0 = torch.FloatTensor(np.sin(math.pi * xx) * np.exp(-self.decay_rt * yy))
        self.diffusivity = 0.1
        self.forcing = np.zeros_like(xx)
        self.domain = np.random.normal(0, 1., size=(n, n))
        self.initial_guess = torch.FloatTensor(np.tile(self.u0[0, :], (n, 1))) + 0.1 * torch.rand((n, n), requires_grad=False)


class AdvDiff1dRectangle(_Square):
    """rectangles.py:103-136: u = 0 on the left and right columns, unit forcing."""

    def __init__(self, domain_size=64):
        self._setup(domain_size, bc2=[(1, 0), (1, -1)])
        self.xx, self.yy = unit_grid(domain_size)
        self.forcing = np.ones_like(self.xx)


class AdvDiff2dRectangle(_Square):
    """rectangles.py:138-174: inflow column split at 20 % of the height, u = 0 on the first row."""

    def __init__(self, domain_size=64):
        n = domain_size
        self._setup(n, bc2=[(0, 0)])
        cut = int(0.2 * n)
        self.bc1[cut:, 0] = 1
        self.bc2[:cut, 0] = 1
        self.xx, self.yy = unit_grid(n)
        self.forcing = np.zeros_like(self.xx)


class AllenCahnIceMeltRectangle(_Square):
    """rectangles.py:176-222: tanh interface profile as the t = 0 row and as the initial guess."""

    def __init__(self, domain_size=64):
        n = domain_size
        self.ac_A, self.ac_Cn, self.ac_D, self.ac_k = 16., 0.1, 1., 2.
        self._setup(n, bc1=[(0, 0)])
        x = np.linspace(0, 1, n)
        self.xx, self.yy = unit_grid(n)
        thickness = self.ac_Cn * np.sqrt(2. / self.ac_A)
        u_t0 = (0.5 + 0.5 * np.tanh((x - 0.5) / thickness))[np.newaxis, :]
        self.u0 = torch.zeros((n, n))
        self.u0[0, :] = torch.FloatTensor(u_t0)
        self.initial_guess = np.tile(u_t0, (n, 1))
        self.forcing = np.zeros_like(self.xx)


class RectangleManufacturedNonZeroBC(_Square):
    """Laplace problem with u = exp(-pi x) sin(pi y) (rectangles.py:222-256)."""

    def __init__(self, domain_size=64):
        self._setup(domain_size, bc1=[(1, 0), (1, -1)], bc2=[(0, -1), (0, 0)])
        self.xx, self.yy = unit_grid(domain_size)
        self.om = np.pi
        self.u_exact = np.exp(-self.om * self.xx) * np.sin(self.om * self.yy)
        self.forcing = np.zeros_like(self.xx)


class RectangleHelmholtzManufactured(_Square):
    """rectangles.py:258-289: f = (2 pi^2 - k^2) sin(pi x) sin(pi y), k = 0.5."""

    def __init__(self, domain_size=64):
        self.khh = 0.5
        self._setup(domain_size, bc2=("all",))
        xx, yy = unit_grid(domain_size)
        self.forcing = (2. * math.pi ** 2 - self.khh ** 2) * np.sin(math.pi * xx) * np.sin(math.pi * yy)


class RectangleHelmholtzDeltaForce(_Square):
    """rectangles.py:291-326: Gaussian point source at (0.1875, 0.1875), sigma 0.05, k = 1/8."""

    def __init__(self, domain_size=64):
        self.khh = 1. / 8.
        self._setup(domain_size, bc2=("all",))
        xx, yy = unit_grid(domain_size)
        mu, sigma = 0.1875, 0.05
        self.forcing = np.exp(-0.5 * ((xx - mu) / sigma) ** 2 - 0.5 * ((yy - mu) / sigma) ** 2) / (2 * np.pi * sigma * sigma)


class RectangleManufacturedStokes(_Square):
    """rectangles.py:328-363 (bc3 / bc4 are allocated and not returned, as in the reference)."""

    def __init__(self, domain_size=64):
        self._setup(domain_size, bc2=[(0, -1), (0, 0)])
        self.bc3 = np.zeros((domain_size, domain_size))
        self.bc4 = np.zeros((domain_size, domain_size))
        xx, yy = unit_grid(domain_size)
        self.forcing = 2. * math.pi ** 2 * np.sin(math.pi * xx) * np.sin(math.pi * yy)


class RectangleIM(GridDataset):
    """Immersed rectangle x0 = y0 = 10, 30 x 50: domain = 1 inside, source on its first row, sink on the row below its
    last (rectangles.py:365-392)."""

    def __init__(self, domain_size=64):
        n = domain_size
        x0, y0, w, h = 10, 10, 30, 50
        self.domain = np.zeros((n, n))
        self.domain[y0:y0 + h, x0:x0 + w] = 1.0
        self.bc1 = np.zeros((n, n))
        self.bc1[y0, x0:x0 + w] = 1
        self.bc2 = np.zeros((n, n))
        self.bc2[y0 + h, x0:x0 + w] = 1
        self.n_samples = 200


class RectangleIMBack(GridDataset):
    """Background mesh with a 30 x 20 rectangular inclusion held at u = 1, u = 0 on the outer boundary
    (rectangles.py:394-425)."""

    def __init__(self, domain_size=64):
        n = domain_size
        x0, y0, w, h = 10, 10, 30, 20
        self.domain = np.ones((n, n))
        self.domain[y0:y0 + h, x0:x0 + w] = 0.0
        self.bc1 = np.zeros((n, n))
        self.bc1[y0:y0 + h, x0:x0 + w] = 1.0
        self.bc2 = faces((n, n), "all")
        self.n_samples = 200
