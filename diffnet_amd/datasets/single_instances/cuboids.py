"""Single-instance problems on the unit cube (reference: DiffNet/datasets/single_instances/cuboids.py:8-80)."""
import math

import numpy as np

from .. import GridDataset, faces
from ...cuboid_mesh import CuboidMesh


class Cuboid(GridDataset):
    """Source on the first plane of axis 0, sink on the last (cuboids.py:8-33)."""

    def __init__(self, domain_size=64):
        n = domain_size
        self.domain = np.ones((n, n, n))
        self.bc1 = faces((n, n, n), (0, 0))
        self.bc2 = faces((n, n, n), (0, -1))
        self.n_samples = 100


class CuboidManufactured(GridDataset):
    """u = sin(pi x) sin(3 pi y) sin(3 pi z): f = 19 pi^2 u, homogeneous Dirichlet on all six faces (cuboids.py:37-80)."""

    def __init__(self, domain_size=64):
        n = domain_size
        self.domain = np.ones((n, n, n))
        self.bc1 = np.zeros((n, n, n))
        self.bc2 = faces((n, n, n), "all")
        self.n_samples = 100
        t = np.linspace(0, 1, n)
        self.xx, self.yy, self.zz = CuboidMesh.meshgrid_3d(t, t, t)
        self.forcing = self.forcing(self.xx, self.yy, self.zz)      # the method is replaced by its value, as in the reference

    def forcing(self, x, y, z):
        pi = math.pi
        return 19. * pi ** 2 * np.sin(pi * x) * np.sin(3. * pi * y) * np.sin(3 * pi * z)
