"""Disk inclusion on a background mesh (reference: DiffNet/datasets/single_instances/circles.py:8-41)."""
import numpy as np

from .. import GridDataset, faces


class CircleIMBack(GridDataset):
    """Disk of radius 15 centred at (x, y) = (15, 40) in grid units: u = 1 strictly inside, domain = 1 strictly outside,
    u = 0 on the outer boundary."""

    def __init__(self, domain_size=64):
        n = domain_size
        cx, cy, r = 15, 40, 15
        t = np.linspace(0, 1, n) * n
        xx, yy = np.meshgrid(t, t)
        level = (xx - cx) ** 2 + (yy - cy) ** 2 - r ** 2
        self.domain = np.zeros((n, n))
        self.domain[level > 0.0] = 1.0
        self.bc1 = np.zeros((n, n))
        self.bc1[level < 0.0] = 1
        self.bc2 = faces((n, n), "all")
        self.n_samples = 100
