"""L-shaped domain (reference: DiffNet/datasets/single_instances/Lshaped.py:8-42)."""
import copy

import numpy as np

from .. import GridDataset


class LShaped(GridDataset):
    """Union of a 50 x 20 and a 20 x 50 block anchored at (5, 5); u = 0 on the six boundary segments; f = 10 inside."""

    def __init__(self, domain_size=64):
        n = domain_size
        r0, c0, long_, short = 5, 5, 50, 20
        self.domain = np.zeros((n, n))
        self.domain[r0:r0 + long_, c0:c0 + short] = 1.0
        self.domain[r0:r0 + short, c0:c0 + long_] = 1.0
        self.bc1 = np.zeros((n, n))
        b = np.zeros((n, n))
        b[r0:r0 + long_, c0] = 1
        b[r0 + long_, c0:c0 + short] = 1
        b[r0 + short:r0 + long_, c0 + short] = 1
        b[r0 + short, c0 + short:c0 + long_] = 1
        b[r0:r0 + short, c0 + long_] = 1
        b[r0, c0:c0 + long_] = 1
        self.bc2 = b
        self.n_samples = 200
        self.forcing = copy.deepcopy(self.domain) * 10
