"""One KL-sum diffusivity field from a coefficient file (reference: DiffNet/datasets/single_instances/klsum.py:7-35)."""
import os

import numpy as np

from .. import GridDataset, faces
from ...gen_input_calc import generate_diffusivity_tensor


class Dataset(GridDataset):
    """nu = exp(KL sum) from `np.loadtxt(coeff_file)`; u = 1 on the first column, u = 0 on the last."""

    n_samples_default = 1000

    def __init__(self, coeff_file, domain_size=64):
        if not os.path.exists(coeff_file):
            raise FileNotFoundError("Single instance: Wrong path to coefficient file.")
        self.coeff = np.loadtxt(coeff_file, dtype=np.float32)
        self.domain_size = domain_size
        self.nu = generate_diffusivity_tensor(self.coeff, output_size=domain_size).squeeze()
        self.bc1 = faces((domain_size, domain_size), (1, 0))
        self.bc2 = faces((domain_size, domain_size), (1, -1))
        self.n_samples = self.n_samples_default

    def channels(self):
        return [self.nu, self.bc1, self.bc2]

    def forcing_array(self):
        return np.zeros_like(self.nu)
