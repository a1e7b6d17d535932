"""KL-sum diffusivity families (reference: DiffNet/datasets/parametric/klsum.py:11-79)."""
import numpy as np

from .. import StackedDataset, faces
from ..single_instances.klsum import Dataset as _SingleKL
from ...gen_input_calc import calculate_omega_based_on_eta, grid2D


class KLSumStochastic(StackedDataset):
    """One field per row of the `.npy` coefficient table (e.g. a Sobol sequence (65536, 6)); u = 1 on the first column,
    u = 0 on the last.  The trigonometric factors are the same for every sample, so they are tabulated once and the
    table is evaluated in blocks of samples (sum over the terms in the reference's order) instead of one Python call per
    sample; samples are stored as float32, which is what `__getitem__` hands out."""

    def __init__(self, filename, domain_size=64, kl_terms=6, eta=0.5, block=1024):
        self.coeffs = np.load(filename)
        self.domain_size, self.kl_terms = domain_size, kl_terms
        n = domain_size
        x, y = grid2D(n, n)
        omega = calculate_omega_based_on_eta(eta)
        lam = 2.0 * eta / (1.0 + (eta * omega) ** 2)
        terms = [np.sqrt(lam[i]) * np.sqrt(lam[i]) * (eta * omega[i] * np.cos(omega[i] * x) + np.sin(omega[i] * x))
                 * (eta * omega[i] * np.cos(omega[i] * y) + np.sin(omega[i] * y)) for i in range(min(kl_terms, 6))]
        coeffs = np.asarray(self.coeffs, dtype=np.float64).reshape(len(self.coeffs), -1)
        self.dataset = np.empty((len(coeffs), 3, n, n), dtype=np.float32)
        self.dataset[:, 1] = faces((n, n), (1, 0))
        self.dataset[:, 2] = faces((n, n), (1, -1))
        for lo in range(0, len(coeffs), block):
            a = coeffs[lo:lo + block]
            total = np.zeros((len(a), n, n))
            for i, t in enumerate(terms):
                total += a[:, i, None, None] * t
            self.dataset[lo:lo + block, 0] = np.exp(total)
        self.n_samples = self.dataset.shape[0]


class Dataset(_SingleKL):
    """The parametric module's single-field dataset (klsum.py:50-79): as the single-instance one with 100 samples."""

    n_samples_default = 100
