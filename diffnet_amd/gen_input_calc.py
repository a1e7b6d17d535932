"""Karhunen-Loeve sum log-diffusivity fields (reference: DiffNet/gen_input_calc.py:3-183) -- host-side input
generation for the KL-sum datasets.

    nu(x, y) = exp( sum_{i<6} a_i sqrt(lam_x,i lam_y,i) (eta w_i cos(w_i x) + sin(w_i x)) (eta w_i cos(w_i y) + sin(w_i y)) )

with w_i the positive roots of (eta^2 w^2 - 1) sin w - 2 eta w cos w = 0 (exponential covariance on [0, 1]) and
lam_i = 2 eta / (1 + eta^2 w_i^2).  The reference tabulates ten roots for eta in {0.1, 0.2, 0.5, 0.7, 1.0}; here they are
computed (bracketing + bisection/Newton in float64), which reproduces the tabulated digits and works for any eta."""
import functools
import math

import numpy as np


@functools.lru_cache(maxsize=None)
def _omega_roots(eta, n=10):
    def g(w):
        return (eta * eta * w * w - 1.0) * math.sin(w) - 2.0 * eta * w * math.cos(w)

    roots, w, step = [], 1e-9, 1e-3
    prev = g(w)
    while len(roots) < n:
        w2 = w + step
        cur = g(w2)
        if prev == 0.0 or (prev < 0.0) != (cur < 0.0):
            lo, hi, flo = w, w2, prev
            for _ in range(200):
                mid = 0.5 * (lo + hi)
                fm = g(mid)
                if (fm < 0.0) == (flo < 0.0):
                    lo, flo = mid, fm
                else:
                    hi = mid
                if hi - lo < 1e-15 * max(1.0, hi):
                    break
            roots.append(0.5 * (lo + hi))
        w, prev = w2, cur
    return tuple(roots)


def calculate_omega_based_on_eta(eta):
    """First ten positive roots w of the KL eigen-equation for correlation length eta."""
    return np.array(_omega_roots(round(float(eta), 12)))


def _kl_factor(coord, eta, omega):
    return eta * omega * np.cos(omega * coord) + np.sin(omega * coord)


def construct_KL_sum_2D(x, y, rand_tensor_list, eta_x=0.5, eta_y=0.5):
    ox, oy = calculate_omega_based_on_eta(eta_x), calculate_omega_based_on_eta(eta_y)
    lx = 2.0 * eta_x / (1.0 + (eta_x * ox) ** 2)
    ly = 2.0 * eta_y / (1.0 + (eta_y * oy) ** 2)
    total = 0 * x
    for i in range(6):
        total = total + rand_tensor_list[i] * np.sqrt(lx[i]) * np.sqrt(ly[i]) * _kl_factor(x, eta_x, ox[i]) * _kl_factor(y, eta_y, oy[i])
    return total


def construct_KL_sum_3D(x, y, z, rand_tensor_list, eta_x=0.5, eta_y=0.5, eta_z=0.5):
    ox, oy, oz = (calculate_omega_based_on_eta(e) for e in (eta_x, eta_y, eta_z))
    lx = 2.0 * eta_x / (1.0 + (eta_x * ox) ** 2)
    ly = 2.0 * eta_y / (1.0 + (eta_y * oy) ** 2)
    lz = 2.0 * eta_z / (1.0 + (eta_z * oz) ** 2)
    total = 0 * x
    for i in range(6):
        total = total + (rand_tensor_list[i] * np.sqrt(lx[i]) * np.sqrt(ly[i]) * np.sqrt(lz[i]) * _kl_factor(x, eta_x, ox[i])
                         * _kl_factor(y, eta_y, oy[i]) * _kl_factor(z, eta_z, oz[i]))
    return total


def grid2D(nx, ny):
    return tuple(np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1, ny)))


def grid3D(nx, ny, nz):
    return tuple(np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1, ny), np.linspace(0, 1, nz)))


def generate_diffusivity_tensor(coeff, output_size=64, nsd=2, n_sum_nu=6):
    """exp(KL sum) on the unit square / cube grid, shape (1, n, n[, n]); coefficients beyond `n_sum_nu` count as zero."""
    a = [float(c) for c in np.asarray(coeff).tolist()[:n_sum_nu]]
    a += [0.0] * (6 - len(a))
    n = output_size
    if nsd == 2:
        x, y = (g[None] for g in grid2D(n, n))
        return np.exp(construct_KL_sum_2D(x, y, a))
    x, y, z = (g[None] for g in grid3D(n, n, n))
    return np.exp(construct_KL_sum_3D(x, y, z, a))
