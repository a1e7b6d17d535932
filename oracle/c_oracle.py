"""ctypes wrapper of oracle/fem_oracle.c (float64 direct-loop checker) -- TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(HERE, "libfem_oracle_c.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(HERE, "fem_oracle.c")):
            subprocess.check_call(["make", "-s", "-C", HERE])
        _LIB = C.CDLL(so)
        _LIB.dn_oracle_energy.restype = C.c_double
    return _LIB


def energy(spec, u, nu=None, f=None, mask=None, mask_value=0.0, c=1.0, jac=1.0):
    """spec: oracle.fem_oracle.FemSpec; arrays (B,1,*N) convertible to float64.  Returns (loss, grad float64)."""
    from .fem_oracle import basis_1d
    u = np.ascontiguousarray(u, dtype=np.float64)
    B = u.shape[0]
    n = (C.c_int * 3)(*(list(spec.sizes) + [1] * (3 - spec.nsd)))
    Bv = np.ascontiguousarray([basis_1d(spec.deg, float(x))[0] for x in spec.gpx_1d], dtype=np.float64)
    Dv = np.ascontiguousarray([basis_1d(spec.deg, float(x))[1] for x in spec.gpx_1d], dtype=np.float64)
    w = np.ascontiguousarray(spec.gpw_1d, dtype=np.float64)
    sc = (C.c_double * 3)(*([2.0 / h for h in spec.hs] + [1.0] * (3 - spec.nsd)))
    grad = np.empty_like(u)

    def p(a):
        if a is None:
            return None
        a = np.ascontiguousarray(np.broadcast_to(a, u.shape), dtype=np.float64)
        keep.append(a)
        return a.ctypes.data_as(C.c_void_p)

    keep = []
    loss = lib().dn_oracle_energy(C.c_int(spec.nsd), C.c_int(spec.nbf_1d), C.c_int(spec.ngp_1d), C.c_int(B), n,
                                  Bv.ctypes.data_as(C.c_void_p), Dv.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p), sc,
                                  u.ctypes.data_as(C.c_void_p), p(nu), p(f), p(mask), C.c_double(mask_value), C.c_double(c),
                                  C.c_double(jac), grad.ctypes.data_as(C.c_void_p))
    return loss, grad
