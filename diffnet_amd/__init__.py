"""diffnet_amd -- MI355X-native implementation of DiffNet's FEM Gauss-quadrature loss/residual hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, autograd glue); all arithmetic of the path runs
in hand-written HIP kernels for gfx950 behind the C ABI of include/diffnet_hip.h (libdiffnet_hip.so).
"""
__version__ = "0.1.0"

from .base import PDE  # noqa: E402,F401
from .fem import DiffNet2DFEM, DiffNet3DFEM, DiffNetFEM, FemGeometry, gauss_pt_eval  # noqa: E402,F401
from .ops import Dirichlet, PackedMask, BoxFaces, LoadVector  # noqa: E402,F401
from . import torch_ops  # noqa: E402,F401  (registers the diffnet_mi:: operators with torch.library)
