"""Networks the BASELINE configs feed into the FEM loss.

Round-1 status: same constructor signatures, default initialisation order and `state_dict` keys as the reference, so
checkpoints and the seeded golden vectors carry over (tests/test_networks.py).  InstanceNorm + LeakyReLU/ReLU pairs
run as ONE hand-written HIP kernel forward and one backward (`dn_instnorm_act_fwd/bwd`, networks/fused.py); the
convolutions / transposed convolutions are still MIOpen (implicit-GEMM HIP kernels for the 4x4-stride-2 blocks are the
next step).  The FEM loss they feed is fused HIP."""
from .autoencoders import AE  # noqa: F401
from .unets import UNet  # noqa: F401
from .wgan3d import GoodGenerator  # noqa: F401
