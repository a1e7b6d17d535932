"""Networks the BASELINE configs feed into the FEM loss.

Round-1 status: same constructor signatures, default initialisation order and `state_dict` keys as the reference, so
checkpoints and the seeded golden vectors carry over (tests/test_networks.py).  Hand-written HIP (networks/fused.py):
InstanceNorm + LeakyReLU/ReLU pairs, the U-Net / 3-D generator output blocks (Upsample -> [ZeroPad] -> Conv -> Sigmoid,
forward and all gradients) and the weight gradient of the 3-D 4^3 stride-2 (transposed) convolutions.  Forward and
input-gradient convolutions are MIOpen.  The FEM loss they feed is fused HIP."""
from .autoencoders import AE  # noqa: F401
from .unets import UNet  # noqa: F401
from .wgan3d import GoodGenerator  # noqa: F401
