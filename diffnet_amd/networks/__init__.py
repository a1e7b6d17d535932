"""Networks the BASELINE configs feed into the FEM loss.

Round-1 status: the layer stacks are declared with stock torch modules (MIOpen convolutions on ROCm), with the same
constructor signatures, default initialisation order and `state_dict` keys as the reference, so checkpoints and the
seeded golden vectors carry over (tests/test_networks.py).  Hand-written HIP kernels for the 4x4-stride-2
conv / conv-transpose + InstanceNorm + activation blocks are the next step; the FEM loss they
feed is already fused HIP."""
from .autoencoders import AE  # noqa: F401
from .unets import UNet  # noqa: F401
from .wgan3d import GoodGenerator  # noqa: F401
