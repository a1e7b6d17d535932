"""3-D U-Net generator `GoodGenerator` (reference: DiffNet/networks/wgan3d.py:23-98): Conv3d 4^3 stride 2 x5
(in->16->32->64->128->128; InstanceNorm3d except the last, Dropout 0.5 on the 4th), ConvTranspose3d x4 with skips,
Upsample x2 -> Conv3d 3^3 pad 1 -> Sigmoid.  Default torch initialisation (the reference defines but never applies
`weights_init_normal`)."""
import torch
from torch import nn

from . import fused
from .fused import Conv3dS2, ConvTranspose3dS2, InstanceNormAct, upsample_conv3


def _down_layers(cin, cout, normalize, dropout):
    """Conv3d 4^3 stride 2 -> [InstanceNorm + LeakyReLU 0.2 | LeakyReLU 0.2] -> [Dropout]; the Identity keeps the reference's
    Sequential indices where its separate activation module sat."""
    layers = [Conv3dS2(cin, cout, 4, 2, 1, bias=False)]            # nn.Conv3d + HIP weight gradient
    layers += [InstanceNormAct(cout, slope=0.2), nn.Identity()] if normalize else [nn.LeakyReLU(0.2)]
    return layers + ([nn.Dropout(dropout)] if dropout else [])


class UNetDown(nn.Module):
    def __init__(self, in_size, out_size, normalize=True, dropout=0.0):
        super().__init__()
        self.model = nn.Sequential(*_down_layers(in_size, out_size, normalize, dropout))

    def forward(self, x):
        return self.model(x)


class UNetUp(nn.Module):
    """ConvTranspose3d 4^3 stride 2 -> InstanceNorm + ReLU -> [Dropout], then concatenation with the skip tensor."""

    def __init__(self, in_size, out_size, dropout=0.0):
        super().__init__()
        stack = [ConvTranspose3dS2(in_size, out_size, 4, 2, 1, bias=False), InstanceNormAct(out_size, slope=0.0), nn.Identity()]
        self.model = nn.Sequential(*(stack + ([nn.Dropout(dropout)] if dropout else [])))

    def forward(self, x, skip_input):
        return torch.cat((self.model(x), skip_input), 1)


class GoodGenerator(nn.Module):
    # (attribute, in, out, normalize, dropout) in construction order = the reference's RNG consumption order
    _DOWN = (("down1", None, 16, True, 0.0), ("down2", 16, 32, True, 0.0), ("down3", 32, 64, True, 0.0), ("down4", 64, 128, True, 0.5),
             ("down5", 128, 128, False, 0.0))
    _UP = (("up3", 128, 128, 0.5), ("up4", 256, 64, 0.5), ("up5", 128, 32, 0.0), ("up6", 64, 16, 0.0))

    def __init__(self, in_channels=1, out_channels=3):
        super().__init__()
        for name, cin, cout, norm, drop in self._DOWN:
            setattr(self, name, UNetDown(in_channels if cin is None else cin, cout, normalize=norm, dropout=drop))
        for name, cin, cout, drop in self._UP:
            setattr(self, name, UNetUp(cin, cout, dropout=drop))
        self.final = nn.Sequential(nn.Upsample(scale_factor=2), nn.Conv3d(32, out_channels, 3, padding=1), nn.Sigmoid())

    def forward(self, x):
        skips = []
        for name, *_ in self._DOWN:
            x = getattr(self, name)(x)
            skips.append(x)
        u = skips.pop()                                # bottleneck
        for name, *_ in self._UP:
            u = getattr(self, name)(u, skips.pop())
        if fused._hip(u):
            # Upsample -> Conv3d(32 -> out, 3^3) -> Sigmoid as one HIP kernel each way (dn_upconv3d_out_*); self.final keeps
            # the parameters (state_dict keys final.1.weight / final.1.bias)
            conv = self.final[1]
            return upsample_conv3(u, conv.weight, conv.bias, sigmoid=True)
        return self.final(u)
