"""2-D U-Net generator (reference: DiffNet/networks/unets.py:13-81): five stride-2 4x4 conv stages
(in->32->64->128->256->256, InstanceNorm except the first, LeakyReLU 0.2, Dropout 0.5 on the last two), four
stride-2 4x4 transposed-conv stages with skip concatenation, then Upsample x2 -> ZeroPad(1,0,1,0) -> Conv 4x4 pad 1 ->
Sigmoid.  Input sizes must be multiples of 32 (>= 64: InstanceNorm needs more than one pixel at the bottleneck)."""
import torch
from torch import nn

from . import fused
from .fused import Conv2dS2, ConvTranspose2dS2, InstanceNormAct, upsample_pad_conv4


def _down(cin, cout, normalize=True, dropout=0.0):
    layers = [Conv2dS2(cin, cout, 4, 2, 1, bias=False)]          # HIP: csrc/conv2d_k4s2.hip (fp32 MFMA) on the GPU
    if normalize:                                  # fused InstanceNorm + LeakyReLU (HIP); Identity keeps the reference's indices
        layers += [InstanceNormAct(cout, slope=0.2), nn.Identity()]
    else:
        layers.append(nn.LeakyReLU(0.2))
    if dropout:
        layers.append(nn.Dropout(dropout))
    return nn.Sequential(*layers)


def _up(cin, cout, dropout=0.0):
    layers = [ConvTranspose2dS2(cin, cout, 4, 2, 1, bias=False), InstanceNormAct(cout, slope=0.0), nn.Identity()]
    if dropout:
        layers.append(nn.Dropout(dropout))
    return nn.Sequential(*layers)


class UNetDown(nn.Module):
    def __init__(self, in_size, out_size, normalize=True, dropout=0.0):
        super().__init__()
        self.model = _down(in_size, out_size, normalize, dropout)

    def forward(self, x):
        return self.model(x)


class UNetUp(nn.Module):
    def __init__(self, in_size, out_size, dropout=0.0):
        super().__init__()
        self.model = _up(in_size, out_size, dropout)

    def forward(self, x, skip_input):
        return torch.cat((self.model(x), skip_input), 1)


class UNet(nn.Module):
    def __init__(self, in_channels=3, out_channels=1):
        super().__init__()
        widths = [(in_channels, 32, False, 0.0), (32, 64, True, 0.0), (64, 128, True, 0.0), (128, 256, True, 0.5), (256, 256, True, 0.5)]
        for i, (ci, co, norm, drop) in enumerate(widths, start=1):
            setattr(self, f"down{i}", UNetDown(ci, co, normalize=norm, dropout=drop))
        for i, (ci, co, drop) in enumerate([(256, 256, 0.5), (512, 128, 0.5), (256, 64, 0.0), (128, 32, 0.0)], start=1):
            setattr(self, f"up{i}", UNetUp(ci, co, dropout=drop))
        self.final = nn.Sequential(nn.Upsample(scale_factor=2), nn.ZeroPad2d((1, 0, 1, 0)), nn.Conv2d(64, out_channels, 4, padding=1),
                                   nn.Sigmoid())

    def forward(self, x):
        d = [x]
        for i in range(1, 6):
            d.append(getattr(self, f"down{i}")(d[-1]))
        u = d[5]
        for i in range(1, 5):
            u = getattr(self, f"up{i}")(u, d[5 - i])
        if fused._hip(u):
            # Upsample -> ZeroPad -> Conv(64 -> out, 4x4) -> Sigmoid as one HIP kernel each way (dn_upconv_out_*); the
            # modules in self.final stay the parameter holders (state_dict keys final.2.weight / final.2.bias)
            conv = self.final[2]
            return upsample_pad_conv4(u, conv.weight, conv.bias, sigmoid=True)
        return self.final(u)
