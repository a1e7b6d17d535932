"""3-D U-Net generator `GoodGenerator` (reference: DiffNet/networks/wgan3d.py:23-98): Conv3d 4^3 stride 2 x5
(in->16->32->64->128->128; InstanceNorm3d except the last, Dropout 0.5 on the 4th), ConvTranspose3d x4 with skips,
Upsample x2 -> Conv3d 3^3 pad 1 -> Sigmoid.  Default torch initialisation (the reference defines but never applies
`weights_init_normal`)."""
import torch
from torch import nn

from .fused import Conv3dS2, ConvTranspose3dS2, InstanceNormAct, upsample_conv3


class UNetDown(nn.Module):
    def __init__(self, in_size, out_size, normalize=True, dropout=0.0):
        super().__init__()
        layers = [Conv3dS2(in_size, out_size, 4, 2, 1, bias=False)]           # nn.Conv3d + HIP weight gradient
        if normalize:                              # fused InstanceNorm + LeakyReLU (HIP); Identity keeps the reference's indices
            layers += [InstanceNormAct(out_size, slope=0.2), nn.Identity()]
        else:
            layers.append(nn.LeakyReLU(0.2))
        if dropout:
            layers.append(nn.Dropout(dropout))
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        return self.model(x)


class UNetUp(nn.Module):
    def __init__(self, in_size, out_size, dropout=0.0):
        super().__init__()
        layers = [ConvTranspose3dS2(in_size, out_size, 4, 2, 1, bias=False), InstanceNormAct(out_size, slope=0.0), nn.Identity()]
        if dropout:
            layers.append(nn.Dropout(dropout))
        self.model = nn.Sequential(*layers)

    def forward(self, x, skip_input):
        return torch.cat((self.model(x), skip_input), 1)


class GoodGenerator(nn.Module):
    def __init__(self, in_channels=1, out_channels=3):
        super().__init__()
        self.down1 = UNetDown(in_channels, 16)
        self.down2 = UNetDown(16, 32)
        self.down3 = UNetDown(32, 64)
        self.down4 = UNetDown(64, 128, dropout=0.5)
        self.down5 = UNetDown(128, 128, normalize=False)
        self.up3 = UNetUp(128, 128, dropout=0.5)
        self.up4 = UNetUp(256, 64, dropout=0.5)
        self.up5 = UNetUp(128, 32)
        self.up6 = UNetUp(64, 16)
        self.final = nn.Sequential(nn.Upsample(scale_factor=2), nn.Conv3d(32, out_channels, 3, padding=1), nn.Sigmoid())

    def forward(self, x):
        d1 = self.down1(x)
        d2 = self.down2(d1)
        d3 = self.down3(d2)
        d4 = self.down4(d3)
        d5 = self.down5(d4)
        u = self.up3(d5, d4)
        u = self.up4(u, d3)
        u = self.up5(u, d2)
        u = self.up6(u, d1)
        if u.is_cuda and u.dtype == torch.float32:
            # Upsample -> Conv3d(32 -> out, 3^3) -> Sigmoid as one HIP kernel each way (dn_upconv3d_out_*); self.final keeps
            # the parameters (state_dict keys final.1.weight / final.1.bias)
            conv = self.final[1]
            return upsample_conv3(u, conv.weight, conv.bias, sigmoid=True)
        return self.final(u)
