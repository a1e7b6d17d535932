"""Convolutional auto-encoder used by IBN_2D.py:186 (`AE(1, 1, n_downsample=2)`); reference:
DiffNet/networks/autoencoders.py:7-95.  Encoder: ReflPad3 -> Conv7x7(in -> 2*dim) -> InstanceNorm -> LeakyReLU, then
n_downsample x [Conv4x4 s2 -> InstanceNorm -> ReLU] with widths 2*dim*(i+1) -> 2*dim*(i+2), Tanh.  Decoder mirrors it
with transposed convolutions + LeakyReLU, then ReflPad4 -> Conv3x3 -> Conv7x7.  The reference leaves the final
activation commented out (:81) and so does this one."""
from torch import nn

from .fused import Conv2dS2, Conv2dValid, ConvTranspose2dS2, InstanceNormAct


def _widths(dim, i):
    """Channel widths around stride-2 stage i: 2 dim (i + 1) <-> 2 dim (i + 2) for i <= 3, 10 dim beyond."""
    return (dim * 2 * (i + 1), dim * 2 * (i + 2)) if i <= 3 else (dim * 10, dim * 10)


def _stage(conv, cout, slope):
    # convolution -> fused InstanceNorm + activation (HIP); the Identity holds the index of the reference's activation module
    return [conv, InstanceNormAct(cout, slope=slope), nn.Identity()]


class Encoder(nn.Module):
    def __init__(self, in_channels=3, dim=64, n_downsample=3, encoder_type='convolutional'):
        super().__init__()
        # the reference declares the first InstanceNorm with `dim` features after a 2*dim-channel conv: harmless
        # (affine=False); the number is kept for the repr only
        stem = [nn.ReflectionPad2d(3)] + _stage(Conv2dValid(in_channels, dim * 2, 7), dim, 0.2)
        downs = []
        for i in range(n_downsample):
            cin, cout = _widths(dim, i)
            downs += _stage(Conv2dS2(cin, cout, 4, stride=2, padding=1), cout, 0.0)
        self.model_blocks = nn.Sequential(*stem, *downs, nn.Tanh())

    def forward(self, x):
        return self.model_blocks(x)


class Decoder(nn.Module):
    def __init__(self, out_channels=3, dim=64, n_upsample=3, encoder_type='convolutional', activation='relu'):
        super().__init__()
        ups, i = [], 0
        for i in reversed(range(n_upsample)):
            cout, cin = _widths(dim, i)                   # mirrored: the decoder walks the widths backwards
            ups += _stage(ConvTranspose2dS2(cin, cout, 4, stride=2, padding=1), cout, 0.2)
        head = [nn.ReflectionPad2d(4), Conv2dValid(dim * (i + 1) * 2, out_channels, 3), Conv2dValid(out_channels, out_channels, 7)]
        self.model_blocks = nn.Sequential(*ups, *head)
        self.activation = nn.Sigmoid() if activation == 'sigmoid' else nn.ReLU()   # declared, not applied (as the reference)

    def forward(self, x):
        return self.model_blocks(x)


class AE(nn.Module):
    def __init__(self, in_channels, out_channels, dims=64, n_downsample=4):
        super().__init__()
        self.encoder = Encoder(in_channels, dim=dims, n_downsample=n_downsample, encoder_type='regular')
        self.decoder = Decoder(out_channels, dim=dims, n_upsample=n_downsample, activation='relu')

    def forward(self, x):
        return self.decoder(self.encoder(x))
