"""A_UNet — drop-in for the reference's model/a_unet.py:8-105: the U-Net with an AMSFN fusion
block after every DoubleConv and a DoubleConv as the output layer.  Same constructor, attribute
names (`down`, `up`, `encoder1-4`, `acpn0-3`, `center_conv`, `decoder1-4`, `acpn5-8`, dropouts,
`out_conv`) and state_dict; every activation-sized operation runs in the gfx950 kernels.

Skip connections take the DoubleConv outputs (e1..e4), pooling and upsampling take the AMSFN
outputs (a_unet.py:60-103).  All block outputs stay lazy (conv output + BatchNorm affine) and
the decoder concatenations are virtual where the conv kernels accept two inputs, as in UNet.
"""
import torch
import torch.nn as nn

from .. import functional as F_
from .unet import UNet
from .utils.modules import AMSFN, DoubleConv, prepare_conv_weights


class A_UNet(nn.Module):
    def __init__(self, in_channels, n_classes, p, base_channels=32, compute_dtype=None):
        super().__init__()
        self.in_channels = in_channels
        self.n_classes = n_classes
        self.base_channels = base_channels
        self.compute_dtype = compute_dtype
        b = base_channels
        self.down = nn.MaxPool2d(kernel_size=2, stride=2)
        self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
        self.encoder1 = DoubleConv(in_channels, b)
        self.encoder2 = DoubleConv(b, b * 2)
        self.encoder3 = DoubleConv(b * 2, b * 4)
        self.encoder4 = DoubleConv(b * 4, b * 8)
        self.acpn0 = AMSFN(b)
        self.acpn1 = AMSFN(b * 2)
        self.acpn2 = AMSFN(b * 4)
        self.acpn3 = AMSFN(b * 8)
        self.encoder_dropout1 = nn.Dropout2d(p=p * 0.3 if p != 0 else 0)
        self.encoder_dropout2 = nn.Dropout2d(p=p * 0.5 if p != 0 else 0)
        self.encoder_dropout3 = nn.Dropout2d(p=p * 0.7 if p != 0 else 0)
        self.encoder_dropout4 = nn.Dropout2d(p=p * 0.9 if p != 0 else 0)
        self.center_conv = DoubleConv(b * 8, b * 8, mid_channels=b * 16)
        self.bottleneck_dropout = nn.Dropout2d(p=p if p != 0.0 else 0.0)
        self.decoder1 = DoubleConv(b * 16, b * 4)
        self.decoder2 = DoubleConv(b * 8, b * 2)
        self.decoder3 = DoubleConv(b * 4, b)
        self.decoder4 = DoubleConv(b * 2, b)
        self.acpn5 = AMSFN(b * 4)
        self.acpn6 = AMSFN(b * 2)
        self.acpn7 = AMSFN(b)
        self.acpn8 = AMSFN(b)
        self.decoder_dropout1 = nn.Dropout2d(p=p * 0.3 if p != 0 else 0)
        self.decoder_dropout2 = nn.Dropout2d(p=p * 0.2 if p != 0 else 0)
        self.out_conv = DoubleConv(b, n_classes)

    def _dtype(self):
        if self.compute_dtype is not None:
            return self.compute_dtype
        return torch.bfloat16 if torch.is_autocast_enabled() else torch.float32

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError(f"expected [B, {self.in_channels}, H, W], got {tuple(x.shape)}")
        if x.shape[2] % 16 or x.shape[3] % 16:
            raise ValueError(f"H and W must be multiples of 16 (four 2x2 poolings), got {tuple(x.shape[2:])}")
        pool, noise, up = F_.MaxPool2x2Fn.apply, F_.dropout2d_noise, UNet._up
        x = F_.to_act(x, self._dtype())
        prepare_conv_weights(self, x.dtype)
        # an encoder output is read by the four branches of its AMSFN block and by the decoder's skip connection:
        # five aliases whose gradients meet in one kernel (which also emits the encoder layer's BatchNorm-backward sums)
        def block(acpn, e):
            *four, skip = F_.fan_out(e[0], e[1], 5)
            return acpn.forward_lazy(None, e[1], aliases=four), (skip, e[1])

        a, s1 = block(self.acpn0, self.encoder1.forward_lazy(x))
        a, s2 = block(self.acpn1, self.encoder2.forward_lazy(pool(*a, noise(self.encoder_dropout1, a[0]))))
        a, s3 = block(self.acpn2, self.encoder3.forward_lazy(pool(*a, noise(self.encoder_dropout2, a[0]))))
        a, s4 = block(self.acpn3, self.encoder4.forward_lazy(pool(*a, noise(self.encoder_dropout3, a[0]))))
        c5 = self.center_conv.forward_lazy(pool(*a, noise(self.encoder_dropout4, a[0])))
        d = self.acpn5.forward_lazy(*up(self.decoder1, *c5, *s4, noise(self.bottleneck_dropout, c5[0])))
        d = self.acpn6.forward_lazy(*up(self.decoder2, *d, *s3, noise(self.decoder_dropout1, d[0])))
        d = self.acpn7.forward_lazy(*up(self.decoder3, *d, *s2, noise(self.decoder_dropout2, d[0])))
        d = self.acpn8.forward_lazy(*up(self.decoder4, *d, *s1, None))
        return F_.materialize(*self.out_conv.forward_lazy(*d))
