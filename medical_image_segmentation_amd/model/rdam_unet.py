"""RDAM_UNet — drop-in for the reference's model/rdam_unet.py:7-115: plain 3x3 convs at full
resolution (encoder1 / decoder4), ResDConv blocks elsewhere, an AMSFN after every block, MDAM
attention on the skip connections, a DoubleConv as the output layer.  Same constructor, attribute
names (`down`, `up`, `encoder1-4`, `acpn0-3`, `att1-4`, `center_conv`, `decoder1-4`, `acpn5-8`,
dropouts, `out_conv`) and state_dict.

Data flow as in the reference's forward (rdam_unet.py:63-115): pooling takes the AMSFN outputs, the
skips are MDAM(block output), the decoder blocks read [up(previous) | skip] — as two tensors where
the conv kernels accept that, so the concatenation is never written.  AMSFN and DoubleConv outputs
stay lazy (conv output + BatchNorm affine); Dropout2d is applied by the pooling / upsampling kernel
that consumes the dropped tensor.
"""
import torch
import torch.nn as nn

from .. import functional as F_
from .utils.attention import MDAM
from .utils.modules import AMSFN, DoubleConv, ResDConv, prepare_conv_weights


class RDAM_UNet(nn.Module):
    def __init__(self, in_channels, n_classes, p, base_channels=32, compute_dtype=None):
        super().__init__()
        self.in_channels = in_channels
        self.n_classes = n_classes
        self.base_channels = base_channels
        self.compute_dtype = compute_dtype
        b = base_channels
        self.down = nn.MaxPool2d(kernel_size=2, stride=2)
        self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
        self.encoder1 = nn.Conv2d(in_channels, b, kernel_size=3, padding=1)
        self.encoder2 = ResDConv(b, b * 2)
        self.encoder3 = ResDConv(b * 2, b * 4)
        self.encoder4 = ResDConv(b * 4, b * 8)
        self.acpn0 = AMSFN(b)
        self.acpn1 = AMSFN(b * 2)
        self.acpn2 = AMSFN(b * 4)
        self.acpn3 = AMSFN(b * 8)
        self.att1 = MDAM(b)
        self.att2 = MDAM(b * 2)
        self.att3 = MDAM(b * 4)
        self.att4 = MDAM(b * 8)
        self.encoder_dropout1 = nn.Dropout2d(p=p * 0.3 if p != 0 else 0)
        self.encoder_dropout2 = nn.Dropout2d(p=p * 0.5 if p != 0 else 0)
        self.encoder_dropout3 = nn.Dropout2d(p=p * 0.7 if p != 0 else 0)
        self.encoder_dropout4 = nn.Dropout2d(p=p * 0.9 if p != 0 else 0)
        self.center_conv = DoubleConv(b * 8, b * 8, mid_channels=b * 16)
        self.bottleneck_dropout = nn.Dropout2d(p=p if p != 0.0 else 0.0)
        self.decoder1 = ResDConv(b * 16, b * 4)
        self.decoder2 = ResDConv(b * 8, b * 2)
        self.decoder3 = ResDConv(b * 4, b)
        self.decoder4 = nn.Conv2d(b * 2, b, kernel_size=3, padding=1)
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

    @staticmethod
    def _up_res(block, low, low_ss, skip, noise):
        """x = up(dropout(low)); x = cat([x, skip], 1); x = block(x) — rdam_unet.py:93-95."""
        if low.shape[1] == skip.shape[1] and block.two_inputs_supported(skip):
            return block.forward_plain(F_.UpsampleFn.apply(low, low_ss, noise), None, skip, None)
        return block.forward_plain(F_.UpsampleCatFn.apply(low, low_ss, skip, None, noise))

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError(f"expected [B, {self.in_channels}, H, W], got {tuple(x.shape)}")
        if x.shape[2] % 16 or x.shape[3] % 16:
            raise ValueError(f"H and W must be multiples of 16 (four 2x2 poolings), got {tuple(x.shape[2:])}")
        pool, noise = F_.MaxPool2x2Fn.apply, F_.dropout2d_noise
        x = F_.to_act(x, self._dtype())
        prepare_conv_weights(self, x.dtype)
        e1 = F_.Conv3x3Fn.apply(x, None, None, None, self.encoder1.weight, self.encoder1.bias)
        a = self.acpn0.forward_lazy(e1)
        m1 = self.att1(e1)
        e2 = self.encoder2.forward_plain(pool(*a, noise(self.encoder_dropout1, a[0])))
        a = self.acpn1.forward_lazy(e2)
        m2 = self.att2(e2)
        e3 = self.encoder3.forward_plain(pool(*a, noise(self.encoder_dropout2, a[0])))
        a = self.acpn2.forward_lazy(e3)
        m3 = self.att3(e3)
        e4 = self.encoder4.forward_plain(pool(*a, noise(self.encoder_dropout3, a[0])))
        a = self.acpn3.forward_lazy(e4)
        m4 = self.att4(e4)
        c5 = self.center_conv.forward_lazy(pool(*a, noise(self.encoder_dropout4, a[0])))
        d = self.acpn5.forward_lazy(self._up_res(self.decoder1, *c5, m4, noise(self.bottleneck_dropout, c5[0])))
        d = self.acpn6.forward_lazy(self._up_res(self.decoder2, *d, m3, noise(self.decoder_dropout1, d[0])))
        d = self.acpn7.forward_lazy(self._up_res(self.decoder3, *d, m2, noise(self.decoder_dropout2, d[0])))
        # decoder4 is a bare conv over [up(d) | m1]
        wd = self.decoder4.weight
        if d[0].shape[1] == m1.shape[1] and F_.cat_supported(m1, m1, wd.shape[0]):
            d1 = F_.Conv3x3Fn.apply(F_.UpsampleFn.apply(*d, None), None, m1, None, wd, self.decoder4.bias)
        else:
            d1 = F_.Conv3x3Fn.apply(F_.UpsampleCatFn.apply(*d, m1, None, None), None, None, None, wd, self.decoder4.bias)
        a = self.acpn8.forward_lazy(d1)
        return F_.materialize(*self.out_conv.forward_lazy(*a))
