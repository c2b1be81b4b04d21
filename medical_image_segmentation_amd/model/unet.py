"""UNet — drop-in for the reference's model/unet.py:10-94.

Same constructor signature, attribute names (`dowm`, `up`, `encoder1-4`, `center_conv`,
`decoder1-4`, `*_dropout*`, `out_conv`) and state_dict; `forward(x)` takes the same
[B, in_channels, H, W] tensor and returns [B, n_classes, H, W] logits (stored
channels-last).  Every layer runs hand-written gfx950 kernels.

compute dtype: bfloat16 inside an autocast region (the reference trains under
`autocast("cuda")`, train_and_eval.py:71), float32 outside one — or whatever
`compute_dtype` pins.
"""
import torch
import torch.nn as nn

from .. import functional as F_
from .utils.modules import DoubleConv


class UNet(nn.Module):
    def __init__(self, in_channels, n_classes, p, base_channels=32, compute_dtype=None):
        super().__init__()
        self.in_channels = in_channels
        self.n_classes = n_classes
        self.compute_dtype = compute_dtype
        self.dowm = nn.MaxPool2d(kernel_size=2, stride=2)
        self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
        self.encoder1 = DoubleConv(in_channels, base_channels)
        self.encoder2 = DoubleConv(base_channels, base_channels * 2)
        self.encoder3 = DoubleConv(base_channels * 2, base_channels * 4)
        self.encoder4 = DoubleConv(base_channels * 4, base_channels * 8)
        self.encoder_dropout1 = nn.Dropout2d(p=p * 0.3 if p != 0 else 0)
        self.encoder_dropout2 = nn.Dropout2d(p=p * 0.5 if p != 0 else 0)
        self.encoder_dropout3 = nn.Dropout2d(p=p * 0.7 if p != 0 else 0)
        self.encoder_dropout4 = nn.Dropout2d(p=p * 0.9 if p != 0 else 0)
        self.pool = nn.MaxPool2d(kernel_size=2, stride=2)
        self.center_conv = DoubleConv(base_channels * 8, base_channels * 8, mid_channels=base_channels * 16)
        self.bottleneck_dropout = nn.Dropout2d(p=p if p != 0.0 else 0.0)
        self.decoder1 = DoubleConv(base_channels * 16, base_channels * 4)
        self.decoder2 = DoubleConv(base_channels * 8, base_channels * 2)
        self.decoder3 = DoubleConv(base_channels * 4, base_channels)
        self.decoder4 = DoubleConv(base_channels * 2, base_channels)
        self.decoder_dropout1 = nn.Dropout2d(p=p * 0.3 if p != 0 else 0)
        self.decoder_dropout2 = nn.Dropout2d(p=p * 0.2 if p != 0 else 0)
        self.out_conv = nn.Conv2d(base_channels, n_classes, kernel_size=1)

    def _dtype(self):
        if self.compute_dtype is not None:
            return self.compute_dtype
        return torch.bfloat16 if torch.is_autocast_enabled() else torch.float32

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError(f"expected [B, {self.in_channels}, H, W], got {tuple(x.shape)}")
        if x.shape[2] % 16 or x.shape[3] % 16:
            raise ValueError(f"H and W must be multiples of 16 (four 2x2 poolings), got {tuple(x.shape[2:])}")
        down, noise = F_.PoolSkipFn.apply, F_.dropout2d_noise
        x = F_.to_act(x, self._dtype())
        self._pack_weights(x.dtype)
        # every DoubleConv output stays lazy (z, ss): pool, upsample+concat, the next conv and the
        # 1x1 head apply its BatchNorm affine + ReLU while loading it.  `down` returns the pooled
        # tensor and the skip alias of its input, so both gradients meet in one kernel.
        # Dropout2d (identity at p == 0 / eval: no noise tensor, no extra work) is one multiplier per
        # (sample, channel): the kernel that consumes the dropped tensor's source applies it — the
        # pooling for the encoder dropouts, the upsampling for the center / decoder ones.  The
        # noise tensors are drawn in the reference's order (unet.py:49-74).
        z1, s1 = self.encoder1.forward_lazy(x)
        p1, z1 = down(z1, s1, noise(self.encoder_dropout1, z1))
        z2, s2 = self.encoder2.forward_lazy(p1)
        p2, z2 = down(z2, s2, noise(self.encoder_dropout2, z2))
        z3, s3 = self.encoder3.forward_lazy(p2)
        p3, z3 = down(z3, s3, noise(self.encoder_dropout3, z3))
        z4, s4 = self.encoder4.forward_lazy(p3)
        p4, z4 = down(z4, s4, noise(self.encoder_dropout4, z4))
        y, sy = self.center_conv.forward_lazy(p4)
        y, sy = self._up(self.decoder1, y, sy, z4, s4, noise(self.bottleneck_dropout, y))
        y, sy = self._up(self.decoder2, y, sy, z3, s3, noise(self.decoder_dropout1, y))
        y, sy = self._up(self.decoder3, y, sy, z2, s2, noise(self.decoder_dropout2, y))
        # decoder4's output has one consumer, the 1x1 head: its gradient is formed inside decoder4's
        # BatchNorm backward from the head's output gradient and never stored (functional.HeadFusion)
        head = F_.HeadFusion() if (self.training and torch.is_grad_enabled()) else None
        y, sy = self._up(self.decoder4, y, sy, z1, s1, None, head)
        return F_.Conv1x1Fn.apply(y, sy, self.out_conv.weight, self.out_conv.bias, head)

    def _pack_weights(self, dtype):
        """All 3x3 conv weights -> the kernels' operand images in one launch (only when a weight
        changed since the last call)."""
        convs = self.__dict__.get("_mis_convs")
        if convs is None:         # the nn.Conv2d modules (their .weight may be re-assigned, the modules stay)
            convs = [blk[0] for m in self.modules() if isinstance(m, DoubleConv) for blk in (m.cbr1, m.cbr2)]
            self.__dict__["_mis_convs"] = convs
        weights = [c.weight for c in convs]
        packer = self.__dict__.get("_mis_packer")
        if packer is None or not packer.valid_for(weights, dtype):
            packer = F_.WeightPacker(weights, dtype)
            self.__dict__["_mis_packer"] = packer
        packer.refresh()

    @staticmethod
    def _up(block, low, low_ss, skip, skip_ss, low_noise, head=None):
        """x = up(dropout(low)); x = cat([x, skip], 1); x = block(x) — unet.py:64-68.  Where the
        conv kernels take two input tensors the concatenation is virtual: only the upsampled half
        is written, the skip is read in place (lazily) by the conv and its weight gradient, and
        the data gradient comes back as two tensors."""
        if F_.cat_supported(skip, skip, block.cbr1[0].out_channels) and low.shape[1] == skip.shape[1]:
            return block.forward_lazy(F_.UpsampleFn.apply(low, low_ss, low_noise), None, skip, skip_ss, head)
        return block.forward_lazy(F_.UpsampleCatFn.apply(low, low_ss, skip, skip_ss, low_noise), head=head)

    def elastic_net(self, l1_lambda, l2_lambda):
        l1_loss = 0
        l2_loss = 0
        for param in self.parameters():
            l1_loss += torch.abs(param).sum()
            l2_loss += torch.pow(param, 2).sum()
        return l1_lambda * l1_loss + l2_lambda * l2_loss
