"""DoubleConv — drop-in for the reference's model/utils/modules.py:78-116.

Same constructor, same children (`cbr1`, `cbr2` = Sequential(conv, bn, relu)) and therefore
the same state_dict keys; the forward runs the fused gfx950 kernels instead of ATen.
"""
import torch.nn as nn

from ... import functional as F_


def _init_like_reference(module):
    """What the reference intends (modules.py:101-116, utils/model_initial.py:8-21):
    kaiming-normal fan_out/relu conv weights, zero biases, BN gamma=1 beta=0."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


class DoubleConv(nn.Module):
    def __init__(self, in_channels, out_channels, mid_channels=None):
        super().__init__()
        if mid_channels is None:
            mid_channels = out_channels // 4
        self.relu = nn.ReLU(inplace=True)
        self.cbr1 = nn.Sequential(
            nn.Conv2d(in_channels, mid_channels, kernel_size=3, padding=1),
            nn.BatchNorm2d(mid_channels),
            self.relu)
        self.cbr2 = nn.Sequential(
            nn.Conv2d(mid_channels, out_channels, kernel_size=3, padding=1),
            nn.BatchNorm2d(out_channels),
            self.relu)
        self._initialize_weights()

    def forward_lazy(self, x, x_ss=None, x2=None, x2_ss=None):
        """(x, x_ss) [| (x2, x2_ss)] -> (z, ss): both convs read their input through the producer's
        BatchNorm affine + ReLU on load; the block's own output stays lazy (conv output + affine).
        With x2 the block's input is the channel concatenation [x | x2], never materialised."""
        z1, ss1 = F_.conv_bn_relu(x, x_ss, self.cbr1[0], self.cbr1[1], self.training, x2, x2_ss)
        return F_.conv_bn_relu(z1, ss1, self.cbr2[0], self.cbr2[1], self.training)

    def forward(self, x):
        """x: [B, C, H, W] tensor on the GPU (float32 or bfloat16) -> the block's activation."""
        z, ss = self.forward_lazy(F_.to_act(x))
        return F_.materialize(z, ss)

    def _initialize_weights(self, init_gain=0.02):
        _init_like_reference(self)
