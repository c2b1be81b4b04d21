"""MDAM — drop-in for the reference's model/utils/attention.py:147-190 (same constructor,
children and state_dict keys).

What the reference computes per group of channels (attention.py:171-190):
    w   = softmax(avgpool(GroupNorm(x * sigmoid(strip gates)))) + fc(avgpool(conv3x3(x)))
    out = x * sigmoid(w)
Two exact simplifications remove every activation-sized pass but two:
  * GroupNorm is built with one channel per group (attention.py:157), so each normalised map has
    mean zero and avgpool(GroupNorm(u)) is the GroupNorm bias for ANY u: the strip pooling, the
    1x1 conv and the gates never reach the output.  (In the reference the difference is float
    round-off of ~1e-7; `conv1x1` and `groupNorm.weight` receive round-off-sized gradients
    there and exact zeros here.)
  * avgpool(conv3x3(x)) needs no conv: it is bias + sum_ci,tap W[co,ci,tap] * S[ci,tap] / (H*W)
    with S the nine zero-padded window sums of x per channel (functional.WindowSumsFn).
So: one reduction pass over x, a few hundred scalars of torch arithmetic (autograd supplies the
parameter gradients), one scaling pass.
"""
import torch
import torch.nn as nn

from ... import functional as F_


class MDAM(nn.Module):
    def __init__(self, channels, factor=32):
        super().__init__()
        self.group = factor
        assert channels // self.group > 0
        cg = channels // self.group
        self.softmax = nn.Softmax(dim=1)
        self.averagePooling = nn.AdaptiveAvgPool2d((1, 1))
        self.Pool_h = nn.AdaptiveAvgPool2d((None, 1))
        self.Pool_w = nn.AdaptiveAvgPool2d((1, None))
        self.groupNorm = nn.GroupNorm(cg, cg)
        self.conv1x1 = nn.Conv2d(cg, cg, kernel_size=1, stride=1, padding=0)
        self.conv3x3 = nn.Conv2d(cg, cg, kernel_size=3, stride=1, padding=1)
        self.gap = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Sequential(nn.Conv2d(cg, cg, kernel_size=1, bias=False), nn.ReLU(),
                                nn.Conv2d(cg, cg, kernel_size=1, bias=False), nn.Sigmoid())

    def channel_weights(self, S, hw):
        """S: window sums [B, C, 9] -> sigmoid(w) [B, C] (float32, differentiable in S and the parameters)."""
        B, C, _ = S.shape
        cg = C // self.group
        if C != cg * self.group:
            raise ValueError(f"{C} channels cannot be split into {self.group} groups")
        s = S.reshape(B * self.group, cg * 9) / float(hw)
        y = s @ self.conv3x3.weight.reshape(cg, cg * 9).t() + self.conv3x3.bias          # avgpool(conv3x3(group_x))
        y = torch.sigmoid(torch.relu(y @ self.fc[0].weight.reshape(cg, cg).t()) @ self.fc[2].weight.reshape(cg, cg).t())
        x = torch.softmax(self.groupNorm.bias, dim=0)                                     # avgpool(GroupNorm(.)) = bias
        # parameters the output does not depend on: exact-zero gradients rather than None
        dead = (self.conv1x1.weight.sum() + self.conv1x1.bias.sum() + self.groupNorm.weight.sum()) * 0.0
        return torch.sigmoid(x[None, :] + y + dead).reshape(B, C)

    def forward(self, inputs):
        x = F_.to_act(inputs)
        B, C, H, W = x.shape
        S = F_.WindowSumsFn.apply(x)
        return F_.ChannelScaleFn.apply(x, self.channel_weights(S, H * W))
