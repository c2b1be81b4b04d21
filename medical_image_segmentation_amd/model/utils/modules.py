"""DoubleConv, DWConv, AMSFN and ResDConv — drop-ins for the reference's model/utils/modules.py:78-116,
52-77, 791-887 and 426-471.

Same constructors, same children (`cbr1`, `cbr2` = Sequential(conv, bn, relu), ...) and therefore
the same state_dict keys; the forwards run the fused gfx950 kernels instead of ATen.
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

    def forward_lazy(self, x, x_ss=None, x2=None, x2_ss=None, head=None):
        """(x, x_ss) [| (x2, x2_ss)] -> (z, ss): both convs read their input through the producer's
        BatchNorm affine + ReLU on load; the block's own output stays lazy (conv output + affine).
        With x2 the block's input is the channel concatenation [x | x2], never materialised.
        head: a functional.HeadFusion, when a 1x1 conv given the same object is the only consumer of (z, ss)."""
        z1, ss1 = F_.conv_bn_relu(x, x_ss, self.cbr1[0], self.cbr1[1], self.training, x2, x2_ss)
        return F_.conv_bn_relu(z1, ss1, self.cbr2[0], self.cbr2[1], self.training, head=head)

    def forward(self, x):
        """x: [B, C, H, W] tensor on the GPU (float32 or bfloat16) -> the block's activation."""
        z, ss = self.forward_lazy(F_.to_act(x))
        return F_.materialize(z, ss)

    def _initialize_weights(self, init_gain=0.02):
        _init_like_reference(self)


class DWConv(nn.Module):
    """Depthwise kxk conv followed by a pointwise 1x1 conv (modules.py:52-60).  Holds the same two
    nn.Conv2d children (`dconv`, `pconv`); the blocks that use it run the pair as one dense conv
    (see functional.dwconv_as_3x3)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1):
        super().__init__()
        self.dconv = nn.Conv2d(in_channels, in_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                               groups=in_channels, dilation=dilation)
        self.pconv = nn.Conv2d(in_channels, out_channels, kernel_size=1)
        _init_like_reference(self)

    def composable(self):
        d = self.dconv
        return (d.kernel_size == (3, 3) and d.stride == (1, 1) and d.padding == (1, 1) and d.dilation == (1, 1))


class AMSFN(nn.Module):
    """Adaptive multi-scale fusion block (modules.py:791-871): four branches over the same input —
    max-pool 3x3 + 1x1 conv, and one, two, three chained DWConv — each ending in BN + ReLU; their
    global averages go through a squeeze MLP + sigmoid to one weight per (sample, branch, channel);
    weighted sum; 1x1 conv + BN + ReLU."""

    def __init__(self, in_channels, out_channels=None, mid_channels=None):
        super().__init__()
        if out_channels is None:
            out_channels = in_channels
        if mid_channels is None:
            mid_channels = in_channels // 2
        C, M = in_channels, mid_channels

        def dw(a, b):
            return [DWConv(a, b, kernel_size=3, padding=1), nn.BatchNorm2d(b), nn.ReLU(inplace=True)]

        self.conv1x1 = nn.Sequential(nn.MaxPool2d(kernel_size=3, stride=1, padding=1),
                                     nn.Conv2d(C, C, kernel_size=1, padding=0), nn.BatchNorm2d(C),
                                     nn.ReLU(inplace=True))
        self.cbr1 = nn.Sequential(*dw(C, C))
        self.cbr2 = nn.Sequential(*dw(C, M), *dw(M, C))
        self.cbr3 = nn.Sequential(*dw(C, M), *dw(M, M), *dw(M, C))
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.sgm = nn.Sigmoid()
        self.mlp = nn.Sequential(nn.Conv2d(C * 4, C, kernel_size=1), nn.ReLU(inplace=True),
                                 nn.Conv2d(C, C * 4, kernel_size=1))
        self.final_conv = nn.Sequential(nn.Conv2d(C, out_channels, kernel_size=1, padding=0),
                                        nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True))
        _init_like_reference(self)

    def _dw_chain(self, seq, x, x_ss, link=None):
        for i in range(0, len(seq), 3):
            W, b = F_.dwconv_as_3x3(seq[i].dconv, seq[i].pconv)
            x, x_ss = F_.conv_bn_relu_w(x, x_ss, W, b, seq[i + 1], self.training,
                                        link if i + 3 >= len(seq) else None)
        return x, x_ss

    def forward_lazy(self, x, x_ss=None, input_read_elsewhere=False, aliases=None):
        """(x, x_ss) lazy or plain activation -> lazy (z, ss) of the final conv.  The input feeds four branches:
        their gradients are summed in one kernel (functional.fan_out), which also produces the BatchNorm-backward
        sums of the layer that wrote x unless the caller says that x has further readers.
        aliases: four aliases of x from a fan-out the caller made itself (one that also serves x's other readers,
        so all gradients of x meet in one kernel); x is then not used."""
        x1, x2, x3, x4 = aliases if aliases is not None else F_.fan_out(x, x_ss, 4, bn_sums=not input_read_elsewhere)
        # each branch output is read by the fusion node alone: its gradient is never written (functional.BranchGradLink)
        links = [F_.BranchGradLink() for _ in range(4)] if self.training else [None] * 4
        pooled = F_.MaxPool3x3Fn.apply(x1, x_ss)
        c1 = F_.conv_bn_relu_w(pooled, None, F_.pointwise_as_3x3(self.conv1x1[1].weight), self.conv1x1[1].bias,
                               self.conv1x1[2], self.training, links[0], pointwise=True)
        c2 = self._dw_chain(self.cbr1, x2, x_ss, links[1])
        c3 = self._dw_chain(self.cbr2, x3, x_ss, links[2])
        c4 = self._dw_chain(self.cbr3, x4, x_ss, links[3])
        fused = F_.AmsfnFuseFn.apply(*c1, *c2, *c3, *c4, self.mlp[0].weight, self.mlp[0].bias, self.mlp[2].weight,
                                     self.mlp[2].bias, links if self.training else None)
        return F_.conv_bn_relu_w(fused, None, F_.pointwise_as_3x3(self.final_conv[0].weight),
                                 self.final_conv[0].bias, self.final_conv[1], self.training, pointwise=True)

    def forward(self, x):
        z, ss = self.forward_lazy(F_.to_act(x))
        return F_.materialize(z, ss)

    def _initialize_weights(self, init_gain=0.02):
        _init_like_reference(self)


def prepare_conv_weights(model, dtype):
    """Top of a model's forward: every DWConv composition, every 1x1-as-3x3 weight and every operand image of
    the model's 3x3 convs in two launches (functional.ComposedWeights), instead of one or two small launches per
    layer and step.  Bare nn.Conv2d children of the model itself (RDAM_UNet.encoder1 / decoder4) count as 3x3 convs."""
    def mods(kind):
        return [m for m in model.modules() if isinstance(m, kind)]

    def leaves():
        out = [blk[0] for m in mods(DoubleConv) for blk in (m.cbr1, m.cbr2)]
        out += [blk[0] for m in mods(ResDConv) for blk in (m.cbr1, m.cbr2, m.cbr3)]
        out += [c for c in model.children() if isinstance(c, nn.Conv2d) and c.kernel_size == (3, 3)]
        return out

    F_.prepare_weights(model, dtype,
                       dw=lambda: [(m.dconv, m.pconv) for m in mods(DWConv)],
                       pw=lambda: [c for m in mods(AMSFN) for c in (m.conv1x1[1], m.final_conv[0])]
                                  + [m.c1[0] for m in mods(ResDConv)],
                       leaves=leaves)


class ResDConv(nn.Module):
    """Residual block of RDAM_UNet (modules.py:426-453): three chained 3x3 convs with dilations 1, 2, 3
    (each + BN, the first two + ReLU) plus a 1x1 conv + BN shortcut; out = relu(main + shortcut).
    Same children (`c1`, `relu`, `cbr1`, `cbr2`, `cbr3`) and state_dict keys.  The dilated convs run
    on the matrix-core kernels with dilated taps, the 1x1 shortcut as a centre-tap 3x3; the two
    BatchNorm applications, the addition and the ReLU are one pass."""

    def __init__(self, in_channels, out_channels, mid_channels=None):
        super().__init__()
        if mid_channels is None:
            mid_channels = in_channels // 2
        self.c1 = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size=1), nn.BatchNorm2d(out_channels))
        self.relu = nn.ReLU(inplace=True)
        self.cbr1 = nn.Sequential(nn.Conv2d(in_channels, mid_channels, kernel_size=3, padding=1),
                                  nn.BatchNorm2d(mid_channels), nn.ReLU(inplace=True))
        self.cbr2 = nn.Sequential(nn.Conv2d(mid_channels, mid_channels, kernel_size=3, padding=2, dilation=2),
                                  nn.BatchNorm2d(mid_channels), nn.ReLU(inplace=True))
        self.cbr3 = nn.Sequential(nn.Conv2d(mid_channels, out_channels, kernel_size=3, padding=3, dilation=3),
                                  nn.BatchNorm2d(out_channels))
        _init_like_reference(self)

    def two_inputs_supported(self, skip):
        """Whether both convs that read the block's input take it as two tensors [up | skip]."""
        return (F_.cat_supported(skip, skip, self.c1[0].out_channels)
                and F_.cat_supported(skip, skip, self.cbr1[0].out_channels))

    def forward_plain(self, x, x_ss=None, x2=None, x2_ss=None):
        """(x, x_ss) [| (x2, x2_ss)] lazy or plain -> the block's activation (a plain tensor)."""
        t = self.training
        re = F_.conv_bn(x, x_ss, F_.pointwise_as_3x3(self.c1[0].weight), self.c1[0].bias, self.c1[1], t, relu=False,
                        dilation=0, x2=x2, x2_ss=x2_ss)           # (0: centre tap only, where the kernels take it)
        y = F_.conv_bn(x, x_ss, self.cbr1[0].weight, self.cbr1[0].bias, self.cbr1[1], t, x2=x2, x2_ss=x2_ss)
        y = F_.conv_bn(*y, self.cbr2[0].weight, self.cbr2[0].bias, self.cbr2[1], t, dilation=2)
        y = F_.conv_bn(*y, self.cbr3[0].weight, self.cbr3[0].bias, self.cbr3[1], t, relu=False, dilation=3)
        return F_.add_bn_relu(y, re)

    def forward(self, x):
        return self.forward_plain(F_.to_act(x))

    def _initialize_weights(self, init_gain=0.02):
        _init_like_reference(self)
