"""TEST INFRASTRUCTURE — CPU restatement of the reference's U-Net training hot path.

This file is the *checker*: only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import it.  The shipped package never does.

Every function restates one piece of the reference (paths relative to the
reference checkout) with explicit formulas on torch-CPU / numpy tensors:

  conv_bn_relu / double_conv   model/utils/modules.py:78-99
  unet_forward                 model/unet.py:46-85
  maxpool2x2                   model/unet.py:19   (nn.MaxPool2d(2, 2))
  upsample2x_bilinear          model/unet.py:20   (nn.Upsample(2,'bilinear',align_corners=True))
  dice_loss                    utils/loss_fn.py:63-88
  ce_map                       utils/loss_fn.py:199-203
  confusion_counts / metrics_update   utils/metrics.py:24-152
  train_step                   utils/train_and_eval.py:63-121 (one batch)
  slide_*                      utils/slide_predict.py:15-117
  dropout2d_noise_philox       model/unet.py:49-74 (nn.Dropout2d) with a counter-based generator

Parity pinning: the reference ships no tests or golden vectors, so this
restatement is pinned by outputs of the reference itself, generated in the build
container by tests/golden/make_golden.py and committed under tests/golden/
(checked in tests/test_oracle_golden.py).
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5          # nn.BatchNorm2d default, modules.py:88,93
BN_MOMENTUM = 0.1      # nn.BatchNorm2d default


# ----------------------------------------------------------------------------
# model pieces
# ----------------------------------------------------------------------------
# The restatements below spell BatchNorm and the bilinear upsampling out step by step (they are
# what the parity tests read).  The reference itself calls ATen's fused kernels for both
# (nn.BatchNorm2d -> F.batch_norm, nn.Upsample -> F.interpolate); with ATEN_FUSED set the same
# calls are used here — same results to rounding (tests/test_oracle_golden.py), the reference's
# own CPU cost.  Only the CPU-baseline timing sets it.
ATEN_FUSED = False


def conv3x3(x, w, b):
    """nn.Conv2d(kernel_size=3, padding=1) — modules.py:87,92."""
    return F.conv2d(x, w, b, stride=1, padding=1)


def batchnorm_train(z, gamma, beta, eps=BN_EPS):
    """Training-mode BatchNorm2d: batch mean / biased variance over (N,H,W).
    Returns (y, mean, biased_var)."""
    mean = z.mean(dim=(0, 2, 3))
    var = z.var(dim=(0, 2, 3), unbiased=False)
    rstd = torch.rsqrt(var + eps)
    y = (z - mean[None, :, None, None]) * (rstd * gamma)[None, :, None, None] + beta[None, :, None, None]
    return y, mean, var


def batchnorm_eval(z, gamma, beta, running_mean, running_var, eps=BN_EPS):
    rstd = torch.rsqrt(running_var + eps)
    return (z - running_mean[None, :, None, None]) * (rstd * gamma)[None, :, None, None] + beta[None, :, None, None]


def update_running_stats(state, prefix, mean, var, count, momentum=BN_MOMENTUM):
    """running = (1-m)*running + m*batch; running_var uses the UNBIASED batch variance."""
    with torch.no_grad():
        unbiased = var * (count / max(count - 1, 1))
        state[prefix + ".running_mean"].mul_(1 - momentum).add_(momentum * mean.detach())
        state[prefix + ".running_var"].mul_(1 - momentum).add_(momentum * unbiased.detach())
        state[prefix + ".num_batches_tracked"] += 1


def conv_bn_relu(x, state, prefix, training=True, update_stats=True):
    """One `cbr` Sequential(conv, bn, relu) — modules.py:86-94.  `prefix` is e.g.
    'encoder1.cbr1'; keys follow the reference state_dict ('.0.weight', '.1.running_mean', ...)."""
    z = conv3x3(x, state[prefix + ".0.weight"], state[prefix + ".0.bias"])
    g, b = state[prefix + ".1.weight"], state[prefix + ".1.bias"]
    if ATEN_FUSED and training and update_stats:
        state[prefix + ".1.num_batches_tracked"] += 1
        return torch.relu(F.batch_norm(z, state[prefix + ".1.running_mean"], state[prefix + ".1.running_var"], g, b,
                                       True, BN_MOMENTUM, BN_EPS))
    if training:
        y, mean, var = batchnorm_train(z, g, b)
        if update_stats:
            count = z.shape[0] * z.shape[2] * z.shape[3]
            update_running_stats(state, prefix + ".1", mean, var, count)
    else:
        y = batchnorm_eval(z, g, b, state[prefix + ".1.running_mean"], state[prefix + ".1.running_var"])
    return torch.relu(y)


def double_conv(x, state, prefix, training=True, update_stats=True):
    """DoubleConv.forward — modules.py:97-99."""
    x = conv_bn_relu(x, state, prefix + ".cbr1", training, update_stats)
    return conv_bn_relu(x, state, prefix + ".cbr2", training, update_stats)


def maxpool2x2(x):
    """nn.MaxPool2d(2,2): window max; backward routes to the first maximum in
    row-major window order (torch CPU semantics, strict '>' scan)."""
    return F.max_pool2d(x, kernel_size=2, stride=2)


def upsample2x_bilinear(x):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True), restated:
    src = dst*(in-1)/(out-1); i0=floor(src); i1=min(i0+1,in-1); l1=src-i0; l0=1-l1;
    out = h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11)."""
    if ATEN_FUSED:
        return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    B, C, H, W = x.shape
    Ho, Wo = 2 * H, 2 * W

    def taps(n_in, n_out):
        scale = np.float32((n_in - 1) / (n_out - 1)) if n_out > 1 else np.float32(0.0)
        src = (np.arange(n_out, dtype=np.float32) * scale).astype(np.float32)
        i0 = np.minimum(src.astype(np.int64), n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        l1 = (src - i0.astype(np.float32)).astype(np.float32)
        l0 = (np.float32(1.0) - l1).astype(np.float32)
        return (torch.from_numpy(i0), torch.from_numpy(i1),
                torch.from_numpy(l0).to(x.dtype), torch.from_numpy(l1).to(x.dtype))

    y0, y1, hy0, hy1 = taps(H, Ho)
    x0, x1, wx0, wx1 = taps(W, Wo)
    r0 = x.index_select(2, y0)
    r1 = x.index_select(2, y1)
    top = r0.index_select(3, x0) * wx0 + r0.index_select(3, x1) * wx1
    bot = r1.index_select(3, x0) * wx0 + r1.index_select(3, x1) * wx1
    return top * hy0[:, None] + bot * hy1[:, None]


UNET_BLOCKS = ("encoder1", "encoder2", "encoder3", "encoder4", "center_conv",
               "decoder1", "decoder2", "decoder3", "decoder4")


def unet_channels(in_channels, base=32):
    """(Cin, Cmid, Cout) per DoubleConv — unet.py:22-39 with modules.py:81-82."""
    b = base
    return OrderedDict([
        ("encoder1", (in_channels, b // 4, b)),
        ("encoder2", (b, b * 2 // 4, b * 2)),
        ("encoder3", (b * 2, b * 4 // 4, b * 4)),
        ("encoder4", (b * 4, b * 8 // 4, b * 8)),
        ("center_conv", (b * 8, b * 16, b * 8)),
        ("decoder1", (b * 16, b * 4 // 4, b * 4)),
        ("decoder2", (b * 8, b * 2 // 4, b * 2)),
        ("decoder3", (b * 4, b // 4, b)),
        ("decoder4", (b * 2, b // 4, b)),
    ])


def make_state(in_channels, n_classes, base=32, seed=1, dtype=torch.float32):
    """Deterministic (hash-generated) parameters with the reference's 128 state_dict
    keys.  Conv weights are kaiming-like scaled (fan_out / relu, model_initial.py:11);
    biases, BN gamma/beta get small non-trivial values so every gradient is exercised."""
    from .datagen import hash_normalish, hash_uniform
    st = OrderedDict()
    k = 0
    for name, (ci, cm, co) in unet_channels(in_channels, base).items():
        for j, (a, b_) in enumerate(((ci, cm), (cm, co)), start=1):
            p = f"{name}.cbr{j}"
            std = math.sqrt(2.0 / (b_ * 9))
            st[p + ".0.weight"] = torch.from_numpy(hash_normalish((b_, a, 3, 3), seed * 1000 + k, std)).to(dtype); k += 1
            st[p + ".0.bias"] = torch.from_numpy(hash_uniform((b_,), seed * 1000 + k, -0.1, 0.1)).to(dtype); k += 1
            st[p + ".1.weight"] = torch.from_numpy(hash_uniform((b_,), seed * 1000 + k, 0.5, 1.5)).to(dtype); k += 1
            st[p + ".1.bias"] = torch.from_numpy(hash_uniform((b_,), seed * 1000 + k, -0.2, 0.2)).to(dtype); k += 1
            st[p + ".1.running_mean"] = torch.zeros(b_, dtype=dtype)
            st[p + ".1.running_var"] = torch.ones(b_, dtype=dtype)
            st[p + ".1.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    std = math.sqrt(2.0 / n_classes)
    st["out_conv.weight"] = torch.from_numpy(hash_normalish((n_classes, base, 1, 1), seed * 1000 + k, std)).to(dtype); k += 1
    st["out_conv.bias"] = torch.from_numpy(hash_uniform((n_classes,), seed * 1000 + k, -0.1, 0.1)).to(dtype)
    return st


def hash_state(spec, seed, dtype=torch.float32):
    """Deterministic state_dict for any module: `spec` = [(key, shape), ...] in state_dict order.
    Conv weights (4-D) are kaiming-like scaled (fan_out), 1-D '.weight' (BatchNorm gamma) in
    [0.5, 1.5], biases in [-0.2, 0.2], running_mean 0, running_var 1, num_batches_tracked 0."""
    from .datagen import hash_normalish, hash_uniform
    st = OrderedDict()
    for k, (key, shape) in enumerate(spec):
        shape = tuple(int(v) for v in shape)
        sd = seed * 1000 + k
        if key.endswith("num_batches_tracked"):
            st[key] = torch.zeros((), dtype=torch.int64)
        elif key.endswith("running_mean"):
            st[key] = torch.zeros(shape, dtype=dtype)
        elif key.endswith("running_var"):
            st[key] = torch.ones(shape, dtype=dtype)
        elif len(shape) == 4:
            std = math.sqrt(2.0 / (shape[0] * shape[2] * shape[3]))
            st[key] = torch.from_numpy(hash_normalish(shape, sd, std)).to(dtype)
        elif key.endswith(".weight"):
            st[key] = torch.from_numpy(hash_uniform(shape, sd, 0.5, 1.5)).to(dtype)
        else:
            st[key] = torch.from_numpy(hash_uniform(shape, sd, -0.2, 0.2)).to(dtype)
    return st


def _cbr(z, state, bn_prefix, training, update_stats):
    """BatchNorm2d + ReLU with the parameters under `bn_prefix`."""
    g, b = state[bn_prefix + ".weight"], state[bn_prefix + ".bias"]
    if training:
        y, mean, var = batchnorm_train(z, g, b)
        if update_stats:
            update_running_stats(state, bn_prefix, mean, var, z.shape[0] * z.shape[2] * z.shape[3])
    else:
        y = batchnorm_eval(z, g, b, state[bn_prefix + ".running_mean"], state[bn_prefix + ".running_var"])
    return torch.relu(y)


def dwconv(x, state, prefix):
    """DWConv.forward — modules.py:52-60: depthwise 3x3 (padding 1, bias) then pointwise 1x1 (bias)."""
    d = F.conv2d(x, state[prefix + ".dconv.weight"], state[prefix + ".dconv.bias"], padding=1, groups=x.shape[1])
    return F.conv2d(d, state[prefix + ".pconv.weight"], state[prefix + ".pconv.bias"])


def amsfn(x, state, prefix, training=True, update_stats=True):
    """AMSFN.forward — modules.py:844-870: four branches (max-pool 3x3 + 1x1 conv; 1, 2 and 3
    chained DWConv-BN-ReLU), global average pools -> MLP -> sigmoid gives one weight per
    (sample, branch, channel), weighted sum, final 1x1 conv-BN-ReLU."""
    cbr = lambda z, p: _cbr(z, state, prefix + p, training, update_stats)
    C = x.shape[1]
    c1 = cbr(F.conv2d(F.max_pool2d(x, 3, 1, 1), state[prefix + ".conv1x1.1.weight"],
                      state[prefix + ".conv1x1.1.bias"]), ".conv1x1.2")
    c2 = cbr(dwconv(x, state, prefix + ".cbr1.0"), ".cbr1.1")
    c3 = cbr(dwconv(x, state, prefix + ".cbr2.0"), ".cbr2.1")
    c3 = cbr(dwconv(c3, state, prefix + ".cbr2.3"), ".cbr2.4")
    c4 = cbr(dwconv(x, state, prefix + ".cbr3.0"), ".cbr3.1")
    c4 = cbr(dwconv(c4, state, prefix + ".cbr3.3"), ".cbr3.4")
    c4 = cbr(dwconv(c4, state, prefix + ".cbr3.6"), ".cbr3.7")
    s = torch.cat([c.mean(dim=(2, 3), keepdim=True) for c in (c1, c2, c3, c4)], dim=1)
    h = torch.relu(F.conv2d(s, state[prefix + ".mlp.0.weight"], state[prefix + ".mlp.0.bias"]))
    w = torch.sigmoid(F.conv2d(h, state[prefix + ".mlp.2.weight"], state[prefix + ".mlp.2.bias"]))
    out = c1 * w[:, 0:C] + c2 * w[:, C:2 * C] + c3 * w[:, 2 * C:3 * C] + c4 * w[:, 3 * C:]
    return cbr(F.conv2d(out, state[prefix + ".final_conv.0.weight"], state[prefix + ".final_conv.0.bias"]),
               ".final_conv.1")


def a_unet_forward(x, state, training=True, update_stats=True):
    """A_UNet.forward with every Dropout2d at p=0 — model/a_unet.py:59-105: the U-Net with an
    AMSFN after each DoubleConv (the skip connections take the DoubleConv outputs, the pooling
    and upsampling take the AMSFN outputs) and a DoubleConv as the output layer."""
    dc = lambda t, name: double_conv(t, state, name, training, update_stats)
    ac = lambda t, name: amsfn(t, state, name, training, update_stats)
    e1 = dc(x, "encoder1")
    e2 = dc(maxpool2x2(ac(e1, "acpn0")), "encoder2")
    e3 = dc(maxpool2x2(ac(e2, "acpn1")), "encoder3")
    e4 = dc(maxpool2x2(ac(e3, "acpn2")), "encoder4")
    c5 = dc(maxpool2x2(ac(e4, "acpn3")), "center_conv")
    d = ac(dc(torch.cat([upsample2x_bilinear(c5), e4], dim=1), "decoder1"), "acpn5")
    d = ac(dc(torch.cat([upsample2x_bilinear(d), e3], dim=1), "decoder2"), "acpn6")
    d = ac(dc(torch.cat([upsample2x_bilinear(d), e2], dim=1), "decoder3"), "acpn7")
    d = ac(dc(torch.cat([upsample2x_bilinear(d), e1], dim=1), "decoder4"), "acpn8")
    return dc(d, "out_conv")


def _bn(z, state, bn_prefix, training, update_stats):
    """BatchNorm2d without an activation, parameters under `bn_prefix`."""
    g, b = state[bn_prefix + ".weight"], state[bn_prefix + ".bias"]
    if training:
        y, mean, var = batchnorm_train(z, g, b)
        if update_stats:
            update_running_stats(state, bn_prefix, mean, var, z.shape[0] * z.shape[2] * z.shape[3])
        return y
    return batchnorm_eval(z, g, b, state[bn_prefix + ".running_mean"], state[bn_prefix + ".running_var"])


def res_dconv(x, state, prefix, training=True, update_stats=True):
    """ResDConv.forward — modules.py:447-453: shortcut c1 = BN(conv1x1(x)); main path conv3x3 d=1 ->
    BN -> ReLU -> conv3x3 d=2 -> BN -> ReLU -> conv3x3 d=3 -> BN; out = relu(main + shortcut)."""
    conv = lambda t, p, d: F.conv2d(t, state[prefix + p + ".weight"], state[prefix + p + ".bias"], padding=d, dilation=d)
    re = _bn(F.conv2d(x, state[prefix + ".c1.0.weight"], state[prefix + ".c1.0.bias"]), state, prefix + ".c1.1",
             training, update_stats)
    y = torch.relu(_bn(conv(x, ".cbr1.0", 1), state, prefix + ".cbr1.1", training, update_stats))
    y = torch.relu(_bn(conv(y, ".cbr2.0", 2), state, prefix + ".cbr2.1", training, update_stats))
    y = _bn(conv(y, ".cbr3.0", 3), state, prefix + ".cbr3.1", training, update_stats)
    return torch.relu(y + re)


def mdam(x, state, prefix, group=32):
    """MDAM.forward — attention.py:171-190, step by step: channels split into `group` groups of
    cg; strip pooling over W and over H -> shared 1x1 conv -> sigmoid gates; GroupNorm (cg groups
    of one channel) of the gated map -> global average -> softmax over the cg channels; squeeze
    path conv3x3 -> global average -> 1x1 -> ReLU -> 1x1 -> sigmoid; out = x * sigmoid(sum)."""
    b, c, h, w = x.shape
    cg = c // group
    gx = x.reshape(b * group, cg, h, w)
    x_h = gx.mean(dim=3, keepdim=True)                                   # [bg, cg, h, 1]
    x_w = gx.mean(dim=2, keepdim=True).permute(0, 1, 3, 2)               # [bg, cg, w, 1]
    hw = F.conv2d(torch.cat([x_h, x_w], dim=2), state[prefix + ".conv1x1.weight"], state[prefix + ".conv1x1.bias"])
    x_h, x_w = torch.split(hw, [h, w], dim=2)
    gated = gx * torch.sigmoid(x_h) * torch.sigmoid(x_w.permute(0, 1, 3, 2))
    gn = F.group_norm(gated, cg, state[prefix + ".groupNorm.weight"], state[prefix + ".groupNorm.bias"], eps=1e-5)
    a = torch.softmax(gn.mean(dim=(2, 3), keepdim=True), dim=1)
    y = F.conv2d(gx, state[prefix + ".conv3x3.weight"], state[prefix + ".conv3x3.bias"], padding=1)
    y = y.mean(dim=(2, 3), keepdim=True)
    y = torch.sigmoid(F.conv2d(torch.relu(F.conv2d(y, state[prefix + ".fc.0.weight"])), state[prefix + ".fc.2.weight"]))
    return (gx * torch.sigmoid(a + y)).reshape(b, c, h, w)


def rdam_unet_forward(x, state, training=True, update_stats=True):
    """RDAM_UNet.forward with every Dropout2d at p=0 — model/rdam_unet.py:63-115."""
    rd = lambda t, name: res_dconv(t, state, name, training, update_stats)
    ac = lambda t, name: amsfn(t, state, name, training, update_stats)
    at = lambda t, name: mdam(t, state, name)
    e1 = F.conv2d(x, state["encoder1.weight"], state["encoder1.bias"], padding=1)
    m1 = at(e1, "att1")
    e2 = rd(maxpool2x2(ac(e1, "acpn0")), "encoder2")
    m2 = at(e2, "att2")
    e3 = rd(maxpool2x2(ac(e2, "acpn1")), "encoder3")
    m3 = at(e3, "att3")
    e4 = rd(maxpool2x2(ac(e3, "acpn2")), "encoder4")
    m4 = at(e4, "att4")
    c5 = double_conv(maxpool2x2(ac(e4, "acpn3")), state, "center_conv", training, update_stats)
    d = ac(rd(torch.cat([upsample2x_bilinear(c5), m4], dim=1), "decoder1"), "acpn5")
    d = ac(rd(torch.cat([upsample2x_bilinear(d), m3], dim=1), "decoder2"), "acpn6")
    d = ac(rd(torch.cat([upsample2x_bilinear(d), m2], dim=1), "decoder3"), "acpn7")
    d1 = F.conv2d(torch.cat([upsample2x_bilinear(d), m1], dim=1), state["decoder4.weight"], state["decoder4.bias"],
                  padding=1)
    return double_conv(ac(d1, "acpn8"), state, "out_conv", training, update_stats)


PARAM_SUFFIXES = (".0.weight", ".0.bias", ".1.weight", ".1.bias")


def param_keys(state):
    return [k for k in state if k.endswith(PARAM_SUFFIXES) or k.startswith("out_conv.")]


def unet_forward(x, state, training=True, update_stats=True, noise=None):
    """UNet.forward — unet.py:46-85.  `noise`: None (every Dropout2d at p=0 / eval) or the 7
    Dropout2d multipliers [B, C, 1, 1] (0 or 1/(1-p) per sample and channel) in call order:
    encoder_dropout1-4, bottleneck_dropout, decoder_dropout1-2 (unet.py:49,53,57,61,64,69,74)."""
    dc = lambda t, name: double_conv(t, state, name, training, update_stats)
    dr = (lambda t, i: t) if noise is None else (lambda t, i: t * noise[i].to(t.dtype))
    x1 = dc(x, "encoder1")
    x2 = dc(dr(maxpool2x2(x1), 0), "encoder2")
    x3 = dc(dr(maxpool2x2(x2), 1), "encoder3")
    x4 = dc(dr(maxpool2x2(x3), 2), "encoder4")
    x5 = dr(maxpool2x2(x4), 3)
    y = dr(dc(x5, "center_conv"), 4)
    y = dr(dc(torch.cat([upsample2x_bilinear(y), x4], dim=1), "decoder1"), 5)
    y = dr(dc(torch.cat([upsample2x_bilinear(y), x3], dim=1), "decoder2"), 6)
    y = dc(torch.cat([upsample2x_bilinear(y), x2], dim=1), "decoder3")
    y = dc(torch.cat([upsample2x_bilinear(y), x1], dim=1), "decoder4")
    return F.conv2d(y, state["out_conv.weight"], state["out_conv.bias"])   # unet.py:44,84


def unet_forward_bf16_storage(x, state):
    """Training-mode forward in float64 arithmetic with every tensor the bfloat16 compute
    mode STORES in bfloat16 rounded at that point: the image, the packed conv weights, each
    conv output z (its BatchNorm statistics are taken before the rounding, from the
    accumulators), each activation, the upsampled tensor and the logits.  It separates the
    error of the storage format (which any bfloat16 implementation has: about 1-2e-2 relative
    L2 on the logits from the weight rounding alone, 2.5-3.5e-2 in total on this network) from
    errors of the kernels."""
    bf = lambda t: t.to(torch.bfloat16).to(t.dtype)

    def cbr(t, p):
        z = F.conv2d(t, bf(state[p + ".0.weight"]), state[p + ".0.bias"], padding=1)
        mean = z.mean(dim=(0, 2, 3))
        var = z.var(dim=(0, 2, 3), unbiased=False)
        scale = torch.rsqrt(var + BN_EPS) * state[p + ".1.weight"]
        y = (bf(z) - mean[None, :, None, None]) * scale[None, :, None, None] + state[p + ".1.bias"][None, :, None, None]
        return bf(torch.relu(y))

    dc = lambda t, n: cbr(cbr(t, n + ".cbr1"), n + ".cbr2")
    up = lambda t: bf(upsample2x_bilinear(t))
    x1 = dc(bf(x), "encoder1")
    x2 = dc(maxpool2x2(x1), "encoder2")
    x3 = dc(maxpool2x2(x2), "encoder3")
    x4 = dc(maxpool2x2(x3), "encoder4")
    y = dc(maxpool2x2(x4), "center_conv")
    y = dc(torch.cat([up(y), x4], dim=1), "decoder1")
    y = dc(torch.cat([up(y), x3], dim=1), "decoder2")
    y = dc(torch.cat([up(y), x2], dim=1), "decoder3")
    y = dc(torch.cat([up(y), x1], dim=1), "decoder4")
    return bf(F.conv2d(y, state["out_conv.weight"], state["out_conv.bias"]))


# ----------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------
def dice_sums(logits, target):
    """The three pooled sums of loss_fn.py:79-80 for the foreground classes:
    I_c = sum p_c t_c, P_c = sum p_c, T_c = sum t_c over (B,H,W)."""
    C = logits.shape[1]
    p = torch.softmax(logits, dim=1)
    t = F.one_hot(target.to(torch.int64), num_classes=C).permute(0, 3, 1, 2).to(p.dtype)
    p, t = p[:, 1:], t[:, 1:]
    return (p * t).sum(dim=(0, 2, 3)), p.sum(dim=(0, 2, 3)), t.sum(dim=(0, 2, 3))


def dice_loss(logits, target, class_names, smooth=1e-5):
    """diceloss.__call__ — loss_fn.py:63-88.  total = sum(fg losses) / C (C includes
    background); 'total_loss' is the LAST key."""
    C = logits.shape[1]
    inter, psum, tsum = dice_sums(logits, target)
    loss = 1 - (2 * inter) / (psum + tsum + smooth)
    out = OrderedDict((name, loss[i]) for i, name in enumerate(class_names))
    out["total_loss"] = loss.sum() / C
    return out


def iou_loss(logits, target, class_names, smooth=1e-5):
    """IOULoss.__call__ — loss_fn.py:98-124: iou = I / (P + T - I + smooth), loss = 1 - iou,
    total = sum(fg losses) / C; 'total_loss' is the LAST key."""
    C = logits.shape[1]
    inter, psum, tsum = dice_sums(logits, target)
    loss = 1 - inter / (psum + tsum - inter + smooth)
    out = OrderedDict((name, loss[i]) for i, name in enumerate(class_names))
    out["total_loss"] = loss.sum() / C
    return out


def ce_map(logits, target, label_smoothing=1e-7):
    """nn.CrossEntropyLoss(label_smoothing=1e-7, reduction='none', ignore_index=0)
    — loss_fn.py:199-203.  Per pixel: (1-ls)*(-logp[t]) + ls*mean_c(-logp[c]);
    zero where target == 0."""
    logp = torch.log_softmax(logits, dim=1)
    nll = -logp.gather(1, target[:, None].to(torch.int64)).squeeze(1)
    smooth = -logp.mean(dim=1)
    out = (1 - label_smoothing) * nll + label_smoothing * smooth
    return torch.where(target == 0, torch.zeros_like(out), out)


# ----------------------------------------------------------------------------
# metrics
# ----------------------------------------------------------------------------
def argmax_first(logits_np):
    """torch.argmax over the class axis: first maximal index; NaN counts as maximal."""
    x = np.asarray(logits_np)
    B, C = x.shape[0], x.shape[1]
    best = x[:, 0].copy()
    idx = np.zeros(best.shape, dtype=np.int64)
    for c in range(1, C):
        v = x[:, c]
        take = (v > best) | (np.isnan(v) & ~np.isnan(best))
        best = np.where(take, v, best)
        idx = np.where(take, c, idx)
    return idx


def confusion_counts(logits, target):
    """Per-sample confusion matrix cm[b, t, p] (int64): pixels of sample b with
    target class t and predicted class p = argmax(logits).  Every TP/FN/FP/TN of
    metrics.py:31-34 is a sum of entries of this matrix."""
    x = logits.detach().to(torch.float32).cpu().numpy()
    t = target.detach().cpu().numpy().astype(np.int64)
    B, C = x.shape[0], x.shape[1]
    pred = argmax_first(x)
    if t.min() < 0 or t.max() >= C:
        raise RuntimeError("Class values must be smaller than num_classes.")
    cm = np.zeros((B, C, C), dtype=np.int64)
    for b in range(B):
        flat = t[b].reshape(-1) * C + pred[b].reshape(-1)
        cm[b] = np.bincount(flat, minlength=C * C).reshape(C, C)
    return cm


def _counts_to_onehot_terms(cm, K):
    """TP, FN, FP, TN as int64 [B, K] for a K-wide one-hot (K may exceed C: the extra
    classes are empty, exactly like F.one_hot(..., num_classes=9) on 4-class data)."""
    B, C, _ = cm.shape
    if C > K:
        raise RuntimeError("Class values must be smaller than num_classes.")
    hw = cm[0].sum()
    tp = np.zeros((B, K), dtype=np.int64)
    fn = np.zeros((B, K), dtype=np.int64)
    fp = np.zeros((B, K), dtype=np.int64)
    d = np.einsum("bcc->bc", cm)
    tp[:, :C] = d
    fn[:, :C] = cm.sum(axis=2) - d
    fp[:, :C] = cm.sum(axis=1) - d
    tn = hw - tp - fn - fp
    return tp, fn, fp, tn


def metrics_from_counts(cm, n_names, smooth=1e-5):
    """Metrics.update post-math (metrics.py:38-152) from the integer counts, with the
    reference's dtypes: int64 counts -> float32 ratios (torch default dtype) ->
    Python floats -> float64 ndarray (6, n+1), rows
    [recall, precision, dice, f1, mIoU, accuracy]; nan_to_num at the end.
    The recall/precision one-hots are 9 wide regardless of n (metrics.py:45-46,59-60)."""
    n = n_names
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    # recall / precision: K = 9, per-sample ratio then batch mean
    tp, fn, fp, _ = map(T, _counts_to_onehot_terms(cm, 9))
    recall = (tp / (tp + fn + smooth)).mean(dim=0)
    recalls = [recall[i + 1].item() for i in range(n)] + [recall[1:].mean().item()]
    precision = (tp / (tp + fp + smooth)).mean(dim=0)
    precisions = [precision[i + 1].item() for i in range(n)] + [precision[1:].mean().item()]
    f1s = [2 * (recalls[i] * precisions[i]) / (recalls[i] + precisions[i] + smooth) for i in range(n + 1)]
    # dice / mIoU: K = n+1, float32 sums of 0/1 pooled over the batch
    if cm.shape[1] > n + 1:
        raise RuntimeError("Class values must be smaller than num_classes.")
    K = n + 1
    tpk, fnk, fpk, tnk = _counts_to_onehot_terms(cm, K)
    inter = T(tpk.sum(axis=0)).to(torch.float32)
    psum = T((tpk + fpk).sum(axis=0)).to(torch.float32)
    tsum = T((tpk + fnk).sum(axis=0)).to(torch.float32)
    union = psum + tsum
    dice = (2 * inter) / (union + smooth)
    dices = [dice[i + 1].item() for i in range(n)] + [dice[1:].mean().item()]
    union_i = psum + tsum - inter
    iou = inter / (union_i + smooth)
    mious = [iou[i + 1].item() for i in range(n)] + [iou[1:].mean().item()]
    # accuracy: K = n+1, per-sample then batch mean
    tpa, fna, fpa, tna = map(T, (tpk, fnk, fpk, tnk))
    acc = ((tpa + tna) / (tpa + tna + fna + fpa + smooth)).mean(dim=0)
    accs = [acc[i + 1].item() for i in range(n)] + [acc[1:].mean().item()]
    m = np.stack([recalls, precisions, dices, f1s, mious, accs], axis=0)
    return np.nan_to_num(m)


def metrics_update(logits, target, class_names, smooth=1e-5):
    return metrics_from_counts(confusion_counts(logits, target), len(class_names), smooth)


def metrics_update_per_call(logits, target, class_names, smooth=1e-5):
    """The same (6, n+1) matrix computed the way Metrics.update (metrics.py:137-152) spends its
    time: every metric function starts over from the logits — recall, precision (each once more
    inside f1_score, metrics.py:76-77) and accuracy take argmax -> one-hot of prediction and
    target (9 wide for recall/precision, n+1 otherwise) -> the four torch.where(...).sum((-2,-1))
    passes of compute_confusion_matrix (metrics.py:31-34); dice and mIoU take softmax -> argmax
    -> one-hots -> pooled products (metrics.py:87-97,106-115).  7 argmax and 14 one-hot tensors
    per call.  Used where the CPU baseline must cost what the reference costs; values equal
    metrics_update's (tests/test_oracle_golden.py)."""
    n = len(class_names)
    target = target.to(torch.int64)

    def onehots(K, softmax_first):
        x = torch.softmax(logits, dim=1) if softmax_first else logits
        pred = F.one_hot(torch.argmax(x, dim=1).to(torch.int64), num_classes=K).permute(0, 3, 1, 2).float()
        return pred, F.one_hot(target, num_classes=K).permute(0, 3, 1, 2).float()

    def four_counts(pred, tgt):
        hit = lambda a, b: torch.where((pred == a) & (tgt == b), 1, 0).sum(dim=(-2, -1))
        return hit(1, 1), hit(0, 1), hit(1, 0), hit(0, 0)          # TP, FN, FP, TN

    def row(v):                                                      # n per-class values + mean of [1:]
        return [v[i + 1].item() for i in range(n)] + [v[1:].mean().item()]

    def recall():
        tp, fn, _, _ = four_counts(*onehots(9, False))
        return row((tp / (tp + fn + smooth)).mean(dim=0))

    def precision():
        tp, _, fp, _ = four_counts(*onehots(9, False))
        return row((tp / (tp + fp + smooth)).mean(dim=0))

    def pooled(iou):
        pred, tgt = onehots(n + 1, True)
        inter = (pred * tgt).sum(dim=(0, -2, -1))
        union = pred.sum(dim=(0, -2, -1)) + tgt.sum(dim=(0, -2, -1))
        return row(inter / (union - inter + smooth) if iou else (2 * inter) / (union + smooth))

    recalls, precisions, dices = recall(), precision(), pooled(False)
    r2, p2 = recall(), precision()                                   # f1_score recomputes both
    f1s = [2 * (r2[i] * p2[i]) / (r2[i] + p2[i] + smooth) for i in range(n + 1)]
    mious = pooled(True)
    tp, fn, fp, tn = four_counts(*onehots(n + 1, False))
    accs = row(((tp + tn) / (tp + tn + fn + fp + smooth)).mean(dim=0))
    return np.nan_to_num(np.stack([recalls, precisions, dices, f1s, mious, accs], axis=0))


# ----------------------------------------------------------------------------
# one training step (the CPU baseline's unit of work)
# ----------------------------------------------------------------------------
def clip_grad_norm(grads, max_norm=1.0, eps=1e-6):
    """torch.nn.utils.clip_grad_norm_ (train_and_eval.py:115): global L2 norm,
    coef = clamp(max_norm/(norm+1e-6), max=1), grads *= coef.  Returns the norm."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads]))
    coef = torch.clamp(max_norm / (total + eps), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


def adamw_step(params, grads, exp_avg, exp_avg_sq, step, lr=8e-4, betas=(0.95, 0.999),
               eps=1e-8, weight_decay=1e-4):
    """torch.optim.AdamW single-tensor update (train.py:221-226 defaults)."""
    b1, b2 = betas
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    with torch.no_grad():
        for p, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
            p.mul_(1 - lr * weight_decay)
            m.lerp_(g, 1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
            p.addcdiv_(m, denom, value=-lr / bc1)


class TrainState:
    """Parameters + AdamW moments for train_step."""

    def __init__(self, state):
        self.state = state
        self.keys = param_keys(state)
        for k in self.keys:
            state[k].requires_grad_(True)
        self.m = [torch.zeros_like(state[k]) for k in self.keys]
        self.v = [torch.zeros_like(state[k]) for k in self.keys]
        self.step = 0


def train_step(ts, images, masks, class_names, lr=8e-4, timers=None, per_call_metrics=False):
    """One batch of train_one_epoch (train_and_eval.py:63-121) on the fp32 CPU path
    (autocast and GradScaler disable themselves without CUDA): forward, Dice loss,
    metrics, backward, clip_grad_norm_(1.0), AdamW.  Returns (loss dict values as
    floats, metrics matrix, logits).  per_call_metrics: compute the metrics with the
    reference's own pass structure (its cost), not from one count matrix."""
    import time
    tick = time.perf_counter
    st = ts.state
    params = [st[k] for k in ts.keys]
    for p in params:
        p.grad = None
    t0 = tick()
    logits = unet_forward(images, st, training=True)
    t1 = tick()
    ld = dice_loss(logits, masks, class_names)
    t2 = tick()
    met = (metrics_update_per_call(logits.detach(), masks, class_names) if per_call_metrics
           else metrics_update(logits, masks, class_names))
    t3 = tick()
    ld["total_loss"].backward()
    t4 = tick()
    grads = [p.grad for p in params]
    clip_grad_norm(grads, 1.0)
    t5 = tick()
    ts.step += 1
    adamw_step(params, grads, ts.m, ts.v, ts.step, lr=lr)
    t6 = tick()
    if timers is not None:
        for k, v in zip(("fwd", "loss", "metrics", "bwd", "clip", "adamw"),
                        (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
            timers[k] = timers.get(k, 0.0) + v
    return [float(v) for v in ld.values()], met, logits.detach()


# ----------------------------------------------------------------------------
# sliding-window prediction
# ----------------------------------------------------------------------------
def slide_pad(image, window, stride):
    """_pad_image — slide_predict.py:15-29 (reflect pad on bottom/right)."""
    h, w = image.shape[:2]
    pad_h = (window - (h % stride)) % window
    pad_w = (window - (w % stride)) % window
    if pad_h > 0 or pad_w > 0:
        image = np.pad(image, ((0, pad_h), (0, pad_w), (0, 0)), "reflect")
    return image, (pad_h, pad_w)


def slide_positions(h, w, window, stride):
    """_split_image — slide_predict.py:31-58: (num+1)^2 windows, the last row/col
    clamped to the border (which duplicates the previous one when sizes divide)."""
    pos = []
    num_h = (h - window) // stride + 1
    num_w = (w - window) // stride + 1
    for i in range(num_h + 1):
        for j in range(num_w + 1):
            y0, x0 = i * stride, j * stride
            y1, x1 = y0 + window, x0 + window
            if y1 > h:
                y1, y0 = h, h - window
            if x1 > w:
                x1, x0 = w, w - window
            pos.append((y0, y1, x0, x1))
    return pos


def slide_gaussian(window):
    """The per-tile weight of slide_predict.py:95-100 (float64):
    d = sqrt((i-w//2)^2 + (j-w//2)^2); weight = exp(-d^2 / (2*(w/4)^2))."""
    sigma = window / 4
    i = np.arange(window, dtype=np.float64) - window // 2
    d = (i[:, None] ** 2 + i[None, :] ** 2) ** 0.5
    return np.exp(-d ** 2 / (2 * sigma ** 2))


def slide_blend(class_maps, positions, padded_hw, orig_hw, window):
    """slide_predict.py:74-112: float32 accumulators updated tile by tile IN ORDER
    (each += rounds float64 -> float32), divide where weight != 0, round, uint8."""
    ph, pw = padded_hw
    output = np.zeros((ph, pw), dtype=np.float32)
    weight = np.zeros((ph, pw), dtype=np.float32)
    g = slide_gaussian(window)
    for cm, (y0, y1, x0, x1) in zip(class_maps, positions):
        output[y0:y1, x0:x1] += cm * g
        weight[y0:y1, x0:x1] += g
    oh, ow = orig_hw
    fo, fw = output[:oh, :ow], weight[:oh, :ow]
    fo = np.divide(fo, fw, where=fw != 0)
    return np.round(fo).astype(np.uint8)


# ----------------------------------------------------------------------------
# Dropout2d multipliers from a counter-based generator (model/unet.py:49-74 applies nn.Dropout2d:
# one Bernoulli(1-p) per (sample, channel), kept maps scaled by 1/(1-p)).  The reference draws them
# from torch's generator; the counter-based form has no reference counterpart to pin against, so it
# is pinned by the published known-answer vectors of Philox4x32-10 (Salmon, Moraes, Dror, Shaw,
# "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123 kat_vectors).
# ----------------------------------------------------------------------------
def philox4x32_10(counter, key):
    """counter: 4 uint32 values (arrays broadcast), key: 2 uint32 values -> 4 uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint64) for v in counter)
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    m32 = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & m32
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & m32
        c1, c3, c0, c2 = p1 & m32, p0 & m32, n0, n2
        k0 = (k0 + np.uint64(0x9E3779B9)) & m32
        k1 = (k1 + np.uint64(0xBB67AE85)) & m32
    return tuple(v.astype(np.uint32) for v in (c0, c1, c2, c3))


def dropout2d_noise_philox(seed, offset, B, C, p):
    """float32 [B, C]: 1/(1-p) where the top 24 bits of word 0 of philox(counter=(b*C+c, 0, offset), key=seed),
    as a fraction of 2**24, are below 1-p, else 0 (float32 comparisons, as the kernel makes them)."""
    i = np.arange(B * C, dtype=np.uint64)
    z = np.zeros_like(i)
    w0 = philox4x32_10((i, z, z + np.uint64(offset & 0xFFFFFFFF), z + np.uint64(offset >> 32)),
                       (seed & 0xFFFFFFFF, seed >> 32))[0]
    u = (w0 >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    keep = np.float32(1.0) - np.float32(p)
    scale = np.float32(1.0) / keep if p < 1.0 else np.float32(0.0)
    return np.where(u < keep, scale, np.float32(0.0)).astype(np.float32).reshape(B, C)
