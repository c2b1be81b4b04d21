"""GPU suite (run with -m gpu on an MI355X): every HIP entry point, called through the
C-ABI (ctypes), against the CPU oracle on the same seeded inputs and against the golden
vectors generated from the reference.

Tolerances: integer outputs (confusion counts, metric matrices) are compared bit-exactly;
float32 mode at 1e-5 relative to the tensor's scale (a few 1e-5 for long reductions);
bfloat16 mode at 1e-2 relative, the bound the task states for bf16.
"""
import numpy as np
import pytest
import torch

import make_golden as mg
from oracle import datagen as dg
from oracle import unet_oracle as orc

pytestmark = pytest.mark.gpu

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
DEV = "cuda:0"


@pytest.fixture(scope="module")
def F_():
    if not torch.cuda.is_available():
        pytest.fail("GPU suite needs a HIP device")
    from medical_image_segmentation_amd import _lib, functional
    _lib.load()        # fails loudly if libmisamd.so is missing
    return functional


def act(a, dtype=torch.float32):
    """numpy/torch NCHW -> channels-last device activation."""
    t = a if torch.is_tensor(a) else T(a)
    return t.to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu() if torch.is_tensor(b) else T(np.asarray(b)).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


TOL = {torch.float32: 2e-5, torch.bfloat16: 1e-2}
DTYPES = [torch.float32, torch.bfloat16]


def q(a, dtype):
    """Round test inputs to the compute dtype so both sides see identical values."""
    t = a if torch.is_tensor(a) else T(a)
    return t.to(dtype).to(torch.float32)


def lazy_pair(C, seed):
    """A BatchNorm-like per-channel affine [2, C] (both signs of scale) for lazy inputs."""
    sc = dg.hash_uniform((C,), seed, 0.4, 1.6) * np.where(dg.hash_uniform((C,), seed + 1, 0, 1) < 0.25, -1.0, 1.0)
    sh = dg.hash_uniform((C,), seed + 2, -0.5, 0.5)
    return T(np.stack([sc, sh]).astype(np.float32))


def lazy_value(z, ss, dtype):
    """relu(z*scale + shift) of a lazy activation, from the dtype-rounded z; one rounding to
    float32 like the kernels' fused multiply-add."""
    zq = q(z, dtype).double()
    a = zq * ss[0].double()[None, :, None, None] + ss[1].double()[None, :, None, None]
    return torch.relu(a).float()


# ---------------------------------------------------------------------------
# loss + metrics
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(mg.loss_cases()))
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_dice_ce_vs_oracle_and_golden(F_, golden, name, dtype, layout):
    g = golden("loss")
    c = mg.loss_cases()[name]
    x, t = mg.loss_inputs(c)
    xq = q(x, dtype)
    names = [f"c{i}" for i in range(1, c["C"])]
    from medical_image_segmentation_amd.utils.loss_fn import diceloss, ce_map
    xd = xq.to(DEV).to(dtype)
    if layout == "nhwc":
        xd = xd.contiguous(memory_format=torch.channels_last)
    xd.requires_grad_(True)
    td = T(t).to(DEV)
    ld = diceloss(names)(xd, td)
    assert list(ld.keys()) == names + ["total_loss"]
    ld["total_loss"].backward()
    xo = xq.clone().double().requires_grad_(True)
    lo = orc.dice_loss(xo, T(t), names)
    lo["total_loss"].backward()
    got = np.array([float(v) for v in ld.values()])
    np.testing.assert_allclose(got, [float(v) for v in lo.values()], rtol=2e-6, atol=2e-6)
    if dtype == torch.float32:
        np.testing.assert_allclose(got, g[f"{name}.dice"], rtol=1e-5, atol=1e-6)
        assert rel_err(xd.grad, g[f"{name}.dice_dlogits"]) < 2e-5
    assert rel_err(xd.grad, xo.grad) < TOL[dtype]
    ce = ce_map(xd.detach(), td)
    assert rel_err(ce, orc.ce_map(xq.double(), T(t))) < 1e-5
    if dtype == torch.float32:
        assert rel_err(ce, g[f"{name}.ce_map"]) < 1e-5


@pytest.mark.parametrize("name", list(mg.loss_cases()))
@pytest.mark.parametrize("dtype", DTYPES)
def test_iou_loss_vs_oracle_and_golden(F_, golden, name, dtype):
    """IOULoss (loss_fn.py:90-124): values against the reference's own (golden) and the oracle,
    gradient against the oracle's autograd."""
    g = golden("loss")
    c = mg.loss_cases()[name]
    x, t = mg.loss_inputs(c)
    xq = q(x, dtype)
    names = [f"c{i}" for i in range(1, c["C"])]
    from medical_image_segmentation_amd.utils.loss_fn import IOULoss
    xd = xq.to(DEV).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ld = IOULoss(names)(xd, T(t).to(DEV))
    assert list(ld.keys()) == names + ["total_loss"]
    ld["total_loss"].backward()
    xo = xq.clone().double().requires_grad_(True)
    lo = orc.iou_loss(xo, T(t), names)
    lo["total_loss"].backward()
    got = np.array([float(v.detach()) for v in ld.values()])
    np.testing.assert_allclose(got, [float(v) for v in lo.values()], rtol=2e-6, atol=2e-6)
    if dtype == torch.float32:
        np.testing.assert_allclose(got, g[f"{name}.iou"], rtol=1e-5, atol=1e-6)
    assert rel_err(xd.grad, xo.grad) < TOL[dtype]


def test_ce_gradient(F_):
    c = mg.loss_cases()["loss_c4"]
    x, t = mg.loss_inputs(c)
    from medical_image_segmentation_amd.utils.loss_fn import DiceCELoss
    xd = T(x).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ld = DiceCELoss(["a", "b", "c"], ce_weight=0.7)(xd, T(t).to(DEV))
    ld["total_loss"].backward()
    xo = T(x).double().requires_grad_(True)
    ref = orc.dice_loss(xo, T(t), ["a", "b", "c"])["total_loss"] + 0.7 * orc.ce_map(xo, T(t)).mean()
    ref.backward()
    assert abs(float(ld["total_loss"]) - float(ref)) < 1e-5
    assert rel_err(xd.grad, xo.grad) < 2e-5


@pytest.mark.parametrize("name", list(mg.metric_cases()))
@pytest.mark.parametrize("dtype", DTYPES)
def test_metrics_bit_exact(F_, golden, name, dtype):
    g = golden("metrics")
    c = mg.metric_cases()[name]
    x, t = mg.metric_inputs(c)
    from medical_image_segmentation_amd.utils.metrics import Metrics
    names = [f"c{i}" for i in range(1, c["n"] + 1)]
    M = Metrics(names)
    xd = T(x).to(DEV).to(dtype)
    td = T(t).to(DEV)
    cm, status = M.confusion_counts(xd, td)
    assert int(status.item()) == 0
    cm_ref = orc.confusion_counts(xd.float().cpu(), T(t))       # same (possibly bf16-rounded) logits
    assert np.array_equal(cm.cpu().numpy(), cm_ref)
    m = M.update(xd, td)
    assert m.dtype == np.float64 and m.shape == (6, c["n"] + 1)
    assert np.array_equal(m, orc.metrics_from_counts(cm_ref, c["n"]))
    if dtype == torch.float32:
        tp, fn, fp, tn = orc._counts_to_onehot_terms(cm.cpu().numpy(), c["n"] + 1)
        for a, k in ((tp, "TP"), (fn, "FN"), (fp, "FP"), (tn, "TN")):
            assert np.array_equal(a, g[f"{name}.{k}"])
        assert np.array_equal(m, g[f"{name}.matrix"])
    # channels-last logits give the same counts
    cm2, _ = M.confusion_counts(xd.contiguous(memory_format=torch.channels_last), td)
    assert torch.equal(cm, cm2)


def test_metrics_reuse_counts_from_loss(F_):
    c = mg.metric_cases()["met_n3"]
    x, t = mg.metric_inputs(c)
    from medical_image_segmentation_amd.utils.metrics import Metrics
    from medical_image_segmentation_amd.utils.loss_fn import diceloss
    xd, td = T(x).to(DEV), T(t).to(DEV)
    diceloss(["a", "b", "c"])(xd, td)
    assert getattr(xd, "_mis_counts")[0] is td
    m1 = Metrics(["a", "b", "c"]).update(xd, td)
    m2 = Metrics(["a", "b", "c"]).update(xd.clone(), td)
    assert np.array_equal(m1, m2)


def test_metrics_bad_target_raises(F_):
    from medical_image_segmentation_amd.utils.metrics import Metrics
    x = torch.zeros(1, 4, 16, 16, device=DEV)
    t = torch.full((1, 16, 16), 7, dtype=torch.int64, device=DEV)
    with pytest.raises(RuntimeError):
        Metrics(["a", "b", "c"]).update(x, t)


def test_loss_metrics_full_size_properties(F_):
    """BASELINE size (16 x 4 x 512^2): counts against the oracle, sums against identities."""
    B, C, H, W = 16, 4, 512, 512
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, C, H, W, generator=g).to(torch.bfloat16)
    t = torch.randint(0, C, (B, H, W), generator=g)
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last)
    sums, _, cm, status, _, _ = F_.loss_metrics_raw(xd, t.to(DEV), True, False, True)
    cm = cm.cpu().numpy()
    assert np.array_equal(cm, orc.confusion_counts(x.float(), t))
    assert (cm.sum(axis=(1, 2)) == H * W).all()
    s = sums.cpu().double().numpy()
    assert abs(s[1].sum() - B * H * W) / (B * H * W) < 1e-6            # softmax sums to one
    assert np.array_equal(s[2], cm.sum(axis=(0, 2)).astype(np.float64))   # T_c are exact integers
    ref = orc.dice_sums(x.double(), t)
    np.testing.assert_allclose(s[0][1:], ref[0].numpy(), rtol=2e-6)
    np.testing.assert_allclose(s[1][1:], ref[1].numpy(), rtol=2e-6)


# ---------------------------------------------------------------------------
# pool / upsample+concat
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
def test_maxpool_golden_and_ties(F_, golden, dtype):
    g = golden("pool_up")
    x = dg.hash_normalish((2, 3, 8, 12), 31)
    x = np.maximum(x, 0)
    x[0, 0, :4, :4] = 0.5
    gy = dg.hash_normalish((2, 3, 4, 6), 32)
    xd = act(x, dtype).requires_grad_(True)
    y = F_.MaxPool2x2Fn.apply(xd, None)
    y.backward(act(gy, dtype))
    xo = q(x, dtype).requires_grad_(True)
    yo = orc.maxpool2x2(xo)
    yo.backward(q(gy, dtype))
    assert torch.equal(y.float().cpu(), yo.detach())
    assert torch.equal(xd.grad.float().cpu(), xo.grad)
    if dtype == torch.float32:
        assert np.array_equal(y.detach().cpu().numpy(), g["pool.y"])
        assert np.array_equal(xd.grad.cpu().numpy(), g["pool.dx"])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 8, 16, 24), (1, 32, 64, 64), (3, 5, 6, 10)])
@pytest.mark.parametrize("lazy", [False, True])
def test_maxpool_random(F_, dtype, shape, lazy):
    """Plain input, and a lazy input (z, ss) whose value relu(z*scale+shift) is formed on load;
    the gradient returned is the one with respect to that value."""
    gy = dg.hash_normalish((shape[0], shape[1], shape[2] // 2, shape[3] // 2), 302)
    if lazy:
        x = dg.hash_normalish(shape, 301)
        ss = lazy_pair(shape[1], 310)
        xo = lazy_value(x, ss, dtype).requires_grad_(True)
        ssd = ss.to(DEV)
    else:
        x = np.maximum(dg.hash_normalish(shape, 301), 0)      # post-ReLU: many exact-zero ties
        xo = q(x, dtype).requires_grad_(True)
        ssd = None
    xd = act(x, dtype).requires_grad_(True)
    y = F_.MaxPool2x2Fn.apply(xd, ssd)
    y.backward(act(gy, dtype))
    yo = orc.maxpool2x2(xo)
    yo.backward(q(gy, dtype))
    assert torch.equal(y.float().cpu(), q(yo.detach(), dtype))
    assert torch.equal(xd.grad.float().cpu(), xo.grad)
    # pooled + skip use of the same tensor: both gradients are summed inside the pooling backward
    gs = dg.hash_normalish(shape, 303)
    xd2 = act(x, dtype).requires_grad_(True)
    y2, skip = F_.PoolSkipFn.apply(xd2, ssd)
    assert skip.data_ptr() == xd2.data_ptr() and torch.equal(y2, y)
    torch.autograd.backward([y2, skip], [act(gy, dtype), act(gs, dtype)])
    want = q(xo.grad + q(gs, dtype), dtype)
    assert torch.equal(xd2.grad.float().cpu(), want)
    xd3 = act(x, dtype).requires_grad_(True)      # only the skip branch carries a gradient
    _, skip3 = F_.PoolSkipFn.apply(xd3, ssd)
    skip3.backward(act(gs, dtype))
    assert torch.equal(xd3.grad.float().cpu(), q(gs, dtype))


@pytest.mark.parametrize("dtype", DTYPES)
def test_upsample_cat_golden(F_, golden, dtype):
    g = golden("pool_up")
    for name, (h, w) in {"up_2": (2, 2), "up_5": (5, 5), "up_3x7": (3, 7)}.items():
        x = dg.hash_normalish((2, 3, h, w), 33 + h)
        skip = dg.hash_normalish((2, 2, 2 * h, 2 * w), 43 + h)
        gy = dg.hash_normalish((2, 5, 2 * h, 2 * w), 53 + h)
        xd, sd = act(x, dtype).requires_grad_(True), act(skip, dtype).requires_grad_(True)
        y = F_.UpsampleCatFn.apply(xd, None, sd, None)
        y.backward(act(gy, dtype))
        tol = TOL[dtype]
        if dtype == torch.float32:
            assert rel_err(y, g[f"{name}.y"]) < tol
            assert rel_err(xd.grad, g[f"{name}.dx"]) < tol
            assert np.array_equal(sd.grad.cpu().numpy(), g[f"{name}.dskip"])
        xo, so = q(x, dtype).requires_grad_(True), q(skip, dtype).requires_grad_(True)
        yo = torch.cat([orc.upsample2x_bilinear(xo), so], dim=1)
        yo.backward(q(gy, dtype))
        assert rel_err(y, yo) < tol and rel_err(xd.grad, xo.grad) < tol
        assert torch.equal(sd.grad.float().cpu(), so.grad)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 16, 8, 8, 8), (1, 64, 16, 16, 32), (2, 8, 4, 6, 24)])
@pytest.mark.parametrize("lazy", [(False, False), (True, True), (True, False), (False, True)])
def test_upsample_cat_random(F_, dtype, shape, lazy):
    """Either input plain or lazy (value relu(z*scale+shift) formed on load)."""
    B, Cup, h, w, Cs = shape
    x = dg.hash_normalish((B, Cup, h, w), 401)
    skip = dg.hash_normalish((B, Cs, 2 * h, 2 * w), 402)
    gy = dg.hash_normalish((B, Cup + Cs, 2 * h, 2 * w), 403)
    ssl = lazy_pair(Cup, 410) if lazy[0] else None
    sss = lazy_pair(Cs, 420) if lazy[1] else None
    xd, sd = act(x, dtype).requires_grad_(True), act(skip, dtype).requires_grad_(True)
    y = F_.UpsampleCatFn.apply(xd, ssl.to(DEV) if lazy[0] else None, sd, sss.to(DEV) if lazy[1] else None)
    y.backward(act(gy, dtype))
    xo = (lazy_value(x, ssl, dtype) if lazy[0] else q(x, dtype)).requires_grad_(True)
    so = (lazy_value(skip, sss, dtype) if lazy[1] else q(skip, dtype)).requires_grad_(True)
    yo = torch.cat([orc.upsample2x_bilinear(xo), so], dim=1)
    yo.backward(q(gy, dtype))
    assert rel_err(y, yo) < TOL[dtype] and rel_err(xd.grad, xo.grad) < TOL[dtype]
    assert torch.equal(sd.grad.float().cpu(), so.grad)
    assert torch.equal(y[:, Cup:].float().cpu(), q(so.detach(), dtype))


# ---------------------------------------------------------------------------
# conv + BN + ReLU, 1x1 conv
# ---------------------------------------------------------------------------
class _RoundGrad(torch.autograd.Function):
    """Identity whose incoming gradient is rounded to `dtype` (a tensor the kernels store)."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.dtype = dtype
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype).to(g.dtype), None


def _rnd(t, dtype):
    """Round values to `dtype`, straight-through for the gradient."""
    return t + (t.detach().to(dtype).to(t.dtype) - t.detach())


def _cbr_oracle(x, w, b, gam, bet, gy, dtype, x_value=None):
    """float64 oracle of one conv+BN(train)+ReLU with the inputs rounded to `dtype`.
    In bfloat16 mode it also rounds the tensors the kernels store in bfloat16 — the packed
    weights, the conv output z (statistics are taken before that rounding, from the float32
    accumulators), the activation a and the gradient dz — so the comparison isolates kernel
    errors from the storage format's own rounding."""
    # parameters stay float32 masters (only the packed conv weights are rounded, below)
    x0 = q(x, dtype) if x_value is None else q(x_value, dtype)
    xs = [x0.double().requires_grad_(True)] + [T(v).double().requires_grad_(True) for v in (w, b, gam, bet)]
    xo, wo, bo, go, beo = xs
    if dtype == torch.float32:
        z = orc.conv3x3(xo, wo, bo)
        y, mean, var = orc.batchnorm_train(z, go, beo)
        a = torch.relu(y)
    else:
        z = orc.conv3x3(xo, _rnd(wo, dtype), bo)
        mean = z.mean(dim=(0, 2, 3))
        var = z.var(dim=(0, 2, 3), unbiased=False)
        zr = _rnd(_RoundGrad.apply(z, dtype), dtype)
        rstd = torch.rsqrt(var + orc.BN_EPS)
        y = (zr - mean[None, :, None, None]) * (rstd * go)[None, :, None, None] + beo[None, :, None, None]
        a = _rnd(torch.relu(y), dtype)
    a.backward(q(gy, dtype).double())
    return a.detach(), [v.grad for v in xs], mean.detach(), var.detach()


CBR_SHAPES = [(2, 1, 8, 16, 16), (2, 3, 8, 16, 16), (2, 8, 32, 16, 16), (1, 32, 16, 24, 40),
              (2, 16, 64, 8, 8), (1, 64, 8, 32, 32), (2, 24, 8, 8, 8),
              # image-facing layer across its 16x64 LDS tiles: ragged edges, several tiles per
              # workgroup, two channel slices, a channel count without 16-byte stores
              (1, 1, 8, 37, 70), (2, 3, 12, 19, 130), (1, 1, 16, 100, 64), (3, 1, 8, 48, 200),
              # narrow outputs: 1, 2, 3 (padded to 4) and 4 channels per thread of the direct kernel
              (2, 32, 1, 16, 16), (1, 16, 2, 24, 40), (2, 32, 3, 16, 24), (1, 8, 4, 19, 33), (2, 5, 1, 9, 11),
              # one output channel from 1 / 2 / 4 / 8 channel vectors (and the 8-byte bf16 vector), one input
              # channel into as many: row strips of 8, ragged widths, images smaller than a strip
              (2, 8, 1, 9, 13), (1, 16, 1, 20, 70), (1, 64, 1, 17, 33), (2, 4, 1, 12, 12), (3, 32, 1, 5, 130),
              (1, 1, 32, 21, 35), (1, 1, 64, 9, 9), (2, 1, 4, 8, 8)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", CBR_SHAPES)
@pytest.mark.parametrize("lazy", [False, True])
def test_conv_bn_relu(F_, dtype, shape, lazy):
    """conv+BN(train)+ReLU returns the lazy pair (z, ss); materialised it is the activation.
    With `lazy` the input itself is a lazy activation applied while the conv loads it."""
    B, Ci, Co, H, W = shape
    x = np.maximum(dg.hash_normalish((B, Ci, H, W), 501), 0) if Ci > 3 else dg.hash_normalish((B, Ci, H, W), 501)
    ss_in = x_value = None
    if lazy:
        x = dg.hash_normalish((B, Ci, H, W), 501)
        ss_in = lazy_pair(Ci, 510)
        x_value = lazy_value(x, ss_in, dtype)
    w = dg.hash_normalish((Co, Ci, 3, 3), 502, std=(2.0 / (9 * Co)) ** 0.5)
    b = dg.hash_uniform((Co,), 503, -0.1, 0.1)
    gam = dg.hash_uniform((Co,), 504, 0.5, 1.5)
    bet = dg.hash_uniform((Co,), 505, -0.2, 0.2)
    gy = dg.hash_normalish((B, Co, H, W), 506)
    conv = torch.nn.Conv2d(Ci, Co, 3, padding=1).to(DEV)
    bn = torch.nn.BatchNorm2d(Co).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(T(w)); conv.bias.copy_(T(b)); bn.weight.copy_(T(gam)); bn.bias.copy_(T(bet))
    xd = act(x, dtype).requires_grad_(True)
    z, ss = F_.conv_bn_relu(xd, ss_in.to(DEV) if lazy else None, conv, bn, True)
    assert ss.shape == (4, Co) and not ss.requires_grad      # scale, shift | mean, rstd
    a = F_.materialize(z, ss)
    a.backward(act(gy, dtype))
    ao, grads, mean, var = _cbr_oracle(x, w, b, gam, bet, gy, dtype, x_value)
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    gtol = 2e-4 if dtype == torch.float32 else 2e-2
    assert rel_err(a, ao) < tol
    assert rel_err(xd.grad, grads[0]) < gtol
    assert rel_err(conv.weight.grad, grads[1]) < gtol
    assert rel_err(bn.weight.grad, grads[3]) < gtol
    assert rel_err(bn.bias.grad, grads[4]) < gtol
    # conv-bias gradient is analytically zero: only its scale can be checked
    assert float(conv.bias.grad.abs().max()) < (1e-3 if dtype == torch.float32 else 5e-2) * float(bn.bias.grad.abs().max()) + 1e-4
    n = B * H * W
    assert rel_err(bn.running_mean, 0.1 * mean) < max(tol, 1e-4)
    assert rel_err(bn.running_var, 0.9 + 0.1 * var * n / (n - 1)) < max(tol, 1e-4)
    assert int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize("name,c", [("dc_a", dict(B=2, cin=3, mid=8, cout=8, hw=16, seed=11)),
                                    ("dc_b", dict(B=2, cin=16, mid=4, cout=16, hw=8, seed=12))])
def test_double_conv_golden(F_, golden, name, c):
    from medical_image_segmentation_amd.model.utils.modules import DoubleConv
    g = golden("double_conv")
    m = DoubleConv(c["cin"], c["cout"], c["mid"])
    st = mg.orc_double_conv_state(c["cin"], c["mid"], c["cout"], c["seed"])
    m.load_state_dict({k[4:]: v.clone() for k, v in st.items()})
    m.to(DEV).train()
    x = dg.hash_normalish((c["B"], c["cin"], c["hw"], c["hw"]), c["seed"] + 100)
    gy = dg.hash_normalish((c["B"], c["cout"], c["hw"], c["hw"]), c["seed"] + 200)
    xd = act(x).requires_grad_(True)
    y = m(xd)
    y.backward(act(gy))
    assert rel_err(y, g[f"{name}.train_y"]) < 2e-5
    assert rel_err(xd.grad, g[f"{name}.dx"]) < 2e-4
    for k, p in m.named_parameters():
        if k.endswith(".0.bias"):
            continue
        assert rel_err(p.grad, g[f"{name}.grad.{k}"]) < 5e-4, k
    for k, b in m.named_buffers():
        assert rel_err(b.float(), g[f"{name}.buf.{k}"]) < 1e-5, k
    m.eval()
    with torch.no_grad():
        assert rel_err(m(xd.detach()), g[f"{name}.eval_y"]) < 2e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 32, 4, 16, 16), (1, 8, 9, 8, 24), (2, 16, 2, 8, 8), (1, 64, 3, 8, 8),
                                   (1, 128, 4, 8, 8)])      # 2..16 lanes per pixel in the forward
@pytest.mark.parametrize("lazy", [False, True])
def test_conv1x1(F_, dtype, shape, lazy):
    B, Ci, Co, H, W = shape
    x = dg.hash_normalish((B, Ci, H, W), 601)
    ss = lazy_pair(Ci, 610) if lazy else None
    w = dg.hash_normalish((Co, Ci, 1, 1), 602, std=0.3)
    b = dg.hash_uniform((Co,), 603, -0.1, 0.1)
    gy = dg.hash_normalish((B, Co, H, W), 604)
    wd, bd = T(w).to(DEV).requires_grad_(True), T(b).to(DEV).requires_grad_(True)
    xd = act(x, dtype).requires_grad_(True)
    y = F_.Conv1x1Fn.apply(xd, ss.to(DEV) if lazy else None, wd, bd)
    y.backward(act(gy, dtype))
    xo = (lazy_value(x, ss, dtype) if lazy else q(x, dtype)).double().requires_grad_(True)
    wo, bo = T(w).double().requires_grad_(True), T(b).double().requires_grad_(True)
    yo = torch.nn.functional.conv2d(xo, wo, bo)
    yo.backward(q(gy, dtype).double())
    tol = TOL[dtype]
    assert rel_err(y, yo) < tol
    assert rel_err(xd.grad, xo.grad) < tol
    assert rel_err(wd.grad, wo.grad) < 5 * tol
    assert rel_err(bd.grad, bo.grad) < 5 * tol


# ---------------------------------------------------------------------------
# matrix-core conv kernels against the vector-ALU kernels and the oracle
# ---------------------------------------------------------------------------
MFMA_SHAPES = [  # B, Cin, Cout, H, W
    (2, 8, 32, 16, 16), (1, 16, 64, 32, 32), (1, 32, 16, 32, 48), (2, 64, 32, 16, 16), (1, 128, 64, 16, 16),
    (1, 64, 256, 16, 16), (1, 256, 512, 8, 8), (1, 512, 32, 16, 16), (2, 64, 8, 32, 32), (1, 24, 8, 20, 12),
    (3, 8, 8, 4, 4), (1, 32, 128, 2, 2), (1, 12, 16, 16, 16), (1, 4, 8, 16, 16), (1, 16, 24, 18, 34)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", MFMA_SHAPES)
@pytest.mark.parametrize("transposed", [False, True])
@pytest.mark.parametrize("lazy", [False, True])
def test_conv3x3_mfma_vs_direct_and_oracle(F_, dtype, shape, transposed, lazy):
    """z = conv(x, W) (+ statistics) for the forward operand and for the data-gradient
    operand (W^T, taps mirrored), on the matrix-core kernel and the vector-ALU kernel."""
    from medical_image_segmentation_amd import _lib
    B, Ci, Co, H, W = shape
    G = 8 if dtype == torch.bfloat16 else 4
    if Ci % G or (Ci > 4 * G and Ci % (4 * G)):
        pytest.skip("shape not taken by the matrix-core kernel in this dtype")
    x = dg.hash_normalish((B, Ci, H, W), 701)
    # forward conv Ci->Co uses master weights [Co,Ci,3,3]; the dgrad operand of master weights
    # [Ci,Co,3,3] also maps Ci->Co channels
    wm = dg.hash_normalish((Ci, Co, 3, 3) if transposed else (Co, Ci, 3, 3), 702, std=(2.0 / (9 * Ci)) ** 0.5)
    b = dg.hash_uniform((Co,), 703, -0.1, 0.1)
    xd = act(x, dtype)
    wd = T(wm).to(DEV)
    bd = None if transposed else T(b).to(DEV)
    ss = lazy_pair(Ci, 710) if lazy else None
    ssd = ss.to(DEV) if lazy else None
    # the matrix-core kernel rounds the applied activation to the compute dtype (its operand type)
    xv = q(lazy_value(x, ss, dtype), dtype) if lazy else q(x, dtype)
    outs = []
    for mode in (0, 1):
        _lib.call("mis_set_conv_path", mode)
        try:
            wp = F_.pack_weights(wd, dtype, transposed=transposed)
            z, stats = F_.conv3x3_raw(xd, wp, bd, Co, True, ssd)
            torch.cuda.synchronize()
        finally:
            _lib.call("mis_set_conv_path", 0)
        outs.append((z.float().cpu(), stats.cpu()))
    wq = q(wm, dtype).double()
    w_eff = wq.flip(2, 3).transpose(0, 1) if transposed else wq
    zo = torch.nn.functional.conv2d(xv.double(), w_eff, None if transposed else T(b).double(), padding=1)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    for z, stats in outs:
        assert rel_err(z, zo) < tol
        assert rel_err(stats[0], zo.sum(dim=(0, 2, 3))) < 1e-4 + (0 if dtype == torch.float32 else 1e-3)
        assert rel_err(stats[1], (zo * zo).sum(dim=(0, 2, 3))) < 1e-4 + (0 if dtype == torch.float32 else 1e-3)
    # the two kernels accumulate the same products in float32: near-identical results
    assert rel_err(outs[0][0], outs[1][0]) < (2e-6 if dtype == torch.float32 else 8e-3)


WGRAD_SHAPES = [  # B, Cin, Cout, H, W
    (2, 8, 32, 16, 32), (1, 16, 64, 32, 32), (1, 32, 16, 32, 64), (2, 64, 32, 16, 16), (1, 128, 64, 16, 32),
    (1, 64, 256, 16, 16), (1, 256, 512, 8, 8), (1, 512, 32, 16, 16), (2, 64, 8, 32, 32), (1, 24, 8, 20, 12),
    (3, 8, 8, 4, 4), (1, 32, 128, 2, 2), (1, 8, 16, 9, 33), (2, 128, 8, 16, 16), (1, 16, 24, 18, 34),
    (1, 64, 32, 40, 96), (2, 128, 64, 24, 64)]      # uneven tile ranges over 8 K-splits, several (co, ci) blocks


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", WGRAD_SHAPES)
@pytest.mark.parametrize("lazy", [False, True])
def test_conv3x3_wgrad_mfma_vs_direct_and_oracle(F_, dtype, shape, lazy):
    from medical_image_segmentation_amd import _lib
    B, Ci, Co, H, W = shape
    x = np.maximum(dg.hash_normalish((B, Ci, H, W), 801), 0)
    gz = dg.hash_normalish((B, Co, H, W), 802)
    ssd = None
    xv = q(x, dtype)
    if lazy:
        x = dg.hash_normalish((B, Ci, H, W), 801)
        ss = lazy_pair(Ci, 810)
        ssd = ss.to(DEV)
        xv = lazy_value(x, ss, dtype)
    xd, gd = act(x, dtype), act(gz, dtype)
    outs = []
    for mode in (0, 1):
        _lib.call("mis_set_conv_path", mode)
        try:
            outs.append(F_.conv3x3_wgrad_raw(xd, gd, ssd).cpu())
            torch.cuda.synchronize()
        finally:
            _lib.call("mis_set_conv_path", 0)
    xo = xv.double()
    wo = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(xo, wo, None, padding=1).backward(q(gz, dtype).double())
    # plain inputs are exact in both dtypes (float32 accumulation); a lazy bf16 input is rounded
    # to bf16 after the affine by the matrix-core kernel
    tol = 2e-5 if not (lazy and dtype == torch.bfloat16) else 3e-3
    for dw in outs:
        assert dw.shape == (Co, Ci, 3, 3)
        assert rel_err(dw, wo.grad) < tol
    assert rel_err(outs[0], outs[1]) < tol


@pytest.mark.parametrize("dtype", DTYPES)
def test_wgrad_slabs_and_multi_layer_reduce(F_, dtype):
    """mis_conv3x3_wgrad_slabs + ONE mis_wgrad_reduce_multi over several layers (plain, lazy, two-tensor,
    dilated, with the conv-bias rows as rider) against the one-call entry points: bit-identical."""
    import ctypes
    from medical_image_segmentation_amd import _lib
    cases = [  # B, Ca, Cb, Cout, H, W, dilation, lazy, bias rows
        (2, 8, 0, 32, 16, 32, 1, False, 0), (1, 64, 0, 32, 40, 96, 1, True, 7), (1, 64, 0, 256, 16, 16, 1, False, 0),
        (2, 32, 32, 8, 20, 12, 1, True, 5), (1, 32, 0, 32, 24, 24, 2, False, 3), (1, 16, 0, 64, 18, 34, 3, True, 0),
        (1, 256, 0, 512, 8, 8, 1, False, 40)]
    st = torch.cuda.current_stream().cuda_stream
    code = _lib.BF16 if dtype == torch.bfloat16 else _lib.F32
    cases = [c for c in cases if _lib.query("mis_conv3x3_wgrad_slabs_supported", code, c[0], c[4], c[5], c[1], c[2], c[3], c[6])]
    assert len(cases) >= 4
    jobs = (F_._WgradJob * len(cases))()
    keep, want = [], []
    for k, (B, Ca, Cb, Co, H, W, dil, lazy, nr) in enumerate(cases):
        Ci = Ca + Cb
        xa = act(dg.hash_normalish((B, Ca, H, W), 1200 + k), dtype)
        xb = act(dg.hash_normalish((B, Cb, H, W), 1300 + k), dtype) if Cb else None
        gz = act(dg.hash_normalish((B, Co, H, W), 1400 + k), dtype)
        ssa = lazy_pair(Ca, 1500 + k).to(DEV) if lazy else None
        ssb = lazy_pair(Cb, 1600 + k).to(DEV) if (lazy and Cb) else None
        rows = T(dg.hash_normalish((nr, Co), 1700 + k)).to(DEV) if nr else None
        ref = F_.conv3x3_wgrad_raw(xa, gz, ssa, xb, ssb, (rows, nr) if nr else None, dilation=dil)
        want.append(ref if nr else (ref, None))
        slabs = torch.empty(_lib.query("mis_conv3x3_wgrad_workspace_bytes", B, H, W, Ci, Co), dtype=torch.uint8, device=DEV)
        n = ctypes.c_int(0)
        _lib.call("mis_conv3x3_wgrad_slabs", xa.data_ptr(), F_._p(ssa), F_._p(xb), F_._p(ssb), gz.data_ptr(), F_._dt(xa),
                  B, H, W, Ca, Cb, Co, dil, slabs.data_ptr(), ctypes.addressof(n), st)
        assert n.value >= 1
        dw = torch.full((Co, Ci, 3, 3), float("nan"), device=DEV)
        db = torch.full((Co,), float("nan"), device=DEV) if nr else None
        j = jobs[k]
        j.slabs, j.dw, j.bias_rows, j.dbias = slabs.data_ptr(), dw.data_ptr(), F_._p(rows), F_._p(db)
        j.n_slabs, j.Cin, j.Cout, j.n_bias_rows = n.value, Ci, Co, nr
        keep.append((xa, xb, gz, ssa, ssb, rows, slabs, dw, db))
    _lib.call("mis_wgrad_reduce_multi", ctypes.addressof(jobs), len(cases), st)
    torch.cuda.synchronize()
    for (ref_dw, ref_db), kp in zip(want, keep):
        assert torch.equal(kp[7], ref_dw)
        if ref_db is not None:
            assert torch.equal(kp[8], ref_db)


# ---------------------------------------------------------------------------
# virtual concatenation: conv / data gradient / weight gradient over [xa | xb]
# ---------------------------------------------------------------------------
CAT_SHAPES = [  # B, Ca (= Cb), Cout, H, W
    (1, 32, 8, 16, 16), (2, 64, 32, 20, 12), (1, 32, 64, 16, 16), (1, 128, 16, 8, 8), (1, 64, 128, 16, 32),
    (2, 32, 40, 18, 34)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", CAT_SHAPES)
@pytest.mark.parametrize("lazy", [(False, False), (False, True), (True, True)])
def test_conv3x3_virtual_concat(F_, dtype, shape, lazy):
    """The two-tensor entry points against the one-tensor ones on the materialised concatenation
    (same kernels, same summation order: bitwise equal), and against the float64 oracle."""
    B, Ca, Co, H, W = shape
    xa, xb = dg.hash_normalish((B, Ca, H, W), 901), dg.hash_normalish((B, Ca, H, W), 902)
    w = dg.hash_normalish((Co, 2 * Ca, 3, 3), 903, std=(2.0 / (18 * Ca)) ** 0.5)
    b = dg.hash_uniform((Co,), 904, -0.1, 0.1)
    gz = dg.hash_normalish((B, Co, H, W), 905)
    ssa = lazy_pair(Ca, 910) if lazy[0] else None
    ssb = lazy_pair(Ca, 920) if lazy[1] else None
    xad, xbd, gd = act(xa, dtype), act(xb, dtype), act(gz, dtype)
    if not F_.cat_supported(xad, xbd, Co):
        pytest.skip("shape not taken by the two-tensor kernels in this dtype")
    dev = lambda t: t.to(DEV) if t is not None else None
    # the materialised concatenation the one-tensor path sees: values of both halves, plain
    va = q(lazy_value(xa, ssa, dtype), dtype) if lazy[0] else q(xa, dtype)
    vb = q(lazy_value(xb, ssb, dtype), dtype) if lazy[1] else q(xb, dtype)
    cat = act(torch.cat([va, vb], dim=1), dtype)
    wd, bd = T(w).to(DEV), T(b).to(DEV)
    wf, wt = F_.pack_weights(wd, dtype), F_.pack_weights(wd, dtype, transposed=True)
    z2, st2 = F_.conv3x3_raw(xad, wf, bd, Co, True, dev(ssa), xbd, dev(ssb))
    z1, st1 = F_.conv3x3_raw(cat, wf, bd, Co, True)
    assert torch.equal(z2, z1) and torch.equal(st2, st1)
    da, db = F_.conv3x3_split_raw(gd, wt, Ca, Ca)
    d1, _ = F_.conv3x3_raw(gd, wt, None, 2 * Ca, False)
    assert torch.equal(da, d1[:, :Ca]) and torch.equal(db, d1[:, Ca:])
    dw2 = F_.conv3x3_wgrad_raw(xad, gd, dev(ssa), xbd, dev(ssb))
    dw1 = F_.conv3x3_wgrad_raw(cat, gd)
    assert torch.equal(dw2, dw1)
    # oracle
    xo = torch.cat([va, vb], dim=1).double().requires_grad_(True)
    wo = q(w, dtype).double().requires_grad_(True)
    zo = torch.nn.functional.conv2d(xo, wo, T(b).double(), padding=1)
    zo.backward(q(gz, dtype).double())
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel_err(z2, zo) < tol
    assert rel_err(torch.cat([da, db], dim=1), xo.grad) < tol
    assert rel_err(dw2, wo.grad) < (2e-5 if dtype == torch.float32 else 3e-3)


def test_double_conv_two_inputs_matches_concat(F_):
    """DoubleConv over (up, skip) as two tensors == DoubleConv over their concatenation, forward
    and every gradient (float32, the Up block of unet.py:66-68)."""
    from medical_image_segmentation_amd.model.utils.modules import DoubleConv
    B, Ca, Co, H, W = 2, 32, 32, 16, 16
    xa, xb = dg.hash_normalish((B, Ca, H, W), 931), dg.hash_normalish((B, Ca, H, W), 932)
    gy = dg.hash_normalish((B, Co, H, W), 933)
    ssb = lazy_pair(Ca, 934).to(DEV)
    m = DoubleConv(2 * Ca, Co).to(DEV).train()
    xad, xbd = act(xa).requires_grad_(True), act(xb).requires_grad_(True)
    assert F_.cat_supported(xad, xbd, m.cbr1[0].out_channels)
    y = F_.materialize(*m.forward_lazy(xad, None, xbd, ssb))
    y.backward(act(gy))
    g2 = [p.grad.clone() for p in m.parameters()]
    m.zero_grad()
    xa1 = act(xa).requires_grad_(True)
    vb = F_.materialize(act(xb), ssb).detach().requires_grad_(True)
    for bn in (m.cbr1[1], m.cbr2[1]):
        bn.reset_running_stats()
    y1 = m(torch.cat([xa1, vb], dim=1).contiguous(memory_format=torch.channels_last))
    y1.backward(act(gy))
    assert rel_err(y, y1) < 1e-6
    assert rel_err(xad.grad, xa1.grad) < 1e-5
    assert rel_err(xbd.grad, vb.grad) < 1e-5      # gradient with respect to the skip's VALUE
    for a, b_ in zip(g2, [p.grad for p in m.parameters()]):
        assert rel_err(a, b_) < 1e-5


# ---------------------------------------------------------------------------
# BatchNorm-backward sums produced by the kernel that writes the gradient
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("consumer", ["pool", "conv1x1"])
def test_bn_sums_fused_into_gradient_producer(F_, dtype, consumer):
    """conv+BN+ReLU followed by max-pool(+skip) or the 1x1 head: with the fusion on, the consumer's
    backward kernel emits (sum dy, sum dy*xhat) and the layer's own reduction pass is skipped; every
    gradient must equal the unfused run's (same arithmetic, different summation order)."""
    B, Ci, Co, H, W = 2, 8, 16, 16, 32
    x = np.maximum(dg.hash_normalish((B, Ci, H, W), 1401), 0)
    conv, bn = _hash_cbr(Ci, Co, 1410)
    head = torch.nn.Conv2d(Co, 4, 1)
    with torch.no_grad():
        head.weight.copy_(T(dg.hash_normalish((4, Co, 1, 1), 1420, std=0.3)))
        head.bias.copy_(T(dg.hash_uniform((4,), 1421, -0.1, 0.1)))
    head = head.to(DEV)
    gy = dg.hash_normalish((B, Co, H // 2, W // 2), 1402)
    gs = dg.hash_normalish((B, Co, H, W), 1403, std=0.3)
    gl = dg.hash_normalish((B, 4, H, W), 1404)
    res, calls = [], []
    orig = F_.call

    def spy(name, *a):
        calls.append(name)
        return orig(name, *a)

    min_numel = F_.BN_SUMS_MIN_NUMEL
    for fuse in (True, False):
        F_.FUSE_BN_SUMS = fuse
        F_.BN_SUMS_MIN_NUMEL = 0          # the production threshold only selects large tensors
        F_.call = spy
        calls.clear()
        try:
            for m in (conv, bn, head):
                m.zero_grad()
            xd = act(x, dtype).requires_grad_(True)
            z, ss = F_.conv_bn_relu(xd, None, conv, bn, True)
            if consumer == "pool":
                y, skip = F_.PoolSkipFn.apply(z, ss)
                torch.autograd.backward([y, skip], [act(gy, dtype), act(gs, dtype)])
            else:
                F_.Conv1x1Fn.apply(z, ss, head.weight, head.bias).backward(act(gl, dtype))
        finally:
            F_.call = orig
            F_.FUSE_BN_SUMS = True
            F_.BN_SUMS_MIN_NUMEL = min_numel
        assert ("mis_bn_relu_bwd_reduce" in calls) == (not fuse)
        res.append([xd.grad.clone(), conv.weight.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone()])
    # The two runs sum in different orders, so float32 results differ in their last bits and a
    # bfloat16-stored value may round the other way: the stored tensor is bounded in bf16 steps
    # (one step = 2^-7 = 3.9e-3 relative), the float32 ones relative to their scale.
    if dtype == torch.bfloat16:
        assert bf16_ulps(res[0][0], res[1][0]) <= 2.0
    for a, b in zip(*res):
        assert rel_err(a, b) < (2e-5 if dtype == torch.float32 else 8e-3)


def _hash_cbr(Ci, Co, seed):
    """Conv2d(3x3)+BatchNorm2d with hash-generated parameters: test inputs never depend on torch's
    global generator (i.e. on which tests ran before)."""
    conv = torch.nn.Conv2d(Ci, Co, 3, padding=1)
    bn = torch.nn.BatchNorm2d(Co)
    with torch.no_grad():
        conv.weight.copy_(T(dg.hash_normalish((Co, Ci, 3, 3), seed, std=(2.0 / (9 * Co)) ** 0.5)))
        conv.bias.copy_(T(dg.hash_uniform((Co,), seed + 1, -0.1, 0.1)))
        bn.weight.copy_(T(dg.hash_uniform((Co,), seed + 2, 0.5, 1.5)))
        bn.bias.copy_(T(dg.hash_uniform((Co,), seed + 3, -0.2, 0.2)))
    return conv.to(DEV), bn.to(DEV)


def bf16_ulps(a, b):
    """Largest difference between two bfloat16-stored tensors in units of one bf16 step (2^-7
    relative) at the larger operand's binade."""
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    scale = torch.maximum(a.abs(), b.abs()).clamp_min(1e-30)
    ulp = torch.exp2(torch.floor(torch.log2(scale)) - 7)
    return float(((a - b).abs() / ulp).max())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n_consumers", [2, 3])
def test_bn_sums_tag_rejected_when_gradients_accumulate(F_, dtype, n_consumers):
    """A lazy activation (z, ss with batch statistics) read by several conv+BN+ReLU nodes: each
    consumer's data-gradient kernel tags its output with BatchNorm-backward sums of ITS gradient
    only, and autograd adds the other consumers' gradients into the first tensor in place.  The
    producer must then ignore the tag and reduce over the summed gradient: every parameter
    gradient equals the run with the fusion switched off."""
    B, Ci, Cm, Co, H, W = 2, 8, 16, 32, 16, 24
    x = np.maximum(dg.hash_normalish((B, Ci, H, W), 1601), 0)
    prod = _hash_cbr(Ci, Cm, 1610)
    cons = [_hash_cbr(Cm, Co, 1620 + 10 * k) for k in range(n_consumers)]
    gys = [dg.hash_normalish((B, Co, H, W), 1650 + k) for k in range(n_consumers)]
    res = []
    for fuse in (True, False):
        F_.FUSE_BN_SUMS = fuse
        for conv, bn in [prod] + cons:
            conv.zero_grad(); bn.zero_grad()
            bn.reset_running_stats()
        xd = act(x, dtype).requires_grad_(True)
        z, ss = F_.conv_bn_relu(xd, None, prod[0], prod[1], True)
        outs = [F_.conv_bn_relu(z, ss, c, b, True)[0] for c, b in cons]
        torch.autograd.backward(outs, [act(g, dtype) for g in gys])
        res.append([xd.grad.float().clone()] + [p.grad.clone() for m in prod for p in m.parameters()])
    names = ["dx", "conv.weight", "conv.bias", "bn.weight", "bn.bias"]
    # fp64 reference of the producer's BatchNorm gradients: the sums over ALL consumers' gradients
    xo = q(x, dtype).double().requires_grad_(True)
    po = [p.detach().double().cpu().requires_grad_(True) for m in prod for p in m.parameters()]
    zo = torch.nn.functional.conv2d(xo, po[0], po[1], padding=1)
    ao = torch.relu(torch.nn.functional.batch_norm(zo, None, None, po[2], po[3], True, 0.1, 1e-5))
    outs_o = []
    for (c, b), g in zip(cons, gys):
        w_, b_, ga_, be_ = [p.detach().double().cpu() for m in (c, b) for p in m.parameters()]
        zc = torch.nn.functional.conv2d(ao, w_, b_, padding=1)
        outs_o.append(torch.relu(torch.nn.functional.batch_norm(zc, None, None, ga_, be_, True, 0.1, 1e-5)))
    torch.autograd.backward(outs_o, [q(g, dtype).double() for g in gys])
    for k, name in enumerate(names):
        if name == "conv.bias":
            continue                 # analytically zero under train-mode BatchNorm
        assert rel_err(res[0][k], res[1][k]) < (5e-5 if dtype == torch.float32 else 8e-3), name
        ref = (xo.grad if k == 0 else po[k - 1].grad).double()
        got = res[0][k].double().cpu()
        if dtype == torch.float32:
            assert rel_err(got, ref) < 2e-4, name
        else:
            # two bfloat16 layers deep (stored z, stored gradients): the float64 reference rounds none of
            # them, so only the overall (L2) agreement is bounded; sums of ONE consumer would be off by O(1)
            l2 = float((got - ref).norm() / ref.norm())
            print(f"{name}: bf16 L2 error vs float64 reference {l2:.3e}")
            assert l2 < 0.1, name


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("ch", [(8, 16, 32), (16, 64, 64), (32, 8, 16), (64, 128, 64)])
def test_bn_sums_fused_into_data_gradient(F_, dtype, ch):
    """DoubleConv backward with FUSE_DGRAD_BN_SUMS (off by default: slower on the full step): the second
    conv's data-gradient kernel also accumulates the first BatchNorm's backward sums (its output is that
    layer's `da`, and z of that layer is read in the epilogue); with the fusion off the first layer runs its
    own reduction pass.  All gradients must agree between the two runs (same arithmetic, different summation
    order)."""
    Ci, Cm, Co = ch
    B, H, W = 2, 24, 40
    from medical_image_segmentation_amd.model.utils.modules import DoubleConv
    torch.manual_seed(5)
    blk = DoubleConv(Ci, Co, Cm).to(DEV).train()
    with torch.no_grad():
        for bn in (blk.cbr1[1], blk.cbr2[1]):
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
    x = dg.hash_normalish((B, Ci, H, W), 1501)
    gy = dg.hash_normalish((B, Co, H, W), 1502)
    res, calls = [], []
    orig = F_.call

    def spy(name, *a):
        calls.append(name)
        return orig(name, *a)

    for fuse in (True, False):
        F_.FUSE_DGRAD_BN_SUMS = fuse
        F_.call = spy
        calls.clear()
        try:
            blk.zero_grad()
            xd = act(x, dtype).requires_grad_(True)
            z, ss = blk.forward_lazy(xd)
            z.backward(act(gy, dtype))
        finally:
            F_.call = orig
            F_.FUSE_DGRAD_BN_SUMS = False
        assert calls.count("mis_bn_relu_bwd_reduce") == (1 if fuse else 2)
        assert ("mis_conv3x3_dgrad_bn_sums" in calls) == fuse
        res.append([xd.grad.float().clone()] + [p.grad.clone() for p in blk.parameters()])
    tol = 5e-5 if dtype == torch.float32 else 5e-3
    names = ["dx"] + [k for k, _ in blk.named_parameters()]
    for k, a, b in zip(names, *res):
        if k.endswith(".0.bias"):
            continue                     # analytically zero under train-mode BatchNorm: round-off only
        assert rel_err(a, b) < tol, k


# ---------------------------------------------------------------------------
# Dropout2d folded into the Down / Up kernels (unet.py:27-42, 49-74)
# ---------------------------------------------------------------------------
def _noise(B, C, p, seed):
    g = torch.Generator().manual_seed(seed)
    return ((torch.rand(B, C, 1, 1, generator=g) >= p).float() / (1.0 - p))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("lazy", [False, True])
def test_maxpool_dropout(F_, dtype, lazy):
    B, C, H, W = 3, 16, 12, 20
    g = torch.Generator().manual_seed(17)
    x = q(torch.randn(B, C, H, W, generator=g), dtype)
    gy = q(torch.randn(B, C, H // 2, W // 2, generator=g), dtype)
    nz = _noise(B, C, 0.4, 3)
    ss = None
    if lazy:
        ss = torch.stack([torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3])
    xd = act(x, dtype).requires_grad_(True)
    y = F_.MaxPool2x2Fn.apply(xd, None if ss is None else ss.to(DEV), nz.to(DEV))
    y.backward(act(gy, dtype))
    xr = x.clone().double().requires_grad_(True)
    a = xr if ss is None else torch.relu(xr * ss[0].double()[None, :, None, None] + ss[1].double()[None, :, None, None])
    if lazy and dtype == torch.bfloat16:
        a = a + (a.detach().to(dtype).double() - a.detach())          # the activation as the kernel rounds it
    yr = torch.nn.functional.max_pool2d(a, 2, 2) * nz.double()
    yr.backward(gy.double())
    assert rel_err(y, yr) < TOL[dtype]
    if not lazy:
        assert rel_err(xd.grad, xr.grad) < TOL[dtype]
    # dropped feature maps are exactly zero, kept ones exactly scaled
    dropped = (nz[:, :, 0, 0] == 0)
    assert float(y.detach().float().cpu()[dropped].abs().max()) == 0.0
    assert float(xd.grad.float().cpu()[dropped].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C", [16, 64, 6])
def test_upsample_dropout(F_, dtype, C):
    B, h, w = 2, 5, 7
    g = torch.Generator().manual_seed(19)
    low = q(torch.randn(B, C, h, w, generator=g), dtype)
    gy = q(torch.randn(B, C, 2 * h, 2 * w, generator=g), dtype)
    nz = _noise(B, C, 0.3, 5)
    ld = act(low, dtype).requires_grad_(True)
    y = F_.UpsampleFn.apply(ld, None, nz.to(DEV))
    y.backward(act(gy, dtype))
    lr = low.clone().double().requires_grad_(True)
    yr = orc.upsample2x_bilinear(lr * nz.double())
    yr.backward(gy.double())
    assert rel_err(y, yr) < TOL[dtype]
    assert rel_err(ld.grad, lr.grad) < TOL[dtype]
    # and as the head of a materialised concatenation
    skip = q(torch.randn(B, C, 2 * h, 2 * w, generator=g), dtype)
    yc = F_.UpsampleCatFn.apply(act(low, dtype), None, act(skip, dtype), None, nz.to(DEV))
    assert rel_err(yc[:, :C], yr) < TOL[dtype]
    assert torch.equal(yc[:, C:].cpu(), skip.to(dtype))


def test_dropout_noise_is_torch_feature_dropout(F_):
    """The multiplier is drawn exactly as nn.Dropout2d draws it on this device: same seed, same
    dropped feature maps."""
    for p in (0.03, 0.5, 0.9):
        layer = torch.nn.Dropout2d(p).train()
        x = torch.ones(5, 37, 4, 4, device=DEV)
        torch.manual_seed(123)
        ref = layer(x)
        torch.manual_seed(123)
        nz = F_.dropout2d_noise(layer, x)
        assert nz.shape == (5, 37, 1, 1) and nz.dtype == torch.float32
        assert torch.equal(ref, x * nz)
        assert F_.dropout2d_noise(layer.eval(), x) is None
    assert F_.dropout2d_noise(torch.nn.Dropout2d(0.0).train(), x) is None


def test_dropout_noise_counter_based_generator(F_):
    """mis_dropout2d_noise against the CPU restatement of Philox4x32-10 (pinned by the published
    known-answer vectors in tests/test_oracle_golden.py): bit-exact multipliers for several (seed, offset,
    shape, p); an element depends on (seed, offset, b, c) only; the keep rate is 1-p; and through
    functional.dropout2d_noise the torch seed reproduces a run while successive calls differ."""
    for seed, offset, B, C, p in [(0, 0, 1, 8, 0.5), (1234, 4, 16, 256, 0.3), ((1 << 63) + 12345, (1 << 40) + 8, 5, 37, 0.03),
                                  (7, 12, 3, 512, 0.9), (7, 12, 2, 16, 1.0), (7, 12, 2, 16, 0.0)]:
        nz = F_.dropout2d_noise_philox(B, C, p, DEV, seed=seed, offset=offset)
        ref = orc.dropout2d_noise_philox(seed, offset, B, C, p)
        assert nz.shape == (B, C, 1, 1) and nz.dtype == torch.float32
        assert np.array_equal(nz.cpu().numpy().reshape(B, C), ref), (seed, offset, B, C, p)
    # the first rows of a larger call are the smaller call (no dependence on the launch shape)
    a = F_.dropout2d_noise_philox(4, 64, 0.4, DEV, seed=99, offset=0).flatten()
    b = F_.dropout2d_noise_philox(64, 64, 0.4, DEV, seed=99, offset=0).flatten()
    assert torch.equal(a, b[:a.numel()])
    big = F_.dropout2d_noise_philox(64, 1024, 0.3, DEV, seed=5, offset=0)
    assert abs(float((big > 0).float().mean()) - 0.7) < 0.01
    assert float(big.max()) == float(np.float32(1.0) / (np.float32(1.0) - np.float32(0.3)))
    # driven by torch's generator: same seed -> same multipliers, next call -> others
    F_.DROPOUT_RNG = "philox"
    try:
        layer = torch.nn.Dropout2d(0.5).train()
        x = torch.ones(8, 128, 2, 2, device=DEV)
        torch.manual_seed(321)
        n1, n2 = F_.dropout2d_noise(layer, x), F_.dropout2d_noise(layer, x)
        torch.manual_seed(321)
        m1 = F_.dropout2d_noise(layer, x)
        assert torch.equal(n1, m1) and not torch.equal(n1, n2)
        assert F_.dropout2d_noise(layer.eval(), x) is None
    finally:
        F_.DROPOUT_RNG = "torch"
