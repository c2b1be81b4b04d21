"""GPU suite: RDAM_UNet (model/rdam_unet.py of the reference) and its two blocks — ResDConv
(dilated 3x3 convs + 1x1 shortcut, modules.py:426-453) and MDAM (attention.py:147-190) — through the
HIP path against the golden vectors the reference itself produced (tests/golden/rdam.npz), plus the
dilated conv kernels against torch's conv on the CPU.

float32 mode: block outputs 1e-4 of their scale (MDAM 1e-5), input gradients 1e-3, parameter
gradients 2e-3 of each tensor's scale; whole network: logits 1e-3 of their scale, losses 1e-5,
gradient norms 5 %.  bfloat16 mode: loss within 2e-2."""
import numpy as np
import pytest
import torch

import make_golden as mg
from oracle import datagen as dg
from oracle import unet_oracle as orc

pytestmark = pytest.mark.gpu
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
DEV = "cuda:0"


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu() if torch.is_tensor(b) else T(np.asarray(b)).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def state_from_golden(g, name, seed):
    spec = [(str(k), tuple(int(v) for v in str(s).split(",") if v)) for k, s in zip(g[f"{name}.keys"], g[f"{name}.shapes"])]
    return orc.hash_state(spec, seed)


def act(a, dtype=torch.float32):
    t = a if torch.is_tensor(a) else T(a)
    return t.to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("dil", [2, 3])
@pytest.mark.parametrize("shape", [(2, 16, 16, 20, 12), (1, 32, 128, 9, 37), (1, 64, 32, 16, 16), (2, 8, 24, 5, 5),
                                   (1, 256, 128, 8, 8)])
@pytest.mark.parametrize("lazy", [False, True])
def test_dilated_conv_kernels(dtype, dil, shape, lazy):
    """mis_conv3x3_dilated_fwd (forward and, on the transposed image, data gradient) and
    mis_conv3x3_dilated_wgrad against torch's dilated conv in float64 on the same (dtype-rounded) values."""
    from medical_image_segmentation_amd import functional as F_
    B, Ci, Co, H, W = shape
    if not F_.query("mis_conv3x3_dilated_supported", F_._DT[dtype], B, H, W, Ci, Co, dil):
        pytest.skip("shape not taken by the matrix-core kernels in this dtype")
    q = lambda a: T(a).to(dtype).double()
    x = dg.hash_normalish((B, Ci, H, W), 2101)
    w = dg.hash_normalish((Co, Ci, 3, 3), 2102, std=(2.0 / (9 * Co)) ** 0.5)
    b = dg.hash_uniform((Co,), 2103, -0.1, 0.1)
    gz = dg.hash_normalish((B, Co, H, W), 2104)
    ss = None
    xv = q(x)
    if lazy:
        sc = dg.hash_uniform((Ci,), 2105, 0.4, 1.6) * np.where(dg.hash_uniform((Ci,), 2106, 0, 1) < 0.25, -1.0, 1.0)
        ss = T(np.stack([sc, dg.hash_uniform((Ci,), 2107, -0.5, 0.5)]).astype(np.float32))
        xv = torch.relu(q(x) * ss[0].double()[None, :, None, None] + ss[1].double()[None, :, None, None]).float()
        xv = xv.to(dtype).double()
    wd_, bd_ = T(w).to(DEV), T(b).to(DEV)
    xd = act(x, dtype)
    ssd = ss.to(DEV) if lazy else None
    wf, wt = F_.pack_weights_pair(wd_, dtype)
    z, _ = F_.conv3x3_raw(xd, wf, bd_, Co, False, ssd, dilation=dil)
    wo = q(w) if dtype == torch.bfloat16 else T(w).double()
    zo = torch.nn.functional.conv2d(xv, wo, T(b).double(), padding=dil, dilation=dil)
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    assert rel_err(z, zo) < tol
    # data gradient: the same kernel on the transposed weight image
    gzd = act(gz, dtype)
    dx, _ = F_.conv3x3_raw(gzd, wt, None, Ci, False, dilation=dil)
    dxo = torch.nn.grad.conv2d_input((B, Ci, H, W), wo, q(gz), padding=dil, dilation=dil)
    assert rel_err(dx, dxo) < tol
    dw = F_.conv3x3_wgrad_raw(xd, gzd, ssd, dilation=dil)
    dwo = torch.nn.grad.conv2d_weight(xv, (Co, Ci, 3, 3), q(gz), padding=dil, dilation=dil)
    assert rel_err(dw, dwo) < (2e-4 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 3, 32, 21, 70), (1, 3, 16, 16, 64), (2, 3, 24, 9, 9), (1, 3, 8, 33, 130),
                                   (1, 1, 32, 17, 40), (2, 1, 16, 8, 8), (1, 3, 40, 12, 12)])
def test_bare_conv_image_facing(dtype, shape):
    """nn.Conv2d(Cin, Cout, 3, padding=1) with nothing after it on the image (RDAM_UNet.encoder1,
    model/rdam_unet.py:21): 1 or 3 input channels into 1..5 slices of 8 output channels — all slices from one
    workgroup, the strip kernel for one input channel, and the per-slice form — against torch's conv2d
    (float32: 2e-5 / 2e-4; bfloat16: 1e-2 / 2e-2 of each tensor's scale)."""
    from medical_image_segmentation_amd import functional as F_
    B, Ci, Co, H, W = shape
    g = torch.Generator().manual_seed(Ci * 100 + Co)
    conv = torch.nn.Conv2d(Ci, Co, 3, padding=1).to(DEV)
    x0 = torch.randn(B, Ci, H, W, generator=g).to(DEV)
    gy = torch.randn(B, Co, H, W, generator=g).to(DEV)
    x = x0.clone().to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    z = F_.Conv3x3Fn.apply(x, None, None, None, conv.weight, conv.bias)
    z.backward(gy.to(dtype).contiguous(memory_format=torch.channels_last))
    got = (z.detach().float(), x.grad.float(), conv.weight.grad.clone(), conv.bias.grad.clone())
    conv.zero_grad()
    xr = x0.to(dtype).double().requires_grad_(True)
    wq = conv.weight.detach().to(dtype).double().requires_grad_(True)       # the kernels multiply weights stored in `dtype`
    bq = conv.bias.detach().double().requires_grad_(True)
    zr = torch.nn.functional.conv2d(xr, wq, bq, padding=1)
    zr.backward(gy.to(dtype).double())
    tol, gtol = (2e-5, 2e-4) if dtype == torch.float32 else (1e-2, 2e-2)
    assert rel_err(got[0], zr) < tol
    assert rel_err(got[1], xr.grad) < gtol
    assert rel_err(got[2], wq.grad) < gtol
    assert rel_err(got[3], bq.grad) < gtol


@pytest.mark.parametrize("name", list(mg.rdam_block_cases()))
def test_rdam_blocks_golden(golden, name):
    from medical_image_segmentation_amd.model.utils.attention import MDAM
    from medical_image_segmentation_amd.model.utils.modules import ResDConv
    g = golden("rdam")
    kind, args, shape, seed = mg.rdam_block_cases()[name]
    st = state_from_golden(g, name, seed)
    blk = ResDConv(*args) if kind == "resd" else MDAM(*args)
    assert list(blk.state_dict().keys()) == list(st.keys())
    blk.load_state_dict(st)
    blk = blk.to(DEV).train()
    x = dg.hash_normalish(shape, seed + 100)
    x = act(np.maximum(x, 0) if kind == "resd" else x).requires_grad_(True)
    y = blk(x)
    gy = act(dg.hash_normalish(tuple(y.shape), seed + 200))
    y.backward(gy)
    assert rel_err(y, g[f"{name}.y"]) < (1e-4 if kind == "resd" else 1e-5)
    assert rel_err(x.grad, g[f"{name}.dx"]) < 1e-3
    scale = max(float(np.abs(g[f"{name}.grad.{k}"]).max()) for k, _ in blk.named_parameters())
    for k, p in blk.named_parameters():
        ref = g[f"{name}.grad.{k}"]
        if (kind == "resd" and k.endswith(".0.bias")) or k in ("conv1x1.weight", "conv1x1.bias", "groupNorm.weight"):
            # analytically zero (conv bias in front of a train-mode BatchNorm; MDAM's strip-pooling
            # branch, whose GroupNorm -> global average is the GroupNorm bias): round-off in the fixture
            assert float(p.grad.abs().max()) < 1e-4 * scale and float(np.abs(ref).max()) < 1e-4 * scale, k
            continue
        assert float((p.grad.double().cpu() - T(ref).double()).abs().max()) < 2e-3 * max(float(np.abs(ref).max()),
                                                                                        1e-2 * scale), k
    for k, b_ in blk.named_buffers():
        assert rel_err(b_.float(), g[f"{name}.buf.{k}"]) < 1e-4, k


@pytest.mark.parametrize("name", list(mg.rdam_cases()))
def test_rdam_unet_fp32_golden(golden, report, name):
    from medical_image_segmentation_amd.model.rdam_unet import RDAM_UNet
    from medical_image_segmentation_amd.utils.loss_fn import diceloss
    g = golden("rdam")
    c = mg.rdam_cases()[name]
    st = state_from_golden(g, name, c["seed"])
    m = RDAM_UNet(c["cin"], c["ncls"], 0, base_channels=c["base"], compute_dtype=torch.float32)
    assert list(m.state_dict().keys()) == list(st.keys())
    m.load_state_dict(st)
    m = m.to(DEV).train()
    x = T(dg.hash_normalish((c["B"], c["cin"], c["hw"], c["hw"]), c["seed"] + 100)).to(DEV)
    t = T(dg.blob_mask(c["B"], c["hw"], c["hw"], c["ncls"], c["seed"] + 300)).to(DEV)
    names = [f"c{i}" for i in range(1, c["ncls"])]
    logits = m(x)
    ld = diceloss(names)(logits, t)
    ld["total_loss"].backward()
    report("fp32 logits max error / max |logit|", rel_err(logits, g[f"{name}.logits"]))
    assert rel_err(logits, g[f"{name}.logits"]) < 1e-3
    np.testing.assert_allclose([float(v.detach()) for v in ld.values()], g[f"{name}.loss"], rtol=1e-5, atol=1e-5)
    params = dict(m.named_parameters())
    keys = [str(k) for k in g[f"{name}.grad_keys"]]
    assert keys == list(params.keys())
    assert all(p.grad is not None for p in params.values())
    norms = np.array([float(params[k].grad.double().norm()) for k in keys])
    ref = g[f"{name}.grad_norms"]
    big = ref > 1e-3 * ref.max()
    dev = float(np.abs(norms[big] - ref[big]).max() / ref.max())
    report("fp32 gradient norm deviation / largest norm", dev)
    assert dev < 5e-2
    assert rel_err(m.encoder3.cbr3[1].running_mean, g[f"{name}.running_mean.encoder3.cbr3"]) < 1e-3
    m.eval()
    with torch.no_grad():
        assert rel_err(m(x), g[f"{name}.eval_logits"]) < 1e-3


def test_rdam_unet_bf16_and_dropout(golden, report):
    from medical_image_segmentation_amd.model.rdam_unet import RDAM_UNet
    from medical_image_segmentation_amd.utils.loss_fn import diceloss
    g = golden("rdam")
    name = "rdam_a"
    c = mg.rdam_cases()[name]
    st = state_from_golden(g, name, c["seed"])
    m = RDAM_UNet(c["cin"], c["ncls"], 0, base_channels=c["base"], compute_dtype=torch.bfloat16)
    m.load_state_dict(st)
    m = m.to(DEV).train()
    x = T(dg.hash_normalish((c["B"], c["cin"], c["hw"], c["hw"]), c["seed"] + 100)).to(DEV)
    t = T(dg.blob_mask(c["B"], c["hw"], c["hw"], c["ncls"], c["seed"] + 300)).to(DEV)
    names = [f"c{i}" for i in range(1, c["ncls"])]
    logits = m(x)
    assert logits.dtype == torch.bfloat16
    ld = diceloss(names)(logits, t)
    ld["total_loss"].backward()
    report("bf16 total loss difference vs fp32 reference", abs(float(ld["total_loss"].detach()) - g[f"{name}.loss"][-1]))
    assert abs(float(ld["total_loss"].detach()) - g[f"{name}.loss"][-1]) < 2e-2
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    m2 = RDAM_UNet(c["cin"], c["ncls"], 0.5, base_channels=c["base"], compute_dtype=torch.float32).to(DEV).train()
    y = m2(x)
    assert torch.isfinite(y).all()


@pytest.mark.parametrize("which", ["a_unet", "rdam_unet"])
def test_attention_variants_full_size_step_properties(which):
    """BASELINE configs[3] at its real image size (512 x 512, 3 channels, base 32, batch 8) — beyond what the CPU
    oracle finishes in seconds, so checked through size-independent properties of one training step: two bfloat16
    runs from the same state are bit-identical (no floating-point atomics on the path, the gradients of the
    four-consumer AMSFN inputs included), the confusion counts reproduce the label histogram, bfloat16 and float32
    modes agree on the loss within the storage format's bound, and AdamW steps lower the loss."""
    from medical_image_segmentation_amd.model.a_unet import A_UNet
    from medical_image_segmentation_amd.model.rdam_unet import RDAM_UNet
    from medical_image_segmentation_amd.utils.loss_fn import diceloss
    from medical_image_segmentation_amd.utils.metrics import Metrics
    from medical_image_segmentation_amd.utils.optim import clip_and_step
    net = A_UNet if which == "a_unet" else RDAM_UNet
    B, S = 8, 512
    g = torch.Generator().manual_seed(91)
    x = torch.randn(B, 3, S, S, generator=g).to(DEV)
    y = torch.randint(0, 4, (B, S, S), generator=g).to(DEV)
    torch.manual_seed(17)
    state = {k: v.clone() for k, v in net(3, 4, 0, base_channels=32).state_dict().items()}
    loss_fn, metric = diceloss(["a", "b", "c"]), Metrics(["a", "b", "c"])

    def run(dtype, steps=1):
        m = net(3, 4, 0, base_channels=32, compute_dtype=dtype)
        m.load_state_dict(state)
        m = m.to(DEV).train()
        opt = torch.optim.AdamW(m.parameters(), lr=2e-3)
        out = []
        for _ in range(steps):
            opt.zero_grad()
            logits = m(x)
            ld = loss_fn(logits, y)
            cm, status = metric.confusion_counts(logits, y)
            ld["total_loss"].backward()
            out.append((logits.detach().float().clone(), float(ld["total_loss"].detach()), cm.clone(),
                        [p.grad.clone() for p in m.parameters() if p.grad is not None]))
            clip_and_step(opt, [p for p in m.parameters() if p.grad is not None], 1.0)
        assert int(status) == 0
        return out

    a, b = run(torch.bfloat16)[0], run(torch.bfloat16)[0]
    assert torch.equal(a[0], b[0]) and a[1] == b[1] and torch.equal(a[2], b[2])
    assert all(torch.equal(p, q) for p, q in zip(a[3], b[3]))
    assert all(torch.isfinite(p).all() for p in a[3])
    assert int(a[2].sum()) == B * S * S
    assert torch.equal(a[2].sum(dim=2).sum(dim=0).cpu(), torch.bincount(y.reshape(-1).cpu(), minlength=4))
    f = run(torch.float32, steps=3)
    assert abs(a[1] - f[0][1]) < 2e-2
    assert f[2][1] < f[0][1]


@pytest.mark.parametrize("which", ["a_unet", "rdam_unet"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_batched_weight_preparation_equals_per_layer(which, dtype):
    """All DWConv compositions, 1x1-as-3x3 weights and operand images prepared by two launches at the top of the
    forward (functional.ComposedWeights) against one or two launches per layer: three optimizer steps from the same
    start give bit-identical losses, gradients and parameters (the expressions are the same; the plan must notice
    every parameter update) — with and without the deferral of the composed weights' gradients to the end of the
    backward pass (functional.DEFER_COMPOSED)."""
    from medical_image_segmentation_amd import functional as F_
    from medical_image_segmentation_amd.model.a_unet import A_UNet
    from medical_image_segmentation_amd.model.rdam_unet import RDAM_UNet
    from medical_image_segmentation_amd.utils.loss_fn import diceloss
    from medical_image_segmentation_amd.utils.optim import clip_and_step
    net = A_UNet if which == "a_unet" else RDAM_UNet
    torch.manual_seed(3)
    x = torch.randn(2, 3, 32, 48, device=DEV)
    y = torch.randint(0, 4, (2, 32, 48), device=DEV)
    out = {}
    for batched in (True, "no deferral", False):
        F_.BATCH_WEIGHT_PREP = bool(batched)
        F_.DEFER_COMPOSED = batched is True
        torch.manual_seed(11)
        m = net(3, 4, 0, base_channels=32, compute_dtype=dtype).to(DEV).train()
        opt = torch.optim.AdamW(m.parameters(), lr=1e-2)
        params = list(m.parameters())
        losses, calls = [], []
        orig = F_.call

        def spy(name, *a):
            calls.append(name)
            return orig(name, *a)

        F_.call = spy
        try:
            for _ in range(3):
                opt.zero_grad()
                loss = diceloss(["a", "b", "c"])(m(x), y)["total_loss"]
                loss.backward()
                losses.append(float(loss.detach()))
                clip_and_step(opt, params, 1.0)
        finally:
            F_.call = orig
        assert ("mis_dwconv_compose_multi" in calls) == bool(batched)
        assert (calls.count("mis_dwconv_compose_fwd") == 0) == bool(batched)
        # the gradients of the composed weights wait for the end of the pass: one chain-rule launch per 24 layers
        assert ("mis_dwconv_compose_bwd_multi" in calls) == (batched is True)
        assert (calls.count("mis_dwconv_compose_bwd") == 0) == (batched is True)
        out[batched] = (losses, [p.detach().clone() for p in params], [p.grad.clone() for p in params])
    F_.BATCH_WEIGHT_PREP = F_.DEFER_COMPOSED = True
    for other in ("no deferral", False):
        assert out[True][0] == out[other][0]
        for a, b in zip(out[True][1], out[other][1]):
            assert torch.equal(a, b)
        for a, b in zip(out[True][2], out[other][2]):
            assert torch.equal(a, b)
