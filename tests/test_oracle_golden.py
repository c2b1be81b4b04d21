"""CPU suite: the oracle (oracle/unet_oracle.py) against the golden vectors that
tests/golden/make_golden.py produced by running the reference itself."""
import numpy as np
import pytest
import torch

import make_golden as mg
from oracle import datagen as dg
from oracle import unet_oracle as orc

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def close(a, b, rtol=1e-5, atol=1e-6):
    np.testing.assert_allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), rtol=rtol, atol=atol)


@pytest.mark.parametrize("name,c", [("dc_a", dict(B=2, cin=3, mid=8, cout=8, hw=16, seed=11)),
                                    ("dc_b", dict(B=2, cin=16, mid=4, cout=16, hw=8, seed=12))])
def test_double_conv(golden, name, c):
    g = golden("double_conv")
    st = mg.orc_double_conv_state(c["cin"], c["mid"], c["cout"], c["seed"])
    pk = [k for k in st if k.endswith(orc.PARAM_SUFFIXES)]
    for k in pk:
        st[k].requires_grad_(True)
    x = T(dg.hash_normalish((c["B"], c["cin"], c["hw"], c["hw"]), c["seed"] + 100)).requires_grad_(True)
    gy = T(dg.hash_normalish((c["B"], c["cout"], c["hw"], c["hw"]), c["seed"] + 200))
    y = orc.double_conv(x, st, "blk", training=True)
    y.backward(gy)
    close(y.detach(), g[f"{name}.train_y"], 1e-4, 1e-5)
    close(x.grad, g[f"{name}.dx"], 1e-4, 1e-5)
    for k in pk:
        # the conv-bias gradient is analytically zero under train-mode BN: only round-off remains
        atol = 1e-4 if k.endswith(".0.bias") else 2e-5
        close(st[k].grad, g[f"{name}.grad.{k[4:]}"], 2e-4, atol)
    for k in st:
        if "running" in k or "num_batches" in k:
            close(st[k].detach(), g[f"{name}.buf.{k[4:]}"], 1e-5, 1e-6)
    with torch.no_grad():
        ye = orc.double_conv(x, st, "blk", training=False)
    close(ye, g[f"{name}.eval_y"], 1e-4, 1e-5)


@pytest.mark.parametrize("name,c", [("unet_a", dict(B=2, cin=1, ncls=4, hw=32, seed=21)),
                                    ("unet_b", dict(B=2, cin=3, ncls=4, hw=64, seed=22)),
                                    ("unet_c", dict(B=1, cin=1, ncls=9, hw=32, seed=23)),
                                    ("unet_d", dict(B=2, cin=3, ncls=4, hw=64, seed=37)),
                                    ("unet_e", dict(B=2, cin=3, ncls=4, hw=64, seed=42))])
def test_unet(golden, name, c):
    g = golden("unet")
    # float64 oracle: the reference's CPU BatchNorm accumulates in double, and the
    # weight gradients of a BN network cancel heavily, so a float32 restatement is
    # noisier (up to 3e-2 on small gradient tensors) than the reference it checks.
    st = orc.make_state(c["cin"], c["ncls"], base=8, seed=c["seed"], dtype=torch.float64)
    assert len(st) == 128
    keys = orc.param_keys(st)
    assert len(keys) == 74
    for k in keys:
        st[k].requires_grad_(True)
    x = T(dg.hash_normalish((c["B"], c["cin"], c["hw"], c["hw"]), c["seed"] + 100)).double()
    t = T(dg.blob_mask(c["B"], c["hw"], c["hw"], c["ncls"], c["seed"] + 300))
    names = [f"c{i}" for i in range(1, c["ncls"])]
    logits = orc.unet_forward(x, st, training=True)
    ld = orc.dice_loss(logits, t, names)
    ld["total_loss"].backward()
    close(logits.detach(), g[f"{name}.logits"], 1e-4, 1e-4)
    close([float(v) for v in ld.values()], g[f"{name}.loss"], 1e-5, 1e-6)
    assert list(g[f"{name}.grad_keys"]) == keys
    norms = np.array([float(st[k].grad.double().norm()) for k in keys])
    is_cb = np.array([k.endswith(".0.bias") and k != "out_conv.bias" for k in keys])
    close(norms[~is_cb], g[f"{name}.grad_norms"][~is_cb], 1e-3, 1e-8)
    assert norms[is_cb].max() < 1e-10 and g[f"{name}.grad_norms"][is_cb].max() < 1e-3   # analytically zero
    samp = np.stack([st[k].grad.numpy().reshape(-1)[mg.sample_idx(st[k].numel())] for k in keys])
    scale = np.array([float(st[k].grad.abs().max()) for k in keys])[:, None]
    assert (np.abs(samp - g[f"{name}.grad_samples"])[~is_cb] / scale[~is_cb]).max() < 2e-3
    close(st["out_conv.weight"].grad, g[f"{name}.grad.out_conv.weight"], 1e-3, 1e-6)
    close(st["center_conv.cbr1.1.running_mean"].detach(), g[f"{name}.running_mean.center_conv.cbr1"], 1e-4, 1e-6)
    close(st["center_conv.cbr1.1.running_var"].detach(), g[f"{name}.running_var.center_conv.cbr1"], 1e-4, 1e-6)
    with torch.no_grad():
        close(orc.unet_forward(x, st, training=False), g[f"{name}.eval_logits"], 1e-4, 1e-4)


def test_pool_and_upsample(golden):
    g = golden("pool_up")
    x = dg.hash_normalish((2, 3, 8, 12), 31)
    x = np.maximum(x, 0)
    x[0, 0, :4, :4] = 0.5
    x = T(x).requires_grad_(True)
    y = orc.maxpool2x2(x)
    y.backward(T(dg.hash_normalish((2, 3, 4, 6), 32)))
    assert np.array_equal(y.detach().numpy(), g["pool.y"])
    assert np.array_equal(x.grad.numpy(), g["pool.dx"])
    for name, (h, w) in {"up_2": (2, 2), "up_5": (5, 5), "up_3x7": (3, 7)}.items():
        x = T(dg.hash_normalish((2, 3, h, w), 33 + h)).requires_grad_(True)
        skip = T(dg.hash_normalish((2, 2, 2 * h, 2 * w), 43 + h)).requires_grad_(True)
        y = torch.cat([orc.upsample2x_bilinear(x), skip], dim=1)
        y.backward(T(dg.hash_normalish(tuple(y.shape), 53 + h)))
        close(y.detach(), g[f"{name}.y"], 1e-6, 1e-6)
        close(x.grad, g[f"{name}.dx"], 1e-5, 1e-6)
        assert np.array_equal(skip.grad.numpy(), g[f"{name}.dskip"])


@pytest.mark.parametrize("name", list(mg.loss_cases()))
def test_losses(golden, name):
    g = golden("loss")
    c = mg.loss_cases()[name]
    x, t = mg.loss_inputs(c)
    names = [f"c{i}" for i in range(1, c["C"])]
    xl = T(x).requires_grad_(True)
    ld = orc.dice_loss(xl, T(t), names)
    assert list(ld.keys()) == names + ["total_loss"]
    ld["total_loss"].backward()
    close([float(v) for v in ld.values()], g[f"{name}.dice"], 1e-6, 1e-7)
    close(xl.grad, g[f"{name}.dice_dlogits"], 1e-4, 1e-9)
    li = orc.iou_loss(T(x), T(t), names)
    assert list(li.keys()) == names + ["total_loss"]
    close([float(v) for v in li.values()], g[f"{name}.iou"], 1e-6, 1e-7)
    ce = orc.ce_map(T(x), T(t))
    close(ce, g[f"{name}.ce_map"], 1e-5, 1e-6)
    # Focal_Loss (loss_fn.py:204-217) rebuilt from the CE map pins that map through the reference class
    pt = torch.exp(-ce)
    focal = 0.25 * ((1 - pt) ** 2) * ce
    vals = [float(focal[T(t) == i].mean()) if (t == i).any() else float("nan") for i in range(1, c["C"])]
    ref = g[f"{name}.focal"]
    np.testing.assert_allclose(vals, ref[:-1], rtol=1e-5, atol=1e-7, equal_nan=True)


@pytest.mark.parametrize("name", list(mg.metric_cases()))
def test_metrics_bit_exact(golden, name):
    g = golden("metrics")
    c = mg.metric_cases()[name]
    x, t = mg.metric_inputs(c)
    cm = orc.confusion_counts(T(x), T(t))
    tp, fn, fp, tn = orc._counts_to_onehot_terms(cm, c["n"] + 1)
    for a, k in ((tp, "TP"), (fn, "FN"), (fp, "FP"), (tn, "TN")):
        assert np.array_equal(a, g[f"{name}.{k}"])
    m = orc.metrics_from_counts(cm, c["n"])
    assert m.dtype == np.float64 and m.shape == (6, c["n"] + 1)
    assert np.array_equal(m, g[f"{name}.matrix"]), (m - g[f"{name}.matrix"])


@pytest.mark.parametrize("name", list(mg.metric_cases()))
def test_metrics_per_call_restatement_equals_reference(golden, name):
    """The pass-by-pass restatement (the one the CPU baseline times) gives the reference's matrix
    bit for bit, like the count-matrix form."""
    g = golden("metrics")
    c = mg.metric_cases()[name]
    x, t = mg.metric_inputs(c)
    m = orc.metrics_update_per_call(T(x), T(t), [f"c{i}" for i in range(1, c["n"] + 1)])
    assert np.array_equal(m, g[f"{name}.matrix"])


def test_aten_fused_timing_mode_equals_restatement():
    """The oracle's timing mode (ATen's fused batch_norm / interpolate, as the reference calls
    them) computes the same training step as the step-by-step restatement."""
    names = ["a", "b", "c"]
    x = T(dg.hash_normalish((2, 1, 32, 32), 9001))
    t = T(dg.blob_mask(2, 32, 32, 4, 9002))
    out = []
    for fused in (False, True):
        orc.ATEN_FUSED = fused
        try:
            ts = orc.TrainState(orc.make_state(1, 4, base=8, seed=33))
            for _ in range(2):
                losses, met, logits = orc.train_step(ts, x, t, names)
        finally:
            orc.ATEN_FUSED = False
        # (conv biases under train-mode BatchNorm have an analytically zero gradient; AdamW turns its
        # round-off into +-lr steps, which also reach the running means: both are left out)
        out.append((losses, met, logits, [ts.state[k].detach().clone() for k in ts.state
                                          if k.endswith((".0.weight", ".1.weight", ".1.bias", ".running_var"))
                                          or k.startswith("out_conv")]))
    close(out[0][0], out[1][0], 1e-5, 1e-6)
    assert float((out[0][2] - out[1][2]).abs().max()) < 1e-4 * float(out[0][2].abs().max())
    for a, b in zip(out[0][3], out[1][3]):
        assert float((a.double() - b.double()).abs().max()) < 2e-3 * float(a.double().abs().max()) + 1e-5


def test_train_steps(golden):
    g = golden("train")
    c = mg.train_case()
    st = orc.make_state(c["cin"], c["ncls"], base=8, seed=c["seed"], dtype=torch.float64)
    ts = orc.TrainState(st)
    names = ["a", "b", "c"]
    losses = np.zeros(4)
    mets = np.zeros((6, 4))
    batches = mg.train_batches(c)
    for x, t in batches:
        l, m, _ = orc.train_step(ts, T(x).double(), T(t), names, lr=c["lr"])
        losses += np.array(l)
        mets += m
    mets /= len(batches)
    close(losses, g["train.losses"], 1e-5, 1e-6)
    close(mets, g["train.metrics"], 1e-6, 1e-7)
    assert list(g["train.keys"]) == list(st.keys())
    # conv biases have an analytically zero gradient: AdamW turns their round-off
    # noise into +-lr moves, so they are not comparable between implementations.
    keep = np.array([not (k.endswith(".0.bias") and k != "out_conv.bias") for k in st])
    sums = np.array([float(v.detach().double().sum()) for v in st.values()])
    # AdamW's first steps move every element by about +-lr whatever the gradient's size,
    # so an element whose gradient is round-off small may legitimately land 2*lr*steps away:
    # bound the sums by a few such flips and require almost all elements to agree closely.
    flip = 2 * c["lr"] * c["steps"]
    assert np.abs(sums[keep] - g["train.param_sums"][keep]).max() < 4 * flip
    for k in ("out_conv.weight", "encoder1.cbr1.0.weight"):
        d = np.abs(st[k].detach().numpy() - g["train." + k])
        assert d.max() <= flip + 1e-5 and (d < 1e-5).mean() > 0.95, (k, d.max(), (d < 1e-5).mean())
    # evaluate(): eval-mode forward with the trained running statistics
    el = np.zeros(4)
    em = np.zeros((6, 4))
    with torch.no_grad():
        for x, t in batches:
            lg = orc.unet_forward(T(x).double(), st, training=False)
            el += np.array([float(v) for v in orc.dice_loss(lg, T(t), names).values()])
            em += orc.metrics_update(lg, T(t), names)
    # eval mode sees the (noise-driven) conv biases through the running means: looser bound
    close(el, g["eval.losses"], 1e-3, 1e-5)
    close(em / len(batches), g["eval.metrics"], 1e-2, 5e-3)


def test_dp_emulation(golden):
    g = golden("dp")
    c = dict(B=4, cin=1, ncls=4, hw=32, seed=81)
    x = dg.hash_normalish((c["B"], c["cin"], c["hw"], c["hw"]), c["seed"] + 100)
    t = dg.blob_mask(c["B"], c["hw"], c["hw"], c["ncls"], c["seed"] + 300)
    grads, losses = None, []
    for r in range(2):
        st = orc.make_state(c["cin"], c["ncls"], base=8, seed=c["seed"])
        keys = orc.param_keys(st)
        for k in keys:
            st[k].requires_grad_(True)
        sl = slice(2 * r, 2 * r + 2)
        ld = orc.dice_loss(orc.unet_forward(T(x[sl]), st), T(t[sl]), ["a", "b", "c"])
        ld["total_loss"].backward()
        losses.append(float(ld["total_loss"]))
        gr = [st[k].grad.double().numpy() for k in keys]
        grads = gr if grads is None else [a + b for a, b in zip(grads, gr)]
    close(losses, g["dp.shard_losses"], 1e-5, 1e-6)
    close([np.sqrt(((a / 2) ** 2).sum()) for a in grads], g["dp.grad_norms"], 2e-3, 1e-7)


def test_slide(golden):
    g = golden("slide")
    model = mg.StubSegModel()
    for name, c in mg.slide_cases().items():
        img = (dg.hash_u64((c["h"], c["w"], 3), c["seed"]) % np.uint64(256)).astype(np.uint8)
        padded, pads = orc.slide_pad(img, c["window"], c["stride"])
        assert tuple(pads) == tuple(g[f"{name}.pads"])
        pos = orc.slide_positions(padded.shape[0], padded.shape[1], c["window"], c["stride"])
        assert np.array_equal(np.array(pos), g[f"{name}.positions"])
        maps = []
        for (y0, y1, x0, x1) in pos:
            tile = torch.from_numpy(padded[y0:y1, x0:x1]).permute(2, 0, 1).float() / 255.0
            maps.append(torch.argmax(model(tile[None]), dim=1)[0].numpy())
        res = orc.slide_blend(maps, pos, padded.shape[:2], img.shape[:2], c["window"])
        assert np.array_equal(res, g[f"{name}.result"])
    for (h, w, win, s, n_tiles, n_unique) in [(4096, 4096, 1024, 512, 64, 49), (4096, 4096, 256, 128, 1024, 961)]:
        pos = orc.slide_positions(h, w, win, s)
        assert np.array_equal(np.array(pos), g[f"slide_pos_{h}x{w}_{win}_{s}.positions"])
        assert len(pos) == n_tiles and len(set(pos)) == n_unique


def _state_from_golden(g, name, seed):
    spec = [(str(k), tuple(int(v) for v in str(s).split(",") if v)) for k, s in zip(g[f"{name}.keys"], g[f"{name}.shapes"])]
    return orc.hash_state(spec, seed)


def test_amsfn_block(golden):
    """The oracle's AMSFN against the reference block (model/utils/modules.py:791-871)."""
    g = golden("a_unet")
    st = _state_from_golden(g, "amsfn", 43)
    st = {"blk." + k: v.double() if v.is_floating_point() else v for k, v in st.items()}
    pk = [k for k in st if st[k].is_floating_point() and "running" not in k]
    for k in pk:
        st[k].requires_grad_(True)
    x = T(np.maximum(dg.hash_normalish((2, 8, 12, 20), 143), 0)).double().requires_grad_(True)
    gy = T(dg.hash_normalish((2, 8, 12, 20), 243)).double()
    y = orc.amsfn(x, st, "blk", training=True)
    y.backward(gy)
    close(y.detach(), g["amsfn.y"], 1e-4, 1e-5)
    close(x.grad, g["amsfn.dx"], 1e-3, 1e-5)
    typical = max(float(np.abs(g[f"amsfn.grad.{k[4:]}"]).max()) for k in pk)
    for k in pk:
        ref = g[f"amsfn.grad.{k[4:]}"]
        # biases that feed a train-mode BatchNorm have an analytically zero gradient: round-off on both sides
        if k.endswith(("pconv.bias", "dconv.bias", "conv1x1.1.bias", "final_conv.0.bias")):
            assert np.abs(ref).max() < 1e-4 * typical and float(st[k].grad.abs().max()) < 1e-4 * typical
            continue
        scale = np.abs(ref).max() + 1e-6
        close(st[k].grad / scale, ref / scale, 0, 2e-4)


@pytest.mark.parametrize("name", list(mg.a_unet_cases()))
def test_a_unet(golden, name):
    g = golden("a_unet")
    c = mg.a_unet_cases()[name]
    st = _state_from_golden(g, name, c["seed"])
    st = {k: v.double() if v.is_floating_point() else v for k, v in st.items()}
    x = T(dg.hash_normalish((c["B"], c["cin"], c["hw"], c["hw"]), c["seed"] + 100)).double()
    t = T(dg.blob_mask(c["B"], c["hw"], c["hw"], c["ncls"], c["seed"] + 300))
    names = [f"c{i}" for i in range(1, c["ncls"])]
    keys = [str(k) for k in g[f"{name}.grad_keys"]]
    for k in keys:
        st[k].requires_grad_(True)
    logits = orc.a_unet_forward(x, st, training=True)
    ld = orc.dice_loss(logits, t, names)
    ld["total_loss"].backward()
    # a 90-layer float32 network against a float64 restatement: relative to the logits' scale
    scale = float(np.abs(g[f"{name}.logits"]).max())
    assert float(np.abs(logits.detach().numpy() - g[f"{name}.logits"]).max()) < 1e-3 * scale
    close([float(v) for v in ld.values()], g[f"{name}.loss"], 1e-5, 1e-6)
    norms = np.array([float(st[k].grad.norm()) for k in keys])
    ref = g[f"{name}.grad_norms"]
    big = ref > 1e-3 * ref.max()          # analytically-zero bias gradients are round-off on both sides
    assert np.abs(norms[big] - ref[big]).max() / ref.max() < 2e-2
    close(st["acpn0.final_conv.1.running_mean"].detach(), g[f"{name}.running_mean.acpn0.final_conv"], 1e-4, 1e-5)
    with torch.no_grad():
        ev = orc.a_unet_forward(x, st, training=False).numpy()
        assert float(np.abs(ev - g[f"{name}.eval_logits"]).max()) < 1e-3 * float(np.abs(g[f"{name}.eval_logits"]).max())


def _state_from_golden(g, name, seed, dtype=torch.float64):
    spec = [(str(k), tuple(int(v) for v in str(s).split(",") if v)) for k, s in zip(g[f"{name}.keys"], g[f"{name}.shapes"])]
    return orc.hash_state(spec, seed, dtype=dtype)


@pytest.mark.parametrize("name", list(mg.rdam_block_cases()))
def test_rdam_blocks(golden, name):
    """ResDConv and MDAM restatements against the reference's own outputs and gradients."""
    g = golden("rdam")
    kind, args, shape, seed = mg.rdam_block_cases()[name]
    st = _state_from_golden(g, name, seed)
    pk = [k for k in st if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]
    for k in pk:
        st[k].requires_grad_(True)
    x = dg.hash_normalish(shape, seed + 100)
    x = T(np.maximum(x, 0) if kind == "resd" else x).double().requires_grad_(True)
    st2 = {"blk." + k: v for k, v in st.items()}
    y = orc.res_dconv(x, st2, "blk") if kind == "resd" else orc.mdam(x, st2, "blk")
    gy = T(dg.hash_normalish(tuple(y.shape), seed + 200)).double()
    y.backward(gy)
    close(y.detach().numpy(), g[f"{name}.y"], 1e-5, 1e-5)
    # (the fixture is the reference's float32 run: its input gradient, three BatchNorm backwards deep,
    # carries ~1e-4 of its scale in round-off)
    assert np.abs(x.grad.numpy() - g[f"{name}.dx"]).max() < 5e-4 * np.abs(g[f"{name}.dx"]).max()
    scale = max(float(np.abs(g[f"{name}.grad.{k}"]).max()) for k in pk)
    for k in pk:
        ref = g[f"{name}.grad.{k}"]
        got = st[k].grad.numpy() if st[k].grad is not None else np.zeros_like(ref)
        if kind == "resd" and k.endswith(".0.bias"):
            # conv bias in front of a train-mode BatchNorm: analytically zero, float32 round-off in the fixture
            assert np.abs(ref).max() < 1e-4 * scale and np.abs(got).max() < 1e-9 * scale, k
            continue
        assert np.abs(got - ref).max() < 5e-4 * max(np.abs(ref).max(), 1e-3 * scale) + 1e-7, k
    if kind == "mdam":
        # the strip-pooling branch ends in GroupNorm -> global average = the GroupNorm bias: its
        # parameters get round-off-sized gradients in the reference (the HIP path relies on that)
        for k in ("conv1x1.weight", "conv1x1.bias", "groupNorm.weight"):
            assert np.abs(g[f"{name}.grad.{k}"]).max() < 1e-5 * scale, k


@pytest.mark.parametrize("name", list(mg.rdam_cases()))
def test_rdam_unet(golden, name):
    g = golden("rdam")
    c = mg.rdam_cases()[name]
    st = _state_from_golden(g, name, c["seed"])
    x = T(dg.hash_normalish((c["B"], c["cin"], c["hw"], c["hw"]), c["seed"] + 100)).double()
    t = T(dg.blob_mask(c["B"], c["hw"], c["hw"], c["ncls"], c["seed"] + 300))
    logits = orc.rdam_unet_forward(x, st)
    rel = lambda a, b: float(np.abs(np.asarray(a) - b).max() / np.abs(b).max())
    assert rel(logits.numpy(), g[f"{name}.logits"]) < 1e-3         # float32 reference, ~100 layers deep
    ld = orc.dice_loss(logits, t, [f"c{i}" for i in range(1, c["ncls"])])
    close([float(v) for v in ld.values()], g[f"{name}.loss"], 1e-5, 1e-6)
    close(st["encoder3.cbr3.1.running_mean"].numpy(), g[f"{name}.running_mean.encoder3.cbr3"], 1e-4, 1e-6)
    assert rel(orc.rdam_unet_forward(x, st, training=False).numpy(), g[f"{name}.eval_logits"]) < 1e-3


def test_philox_known_answer_vectors():
    """The counter-based generator behind mis_dropout2d_noise: the oracle's Philox4x32-10 against the
    published known-answer vectors (Random123 kat_vectors, `philox4x32 10` lines), and the shape of the
    multipliers it derives."""
    kat = [((0, 0, 0, 0), (0, 0), "6627e8d5 e169c58d bc57ac4c 9b00dbd8"),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, "408f276d 41c83b0e a20bc7c6 6d5451fd"),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            "d16cfe09 94fdcceb 5001e420 24126ea1")]
    for ctr, key, want in kat:
        got = " ".join("%08x" % int(v) for v in orc.philox4x32_10(ctr, key))
        assert got == want
    nz = orc.dropout2d_noise_philox(1234, 4, 16, 256, 0.3)
    assert nz.shape == (16, 256) and nz.dtype == np.float32
    assert set(np.unique(nz)) == {np.float32(0.0), np.float32(1.0) / (np.float32(1.0) - np.float32(0.3))}
    assert abs(float((nz > 0).mean()) - 0.7) < 0.02
    assert not np.array_equal(nz, orc.dropout2d_noise_philox(1234, 8, 16, 256, 0.3))
    assert np.all(orc.dropout2d_noise_philox(1, 0, 2, 8, 1.0) == 0) and np.all(orc.dropout2d_noise_philox(1, 0, 2, 8, 0.0) == 1)
