"""GPU suite: whole-model parity — UNet forward/backward, the training harness and the
evaluation loop against the golden vectors produced by the reference, in float32 mode (the
reference's CPU path) and bfloat16 mode (the benchmark's compute type)."""
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


def build(c, dtype):
    from medical_image_segmentation_amd.model.unet import UNet
    m = UNet(c["cin"], c["ncls"], 0, base_channels=8, compute_dtype=dtype)
    st = orc.make_state(c["cin"], c["ncls"], base=8, seed=c["seed"])
    assert list(m.state_dict().keys()) == list(st.keys())
    m.load_state_dict({k: v.clone() for k, v in st.items()})
    return m.to(DEV)


UNET_CASES = [("unet_a", dict(B=2, cin=1, ncls=4, hw=32, seed=21)),
              ("unet_b", dict(B=2, cin=3, ncls=4, hw=64, seed=22)),
              ("unet_c", dict(B=1, cin=1, ncls=9, hw=32, seed=23)),
              ("unet_d", dict(B=2, cin=3, ncls=4, hw=64, seed=37)),
              ("unet_e", dict(B=2, cin=3, ncls=4, hw=64, seed=42))]


def _grad_deviation(m, g, name):
    """(per-parameter grad-norm deviation, per-parameter worst sampled-element deviation relative
    to the gradient's scale) against the golden vectors; conv biases under BatchNorm excluded
    (their gradient is analytically zero: rounding noise on both sides)."""
    params = dict(m.named_parameters())
    keys = list(g[f"{name}.grad_keys"])
    assert keys == list(params.keys())
    norms = np.array([float(params[k].grad.double().norm()) for k in keys])
    is_cb = np.array([k.endswith(".0.bias") and k != "out_conv.bias" for k in keys])
    ref = g[f"{name}.grad_norms"][~is_cb]
    dev = np.abs(norms[~is_cb] - ref) / (ref + 1e-7)
    samp = np.stack([params[k].grad.detach().cpu().numpy().reshape(-1)[mg.sample_idx(params[k].numel())] for k in keys])
    scale = np.array([float(params[k].grad.abs().max()) for k in keys])[:, None]
    sdev = (np.abs(samp - g[f"{name}.grad_samples"])[~is_cb] / scale[~is_cb]).max(axis=1)
    return dev, sdev


LOGITS_FP32_TOL = 1e-5      # (measured: 2-3e-6 on both conv paths)


def _oracle_grads(c, dtype):
    """Gradients of the CPU oracle in `dtype` on the case's weights and batch (conv biases under BatchNorm left out)."""
    x = T(dg.hash_normalish((c["B"], c["cin"], c["hw"], c["hw"]), c["seed"] + 100))
    t = T(dg.blob_mask(c["B"], c["hw"], c["hw"], c["ncls"], c["seed"] + 300))
    st = orc.make_state(c["cin"], c["ncls"], base=8, seed=c["seed"], dtype=dtype)
    keys = [k for k in orc.param_keys(st) if not (k.endswith(".0.bias") and k != "out_conv.bias")]
    for k in orc.param_keys(st):
        st[k].requires_grad_(True)
    lo = orc.unet_forward(x.to(dtype), st, training=True)
    orc.dice_loss(lo, t, [f"c{i}" for i in range(1, c["ncls"])])["total_loss"].backward()
    return {k: st[k].grad.double() for k in keys}


def _deviation(a, b):
    """(worst per-tensor norm deviation, worst element deviation relative to the tensor's scale) of gradient dicts."""
    dn = max(abs(float(a[k].norm()) - float(b[k].norm())) / (float(b[k].norm()) + 1e-12) for k in b)
    de = max(float((a[k] - b[k]).abs().max() / (b[k].abs().max() + 1e-30)) for k in b)
    return dn, de


@pytest.mark.parametrize("name,c", UNET_CASES)
def test_unet_fp32_golden(golden, report, name, c):
    """float32 forward + backward of the whole network against the reference's own outputs, on the default conv
    path (matrix-core kernels where taken) and on the vector-ALU path.

    Forward: logits 1e-5 (measured 2-8e-6), losses 1e-5.  Gradients, on BOTH paths: every tensor within 5e-3 (norm,
    and elements relative to the tensor's scale) of the reference's fixture and of the float64 oracle — whenever the
    fixture is insensitive to float32 round-off, which is measured, not assumed: the Dice gradient is concentrated on
    few pixels, and a ReLU whose pre-activation lies within round-off of zero at such a pixel moves whole tensors by
    per cents in the 2-channel BatchNorm layers of these base-8 networks.  The oracle run in float32 and in float64 —
    the same arithmetic at two precisions — tells the cases apart: the two agree to ~1e-5 on unet_a/c/d/e (fixtures
    d and e were chosen that way, tools/find_stable_unet_seed.py) and differ by 3.3e-2 (norms) / 6.3e-2 (elements)
    on unet_b.  On such a fixture this implementation must still agree to 5e-3 with ONE of the two precisions of
    the oracle (it takes the float32 oracle's side of that ReLU), and stay within 10 % of the reference's fixture."""
    from medical_image_segmentation_amd import _lib
    from medical_image_segmentation_amd.utils.loss_fn import diceloss
    g = golden("unet")
    x = T(dg.hash_normalish((c["B"], c["cin"], c["hw"], c["hw"]), c["seed"] + 100)).to(DEV)
    t = T(dg.blob_mask(c["B"], c["hw"], c["hw"], c["ncls"], c["seed"] + 300)).to(DEV)
    names = [f"c{i}" for i in range(1, c["ncls"])]
    o64, o32 = _oracle_grads(c, torch.float64), _oracle_grads(c, torch.float32)
    sens = max(_deviation(o32, o64))
    report("float32-vs-float64 oracle gradient deviation (sensitivity of the fixture)", sens)
    fixture_is_stable = sens < 1e-3
    assert fixture_is_stable == (name != "unet_b")
    for mode in (0, 1):
        _lib.call("mis_set_conv_path", mode)
        try:
            m = build(c, torch.float32).train()
            logits = m(x)
            assert logits.shape == (c["B"], c["ncls"], c["hw"], c["hw"]) and logits.dtype == torch.float32
            ld = diceloss(names)(logits, t)
            ld["total_loss"].backward()
            e_logits = rel_err(logits, g[f"{name}.logits"])
            got_loss = np.array([float(v.detach()) for v in ld.values()])
            e_loss = float(np.max(np.abs(got_loss - g[f"{name}.loss"]) / np.abs(g[f"{name}.loss"])))
            report(f"conv path {mode}: fp32 logits max error / max |logit|", e_logits)
            report(f"conv path {mode}: fp32 loss max relative error", e_loss)
            assert e_logits < LOGITS_FP32_TOL
            np.testing.assert_allclose(got_loss, g[f"{name}.loss"], rtol=1e-5, atol=1e-6)
            dev, sdev = _grad_deviation(m, g, name)
            report(f"conv path {mode}: fp32 gradient norm deviation vs reference fixture (worst tensor)", float(dev.max()))
            report(f"conv path {mode}: fp32 gradient sample deviation vs reference fixture (worst tensor)", float(sdev.max()))
            params = dict(m.named_parameters())
            got = {k: params[k].grad.double().cpu() for k in o64}
            d64, d32 = max(_deviation(got, o64)), max(_deviation(got, o32))
            report(f"conv path {mode}: fp32 gradient deviation vs float64 oracle (worst tensor, norm or element)", d64)
            report(f"conv path {mode}: fp32 gradient deviation vs float32 oracle (worst tensor, norm or element)", d32)
            if fixture_is_stable:
                assert d64 < 5e-3 and dev.max() < 5e-3 and sdev.max() < 5e-3, (mode, d64, dev.max(), sdev.max())
            else:
                assert min(d64, d32) < 5e-3, (mode, d64, d32)
                assert dev.max() < 0.10 and sdev.max() < 0.10, (mode, dev.max(), sdev.max())
            assert rel_err(m.center_conv.cbr1[1].running_mean, g[f"{name}.running_mean.center_conv.cbr1"]) < 1e-4
            assert rel_err(m.center_conv.cbr1[1].running_var, g[f"{name}.running_var.center_conv.cbr1"]) < 1e-4
            m.eval()
            with torch.no_grad():
                assert rel_err(m(x), g[f"{name}.eval_logits"]) < 1e-4
        finally:
            _lib.call("mis_set_conv_path", 0)


@pytest.mark.parametrize("name,c", UNET_CASES)
def test_unet_bf16(golden, report, name, c):
    """bfloat16 compute mode.  Bounds, all relative:
      loss            within 1e-2 of the float32 reference (the stated bf16 tolerance);
      logits          within 1e-2 (L2) of the float64 oracle that rounds exactly the tensors
                      this mode stores in bfloat16 — i.e. the kernels add nothing beyond the
                      storage format;
      logits vs fp32  within 6e-2 (L2): bfloat16 weight rounding alone moves this network's
                      logits by 1-2e-2 and the stored activations bring it to 2.5-3.5e-2, for
                      any implementation (see oracle.unet_forward_bf16_storage)."""
    from medical_image_segmentation_amd.utils.loss_fn import diceloss
    g = golden("unet")
    m = build(c, torch.bfloat16).train()
    xh = T(dg.hash_normalish((c["B"], c["cin"], c["hw"], c["hw"]), c["seed"] + 100))
    th = T(dg.blob_mask(c["B"], c["hw"], c["hw"], c["ncls"], c["seed"] + 300))
    x, t = xh.to(DEV), th.to(DEV)
    names = [f"c{i}" for i in range(1, c["ncls"])]
    logits = m(x)
    assert logits.dtype == torch.bfloat16
    ld = diceloss(names)(logits, t)
    ld["total_loss"].backward()
    got_loss = np.array([float(v.detach()) for v in ld.values()])
    report("bf16 loss max relative error vs fp32 reference",
           float(np.max(np.abs(got_loss - g[f"{name}.loss"]) / np.abs(g[f"{name}.loss"]))))
    np.testing.assert_allclose(got_loss, g[f"{name}.loss"], rtol=1e-2)
    ref = T(g[f"{name}.logits"]).double()
    st64 = orc.make_state(c["cin"], c["ncls"], base=8, seed=c["seed"], dtype=torch.float64)
    emu = orc.unet_forward_bf16_storage(xh.double(), st64)
    got = logits.detach().double().cpu()
    l2_emu = float((got - emu).norm() / emu.norm())
    l2_ref = float((got - ref).norm() / ref.norm())
    report("bf16 logits L2 error vs bf16-storage oracle", l2_emu)
    report("bf16 logits L2 error vs fp32 reference", l2_ref)
    assert l2_emu < 1e-2, l2_emu
    assert l2_ref < 6e-2, l2_ref
    # gradients: direction agrees with the float32 reference
    params = dict(m.named_parameters())
    gw = params["out_conv.weight"].grad.double().cpu().reshape(-1)
    rw = T(g[f"{name}.grad.out_conv.weight"]).double().reshape(-1)
    assert float(torch.dot(gw, rw) / (gw.norm() * rw.norm())) > 0.99


def test_autocast_selects_bf16():
    c = UNET_CASES[0][1]
    m = build(c, None).eval()
    x = torch.zeros(1, 1, 32, 32, device=DEV)
    with torch.no_grad():
        assert m(x).dtype == torch.float32
        with torch.autocast("cuda", dtype=torch.bfloat16):
            assert m(x).dtype == torch.bfloat16


def test_train_one_epoch_and_evaluate_golden(golden):
    from medical_image_segmentation_amd.utils.loss_fn import diceloss
    from medical_image_segmentation_amd.utils.metrics import Metrics
    from medical_image_segmentation_amd.utils.train_and_eval import train_one_epoch, evaluate
    g = golden("train")
    c = mg.train_case()
    m = build(c, torch.float32)
    names = ["a", "b", "c"]
    opt = torch.optim.AdamW(m.parameters(), lr=c["lr"], betas=(0.95, 0.999), weight_decay=1e-4)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0)
    scaler = torch.amp.GradScaler("cuda", enabled=False)
    loader = [((T(x), T(t)), [f"img{i}"]) for i, (x, t) in enumerate(mg.train_batches(c))]
    losses, mets = train_one_epoch(m, opt, 0, loader, torch.device(DEV), diceloss(names), scaler, Metrics(names),
                                   sched, names, False, 0.0, 0.0)
    assert len(losses) == 4 and mets.shape == (6, 4)
    np.testing.assert_allclose([float(v) for v in losses], g["train.losses"], rtol=2e-4)
    np.testing.assert_allclose(mets, g["train.metrics"], rtol=0, atol=2e-3)
    sd = m.state_dict()
    assert list(sd.keys()) == list(g["train.keys"])
    flip = 2 * c["lr"] * c["steps"]
    for k in ("out_conv.weight", "encoder1.cbr1.0.weight"):
        d = np.abs(sd[k].cpu().numpy() - g["train." + k])
        assert d.max() <= flip + 1e-5 and (d < 2e-5).mean() > 0.9, (k, d.max(), (d < 2e-5).mean())
    assert int(sd["encoder1.cbr1.1.num_batches_tracked"]) == c["steps"]
    el, em = evaluate(m, torch.device(DEV), loader, diceloss(names), Metrics(names), names)
    np.testing.assert_allclose([float(v) for v in el], g["eval.losses"], rtol=2e-3)
    # After training, parameters may differ from the reference's by AdamW sign flips on
    # zero-gradient conv biases (bounded above), which moves near-tie argmax pixels of this barely
    # trained model: the golden metrics are a loose check, the oracle on OUR weights the tight one.
    np.testing.assert_allclose(em, g["eval.metrics"], rtol=0, atol=3e-2)
    state = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    om = np.zeros_like(em)
    for (x, t), _ in loader:
        logits = orc.unet_forward(x.float(), state, training=False, update_stats=False)
        om += orc.metrics_update(logits, t.to(torch.int64).squeeze(1) if t.dim() == 4 else t.to(torch.int64), names)
    np.testing.assert_allclose(em, om / len(loader), rtol=0, atol=2e-3)


def test_checkpoint_roundtrip_with_reference_keys(tmp_path):
    c = UNET_CASES[0][1]
    m = build(c, torch.float32)
    path = tmp_path / "ck.pth"
    torch.save({"model": m.state_dict()}, path)
    sd = torch.load(path, weights_only=True)["model"]
    assert len(sd) == 128
    m2 = build(c, torch.float32)
    m2.load_state_dict(sd)


def test_fused_clip_adamw_matches_torch():
    """utils.optim.clip_and_step against torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW.step on
    identical clones (two param groups, 3 steps, norms above and below the clip threshold), and the
    optimizer state stays loadable by a plain torch AdamW."""
    from medical_image_segmentation_amd.utils.optim import clip_and_step
    shapes = [(8, 1, 3, 3), (8,), (32, 8, 3, 3), (32,), (5000,), (4097,), (1,), (64, 64, 3, 3)]
    mk = lambda: [torch.nn.Parameter(T(dg.hash_normalish(s, 1200 + i, std=0.3)).to(DEV)) for i, s in enumerate(shapes)]
    pa, pb = mk(), mk()
    groups = lambda ps: [{"params": ps[:5], "lr": 8e-4, "weight_decay": 1e-4},
                         {"params": ps[5:], "lr": 3e-3, "weight_decay": 0.0, "betas": (0.9, 0.99)}]
    oa = torch.optim.AdamW(groups(pa), betas=(0.95, 0.999), eps=1e-8)
    ob = torch.optim.AdamW(groups(pb), betas=(0.95, 0.999), eps=1e-8)
    for step, gscale in enumerate([5.0, 1e-3, 0.7]):
        for i, (a, b) in enumerate(zip(pa, pb)):
            g = T(dg.hash_normalish(shapes[i], 1300 + 10 * step + i, std=gscale)).to(DEV)
            a.grad, b.grad = g.clone(), g.clone()
        na = clip_and_step(oa, pa, max_norm=1.0)
        nb = torch.nn.utils.clip_grad_norm_(pb, max_norm=1.0)
        ob.step()
        assert abs(float(na) - float(nb)) <= 2e-6 * float(nb)
        for a, b in zip(pa, pb):
            assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-8
            assert float((a.grad - b.grad).abs().max()) <= 2e-6 * float(b.grad.abs().max()) + 1e-12
            sa, sb = oa.state[a], ob.state[b]
            assert float(sa["step"]) == float(sb["step"]) == step + 1
            assert rel_err(sa["exp_avg"], sb["exp_avg"]) < 2e-6 and rel_err(sa["exp_avg_sq"], sb["exp_avg_sq"]) < 2e-6
    oc = torch.optim.AdamW(groups(mk()), betas=(0.95, 0.999), eps=1e-8)
    oc.load_state_dict(oa.state_dict())          # same state layout as torch's own
    assert all(set(v.keys()) == {"step", "exp_avg", "exp_avg_sq"} for v in oa.state_dict()["state"].values())
    # reloading the state replaces the moment tensors: the cached description must notice
    oa.load_state_dict(oa.state_dict())
    ob.load_state_dict(ob.state_dict())
    for i, (a, b) in enumerate(zip(pa, pb)):
        g = T(dg.hash_normalish(shapes[i], 1400 + i, std=0.5)).to(DEV)
        a.grad, b.grad = g.clone(), g.clone()
    clip_and_step(oa, pa, max_norm=1.0)
    torch.nn.utils.clip_grad_norm_(pb, max_norm=1.0)
    ob.step()
    for a, b in zip(pa, pb):
        assert float((a - b).abs().max()) <= 4e-6 * float(b.abs().max()) + 1e-8
        assert float(oa.state[a]["step"]) == float(ob.state[b]["step"]) == 4
        assert rel_err(oa.state[a]["exp_avg_sq"], ob.state[b]["exp_avg_sq"]) < 4e-6
    # an optimizer the fused path does not take runs torch's calls unchanged
    ps = mk()
    sgd = torch.optim.SGD(ps, lr=0.1)
    for p in ps:
        p.grad = torch.ones_like(p)
    before = [p.detach().clone() for p in ps]
    clip_and_step(sgd, ps, max_norm=1.0)
    assert all(float((b - p).abs().max()) > 0 for b, p in zip(before, ps))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_unet_dropout_matches_oracle(dtype):
    """Train-mode forward/backward with Dropout2d active (p = 0.5 -> layer rates 0.15 .. 0.5):
    the noise tensors the model drew are replayed in the oracle.  float32: logits 1e-4, gradient
    norms 2 % (one ReLU-mask flip moves them by per cents in this tiny net); bfloat16: loss 1e-2."""
    from medical_image_segmentation_amd import functional as F_
    from medical_image_segmentation_amd.model.unet import UNet
    from medical_image_segmentation_amd.utils.loss_fn import diceloss
    c = dict(B=2, cin=3, ncls=4, hw=64, seed=22)
    m = UNet(c["cin"], c["ncls"], 0.5, base_channels=8, compute_dtype=dtype)
    st = orc.make_state(c["cin"], c["ncls"], base=8, seed=c["seed"])
    m.load_state_dict({k: v.clone() for k, v in st.items()})
    m = m.to(DEV).train()
    assert abs(m.encoder_dropout1.p - 0.15) < 1e-12 and m.bottleneck_dropout.p == 0.5
    xh = T(dg.hash_normalish((c["B"], c["cin"], c["hw"], c["hw"]), c["seed"] + 100))
    th = T(dg.blob_mask(c["B"], c["hw"], c["hw"], c["ncls"], c["seed"] + 300))
    drawn = []
    orig = F_.dropout2d_noise

    def recording(layer, x):
        n = orig(layer, x)
        drawn.append(n)
        return n

    F_.dropout2d_noise = recording
    try:
        torch.manual_seed(7)
        logits = m(xh.to(DEV))
    finally:
        F_.dropout2d_noise = orig
    assert len(drawn) == 7 and all(n is not None for n in drawn)
    assert any(bool((n == 0).any()) for n in drawn)            # something was actually dropped
    names = ["a", "b", "c"]
    ld = diceloss(names)(logits, th.to(DEV))
    ld["total_loss"].backward()

    keys = orc.param_keys(st)
    sto = {k: v.clone().double() for k, v in st.items()}
    for k in keys:
        sto[k].requires_grad_(True)
    lo = orc.unet_forward(xh.double(), sto, training=True, noise=[n.cpu().double() for n in drawn])
    ldo = orc.dice_loss(lo, th, names)
    ldo["total_loss"].backward()
    if dtype == torch.float32:
        assert rel_err(logits, lo) < 1e-4
        assert abs(float(ld["total_loss"].detach()) - float(ldo["total_loss"])) < 1e-5
        params = dict(m.named_parameters())
        for k in keys:
            if k.endswith(".0.bias") and k != "out_conv.bias":
                continue
            a, b = float(params[k].grad.double().norm()), float(sto[k].grad.norm())
            assert abs(a - b) <= 2e-2 * b + 1e-7, (k, a, b)
    else:
        assert abs(float(ld["total_loss"].detach()) - float(ldo["total_loss"])) < 1e-2
    # eval mode: dropout is the identity and nothing is drawn
    m.eval()
    state = torch.cuda.get_rng_state()
    with torch.no_grad():
        m(xh.to(DEV))
    assert torch.equal(state, torch.cuda.get_rng_state())


def test_weight_packer_tracks_updates():
    """One launch packs every layer's operand images; it must re-run after an optimizer step
    (fused or torch), after load_state_dict, and not at all between two inference calls."""
    from medical_image_segmentation_amd import functional as F_
    from medical_image_segmentation_amd.utils.optim import clip_and_step
    c = dict(B=2, cin=3, ncls=4, hw=32, seed=22)
    m = build(c, torch.float32).train()
    x = T(dg.hash_normalish((c["B"], c["cin"], c["hw"], c["hw"]), 5)).to(DEV)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-2)
    launches = []
    orig = F_.call

    def counting(name, *a):
        if name in ("mis_pack_conv3x3_weights_multi", "mis_pack_conv3x3_weights_pair", "mis_pack_conv3x3_weights"):
            launches.append(name)
        return orig(name, *a)

    F_.call = counting
    try:
        y0 = m(x)
        n_first = len(launches)
        assert launches.count("mis_pack_conv3x3_weights_multi") == 1
        assert n_first < 18        # layers the matrix-core kernels do not take (tiny channel counts) pack on their own
        y0.square().mean().backward()
        clip_and_step(opt, list(m.parameters()), 1.0)    # fused step writes through raw pointers
        y1 = m(x)
        assert launches.count("mis_pack_conv3x3_weights_multi") == 2
        assert float((y1 - y0).abs().max()) > 0          # the new weights were used
        opt.zero_grad()
        y1.square().mean().backward()
        torch.optim.SGD(m.parameters(), lr=1e-2).step()  # a torch optimizer
        y2 = m(x)
        assert launches.count("mis_pack_conv3x3_weights_multi") == 3
        assert float((y2 - y1).abs().max()) > 0
        # reference check of the cached images: same result as packing per layer
        F_._PACKED.clear()
        m.__dict__.pop("_mis_packer")
        before = len(launches)
        saved, F_.WeightPacker.refresh = F_.WeightPacker.refresh, lambda self: None
        try:
            y2b = m(x)
        finally:
            F_.WeightPacker.refresh = saved
        assert len(launches) - before >= 18 and torch.equal(y2, y2b)
        m.eval()
        with torch.no_grad():
            m(x)
            before = launches.count("mis_pack_conv3x3_weights_multi")
            m(x)
            assert launches.count("mis_pack_conv3x3_weights_multi") == before
    finally:
        F_.call = orig


def test_full_size_step_properties():
    """The benchmark configuration itself (16 x 1 x 512^2, base 32), too large for the CPU oracle:
    (i) no kernel on the path uses floating-point atomics, so two runs of the same step from the same
    state are bit-identical — logits, loss, every gradient; (ii) bfloat16 and float32 compute modes
    agree to the storage format's error (the bound measured on the oracle-sized cases); (iii) the
    per-class confusion counts of the fused loss/metrics pass sum to the pixel count; (iv) three
    steps of AdamW lower the loss."""
    from medical_image_segmentation_amd.model.unet import UNet
    from medical_image_segmentation_amd.utils.loss_fn import diceloss
    from medical_image_segmentation_amd.utils.metrics import Metrics
    from medical_image_segmentation_amd.utils.optim import clip_and_step
    B, S = 16, 512
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B, 1, S, S, generator=g).to(DEV)
    y = torch.randint(0, 4, (B, S, S), generator=g).to(DEV)
    state = {k: v.clone() for k, v in orc.make_state(1, 4, base=32, seed=61).items()}
    loss_fn, metric = diceloss(["a", "b", "c"]), Metrics(["a", "b", "c"])

    def run(dtype, steps=1):
        m = UNet(1, 4, 0, base_channels=32, compute_dtype=dtype)
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
                        [p.grad.clone() for p in m.parameters()]))
            clip_and_step(opt, list(m.parameters()), 1.0)
        assert int(status) == 0
        return out

    a, b = run(torch.bfloat16)[0], run(torch.bfloat16)[0]
    assert torch.equal(a[0], b[0]) and a[1] == b[1] and torch.equal(a[2], b[2])
    assert all(torch.equal(p, q) for p, q in zip(a[3], b[3]))
    assert int(a[2].sum()) == B * S * S and torch.equal(a[2].sum(dim=2).sum(dim=0).cpu(),
                                                          torch.bincount(y.reshape(-1).cpu(), minlength=4))
    f = run(torch.float32, steps=3)
    rel = float((a[0] - f[0][0]).norm() / f[0][0].norm())
    assert rel < 6e-2, rel
    assert abs(a[1] - f[0][1]) < 1e-2
    assert f[2][1] < f[0][1]


def test_cfg1_full_size_step_vs_oracle(report):
    """The reference's own CPU-runnable configuration at its REAL size — UNet(1, 4 classes, base 32),
    batch 4, 1 x 256 x 256, float32, diceloss, Metrics, clip_grad_norm_(1.0), AdamW(8e-4, wd 1e-4,
    betas (0.95, 0.999)) (train.py:221-226, utils/train_and_eval.py:63-121) — one training step against
    the float64 oracle on the same hash-generated batch and weights.  Every conv of the step runs many
    tiles per workgroup here (1024 tiles per image at full resolution).

    Bounds: logits 1e-5 of their scale, every loss entry 1e-5, Metrics.update bit-equal to the oracle's
    post-math on the SAME logits (the contract for integer counting) and within a few near-tie pixels of
    the oracle's own logits, per-tensor gradient norms 5e-3 and elements 3e-2 of the tensor's scale (conv biases under BatchNorm excluded: their
    gradient is analytically zero), updated weights within 2e-6 except AdamW sign flips of elements whose
    gradient is round-off."""
    from medical_image_segmentation_amd.model.unet import UNet
    from medical_image_segmentation_amd.utils.loss_fn import diceloss
    from medical_image_segmentation_amd.utils.metrics import Metrics
    from medical_image_segmentation_amd.utils.optim import clip_and_step
    B, S, names, lr = 4, 256, ["a", "b", "c"], 8e-4
    xh = T(dg.hash_normalish((B, 1, S, S), 9100))
    th = T(dg.blob_mask(B, S, S, 4, 9300))
    st32 = orc.make_state(1, 4, base=32, seed=91)
    m = UNet(1, 4, 0, base_channels=32, compute_dtype=torch.float32)
    m.load_state_dict({k: v.clone() for k, v in st32.items()})
    m = m.to(DEV).train()
    opt = torch.optim.AdamW(m.parameters(), lr=lr, betas=(0.95, 0.999), weight_decay=1e-4)
    logits = m(xh.to(DEV))
    ld = diceloss(names)(logits, th.to(DEV))
    met = Metrics(names).update(logits, th.to(DEV))
    ld["total_loss"].backward()
    params = dict(m.named_parameters())
    grads = {k: p.grad.detach().double().cpu() for k, p in params.items()}
    clip_and_step(opt, list(m.parameters()), 1.0)
    torch.cuda.synchronize()

    st = {k: v.clone().double() for k, v in st32.items()}
    ts = orc.TrainState(st)
    lo = orc.unet_forward(xh.double(), st, training=True)
    ldo = orc.dice_loss(lo, th, names)
    ldo["total_loss"].backward()
    e_logits = rel_err(logits, lo.detach())
    report("cfg1 256^2 base-32 fp32 logits max error / max |logit|", e_logits)
    assert e_logits < LOGITS_FP32_TOL
    got_loss = np.array([float(v.detach()) for v in ld.values()])
    ref_loss = np.array([float(v.detach()) for v in ldo.values()])
    report("cfg1 loss max relative error", float(np.max(np.abs(got_loss - ref_loss) / np.abs(ref_loss))))
    np.testing.assert_allclose(got_loss, ref_loss, rtol=1e-5, atol=1e-7)
    # metrics: exact on identical logits; the oracle's own logits differ by ~1e-6, which may move a near-tie pixel
    own = orc.metrics_update(logits.detach().float().cpu(), th, names)
    assert np.array_equal(met, own)
    flips = int((logits.detach().argmax(1).cpu() != lo.detach().argmax(1)).sum())
    report("cfg1 argmax pixels that differ from the oracle's logits (of 262144)", float(flips))
    assert flips <= 16
    np.testing.assert_allclose(met, orc.metrics_update(lo.detach().float(), th, names), rtol=0, atol=1e-3)
    worst, worst_el = (0.0, ""), (0.0, "")
    for k in ts.keys:
        if k.endswith(".0.bias") and k != "out_conv.bias":
            continue
        a, b = float(grads[k].norm()), float(st[k].grad.norm())
        worst = max(worst, (abs(a - b) / (b + 1e-12), k))
        worst_el = max(worst_el, (rel_err(grads[k], st[k].grad), k))
    report(f"cfg1 gradient norm deviation (worst tensor: {worst[1]})", worst[0])
    report(f"cfg1 gradient element deviation / max |g| (worst tensor: {worst_el[1]})", worst_el[0])
    # (a float32 forward differs from the float64 one by ~1e-6, which flips the ReLU of a few pixels whose
    # pre-activation is that close to zero; where such a pixel carries a large Dice gradient single weight-gradient
    # elements move by some 1e-3 of the tensor's scale — the norms are the sharp check)
    assert worst[0] < 5e-3, worst
    assert worst_el[0] < 3e-2, worst_el
    # the optimizer step
    og = [st[k].grad for k in ts.keys]
    orc.clip_grad_norm(og, 1.0)
    ts.step = 1
    orc.adamw_step([st[k] for k in ts.keys], og, ts.m, ts.v, 1, lr=lr)
    for k in ("out_conv.weight", "center_conv.cbr1.0.weight", "encoder1.cbr1.0.weight", "decoder4.cbr1.1.weight"):
        d = (params[k].detach().double().cpu() - st[k].detach()).abs()
        # (the first AdamW step moves every element by lr * g / (|g| + eps'): an element whose clipped gradient is
        # of the size of its own round-off takes a different step — about 1 % of the center layer's 1.2 M weights)
        assert float(d.max()) <= 2 * lr + 1e-6 and float((d < 2e-6).double().mean()) > 0.97, (k, float(d.max()))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_deferred_weight_gradient_reduction(dtype):
    """The weight-gradient reductions of a backward pass run as one launch when the pass ends.
    Same gradients, bit for bit, as with one reduction per layer; a second backward() into existing
    .grad tensors (autograd accumulates in place, so it reads each gradient at once) and a tensor
    hook on a weight take the immediate form and see complete values."""
    from medical_image_segmentation_amd import functional as F_
    from medical_image_segmentation_amd.utils.loss_fn import diceloss
    c = dict(B=2, cin=1, ncls=3, base=8, hw=32, seed=7)
    x = T(dg.hash_normalish((c["B"], c["cin"], c["hw"], c["hw"]), 4100)).to(DEV)
    t = T(dg.blob_mask(c["B"], c["hw"], c["hw"], c["ncls"], 4300)).to(DEV)
    names = ["a", "b"]

    def grads(defer, twice=False, hook=False):
        F_.DEFER_WGRAD_REDUCE = defer
        m = build(c, dtype).train()
        seen = []
        if hook:
            w = m.encoder2.cbr1[0].weight
            w.register_hook(lambda g: seen.append(g.clone()))
        for _ in range(2 if twice else 1):
            diceloss(names)(m(x), t)["total_loss"].backward()
        torch.cuda.synchronize()
        assert not F_._wgrad_pending
        out = {k: p.grad.clone() for k, p in m.named_parameters()}
        if hook:
            assert len(seen) == 1 and torch.equal(seen[0], out["encoder2.cbr1.0.weight"])
        return out

    ref = grads(False)
    got = grads(True)
    assert set(ref) == set(got)
    for k in ref:
        assert torch.equal(ref[k], got[k]), k
    hooked = grads(True, hook=True)
    for k in ref:
        assert torch.equal(ref[k], hooked[k]), k
    ref2, got2 = grads(False, twice=True), grads(True, twice=True)
    for k in ref2:
        assert torch.equal(ref2[k], got2[k]), k
        assert torch.isfinite(got2[k]).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("wgrad_stream", [True, False])
def test_weights_used_twice_in_one_backward_pass(dtype, wgrad_stream):
    """loss(model(x1)) + loss(model(x2)) under ONE backward(): every conv weight feeds two nodes of the pass and
    autograd adds their gradients as soon as the second arrives.  The deferred weight-gradient reduction must then
    have completed the first one (second stream or not): gradients equal the run with the deferral switched off
    bit for bit, and equal the sum of two separate passes within float32 round-off."""
    from medical_image_segmentation_amd import functional as F_
    from medical_image_segmentation_amd.utils.loss_fn import diceloss
    c = dict(B=2, cin=1, ncls=3, base=8, hw=32, seed=7)
    xs = [T(dg.hash_normalish((c["B"], c["cin"], c["hw"], c["hw"]), 4500 + k)).to(DEV) for k in range(2)]
    ts = [T(dg.blob_mask(c["B"], c["hw"], c["hw"], c["ncls"], 4600 + k)).to(DEV) for k in range(2)]
    names = ["a", "b"]
    saved = F_.WGRAD_SIDE_STREAM

    def grads(defer, together=True):
        F_.DEFER_WGRAD_REDUCE = defer
        m = build(c, dtype).eval()             # eval: BatchNorm uses running statistics, so the two calls are independent
        for p in m.parameters():
            p.requires_grad_(True)
        if together:
            (diceloss(names)(m(xs[0]), ts[0])["total_loss"] + diceloss(names)(m(xs[1]), ts[1])["total_loss"]).backward()
        else:
            for x, t in zip(xs, ts):
                diceloss(names)(m(x), t)["total_loss"].backward()
        torch.cuda.synchronize()
        assert not F_._wgrad_pending
        return {k: p.grad.clone() for k, p in m.named_parameters()}

    F_.WGRAD_SIDE_STREAM = wgrad_stream
    try:
        ref, got = grads(False), grads(True)
        sep = grads(False, together=False)
    finally:
        F_.WGRAD_SIDE_STREAM = saved
    for k in ref:
        assert torch.isfinite(got[k]).all(), k
        assert torch.equal(ref[k], got[k]), k
        assert rel_err(got[k], sep[k]) < (1e-5 if dtype == torch.float32 else 2e-2), k


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ncls", [2, 4, 8])
def test_head_gradient_formed_inside_last_layer_backward(dtype, ncls):
    """UNet.decoder4 -> out_conv: the 1x1 head's data gradient is not stored; decoder4's BatchNorm
    backward forms it from the head's output gradient.  Logits and every gradient equal the
    two-kernel form bit for bit; a second consumer of the fused layer's output is refused."""
    from medical_image_segmentation_amd import functional as F_
    from medical_image_segmentation_amd.utils.loss_fn import diceloss
    c = dict(B=2, cin=3, ncls=ncls, base=8, hw=32, seed=11)
    x = T(dg.hash_normalish((c["B"], c["cin"], c["hw"], c["hw"]), 5100)).to(DEV)
    t = T(dg.blob_mask(c["B"], c["hw"], c["hw"], c["ncls"], 5300)).to(DEV)
    names = [f"c{i}" for i in range(1, ncls)]

    # (both forms take the BatchNorm sums from the 1x1 backward kernel, whatever the tensor size:
    # a separate reduction pass would add in another order)
    F_.BN_SUMS_MIN_NUMEL = 0

    def run(fuse):
        F_.FUSE_HEAD_DZ = fuse
        m = build(c, dtype).train()
        logits = m(x)
        diceloss(names)(logits, t)["total_loss"].backward()
        torch.cuda.synchronize()
        return logits.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}

    l0, g0 = run(False)
    l1, g1 = run(True)
    assert torch.equal(l0, l1)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k

    # misuse: the fused layer's lazy output read by something besides the head
    F_.FUSE_HEAD_DZ = True
    m = build(c, dtype).train()
    head = F_.HeadFusion()
    xa = F_.to_act(x, dtype)
    z, ss = m.encoder1.forward_lazy(xa, head=head)
    w1 = torch.randn(ncls, z.shape[1], 1, 1, device=DEV, requires_grad=True)
    y = F_.Conv1x1Fn.apply(z, ss, w1, None, head)
    other = F_.materialize(z, ss)
    with pytest.raises(RuntimeError, match="consumer besides"):
        (y.float().sum() + other.float().sum()).backward()
