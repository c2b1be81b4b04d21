#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running the REFERENCE implementation itself.

Runs only in the build container (needs /root/reference; CPU only).  Inputs and
weights come from oracle/datagen.py (integer-hash generator), so the fixtures
store nothing but the reference's outputs; the parity tests regenerate the same
inputs and compare the oracle (CPU suite) and the HIP path (GPU suite) to them.

Two local accommodations are needed to import the reference here, neither of
which touches arithmetic on the path:
  * inert placeholder modules for third-party imports that are not installed and
    not used by UNet / diceloss / Focal_Loss / Metrics / train_one_epoch /
    SlidingWindowPredictor (torchinfo, tensorboardX, thop, mamba_ssm,
    torchvision.ops, monai.losses);
  * DoubleConv._initialize_weights (and DWConv's, AMSFN's) is replaced by a no-op
    (its last `if m.bias` is mis-indented and raises AttributeError on
    construction, model/utils/modules.py:115,76,884); all weights are then loaded
    explicitly.

usage:  python tests/golden/make_golden.py  [--ref /root/reference]
"""
import argparse
import os
import sys
import tempfile
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import datagen as dg            # noqa: E402
from oracle import unet_oracle as orc       # noqa: E402


def import_reference(ref_root):
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class Inert:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            raise RuntimeError("inert placeholder called")

    stub("torchinfo", summary=lambda *a, **k: None)
    stub("tensorboardX", SummaryWriter=Inert)
    stub("thop", profile=lambda *a, **k: (0, 0))
    stub("mamba_ssm", Mamba=Inert)
    tv = stub("torchvision")
    tv.ops = stub("torchvision.ops", DeformConv2d=Inert)
    mo = stub("monai")
    mo.losses = stub("monai.losses", DiceLoss=Inert, HausdorffDTLoss=Inert)
    sys.path.insert(0, ref_root)
    from model.utils import modules as M
    M.DoubleConv._initialize_weights = lambda self, init_gain=0.02: None
    # the same mis-indented `if m.bias` ends DWConv's and AMSFN's initialisers (modules.py:76, 884)
    M.DWConv._initialize_weights = lambda self, init_gain=0.02: None
    M.AMSFN._initialize_weights = lambda self, init_gain=0.02: None
    M.ResDConv._initialize_weights = lambda self, init_gain=0.02: None      # same defect, modules.py:470
    from model.utils import attention as A
    from model.unet import UNet
    from model.a_unet import A_UNet
    from model.rdam_unet import RDAM_UNet
    from utils.loss_fn import diceloss, Focal_Loss, IOULoss
    from utils.metrics import Metrics
    from utils.train_and_eval import train_one_epoch, evaluate
    from utils.slide_predict import SlidingWindowPredictor
    return types.SimpleNamespace(DoubleConv=M.DoubleConv, UNet=UNet, A_UNet=A_UNet, AMSFN=M.AMSFN, diceloss=diceloss,
                                 RDAM_UNet=RDAM_UNet, ResDConv=M.ResDConv, MDAM=A.MDAM,
                                 Focal_Loss=Focal_Loss, IOULoss=IOULoss, Metrics=Metrics,
                                 train_one_epoch=train_one_epoch, evaluate=evaluate,
                                 SlidingWindowPredictor=SlidingWindowPredictor)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def sample_idx(n, k=16):
    """Deterministic sample positions inside a flat tensor of n elements."""
    return (np.arange(k, dtype=np.int64) * 2654435761 + 12345) % n


# ---------------------------------------------------------------------------
def gold_double_conv(R, out):
    cases = {"dc_a": dict(B=2, cin=3, mid=8, cout=8, hw=16, seed=11),
             "dc_b": dict(B=2, cin=16, mid=None, cout=16, hw=8, seed=12)}
    for name, c in cases.items():
        m = R.DoubleConv(c["cin"], c["cout"], c["mid"])
        mid = c["mid"] if c["mid"] is not None else c["cout"] // 4
        sd = orc_double_conv_state(c["cin"], mid, c["cout"], c["seed"])
        m.load_state_dict({k[len("blk."):]: v.clone() for k, v in sd.items()})
        x = T(dg.hash_normalish((c["B"], c["cin"], c["hw"], c["hw"]), c["seed"] + 100)).requires_grad_(True)
        gy = T(dg.hash_normalish((c["B"], c["cout"], c["hw"], c["hw"]), c["seed"] + 200))
        m.train()
        y = m(x)
        y.backward(gy)
        out[f"{name}.train_y"] = y.detach().numpy()
        out[f"{name}.dx"] = x.grad.numpy()
        for k, p in m.named_parameters():
            out[f"{name}.grad.{k}"] = p.grad.numpy()
        for k, b in m.named_buffers():
            out[f"{name}.buf.{k}"] = b.detach().numpy()
        m.eval()
        with torch.no_grad():
            out[f"{name}.eval_y"] = m(x).numpy()


def orc_double_conv_state(cin, mid, cout, seed):
    """State for a single DoubleConv under prefix 'blk' (same generator as make_state)."""
    import math
    from collections import OrderedDict
    st = OrderedDict()
    k = 0
    for j, (a, b_) in enumerate(((cin, mid), (mid, cout)), start=1):
        p = f"blk.cbr{j}"
        std = math.sqrt(2.0 / (b_ * 9))
        st[p + ".0.weight"] = T(dg.hash_normalish((b_, a, 3, 3), seed * 1000 + k, std)); k += 1
        st[p + ".0.bias"] = T(dg.hash_uniform((b_,), seed * 1000 + k, -0.1, 0.1)); k += 1
        st[p + ".1.weight"] = T(dg.hash_uniform((b_,), seed * 1000 + k, 0.5, 1.5)); k += 1
        st[p + ".1.bias"] = T(dg.hash_uniform((b_,), seed * 1000 + k, -0.2, 0.2)); k += 1
        st[p + ".1.running_mean"] = torch.zeros(b_)
        st[p + ".1.running_var"] = torch.ones(b_)
        st[p + ".1.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    return st


def gold_unet(R, out):
    cases = {"unet_a": dict(B=2, cin=1, ncls=4, hw=32, seed=21),
             "unet_b": dict(B=2, cin=3, ncls=4, hw=64, seed=22),
             "unet_c": dict(B=1, cin=1, ncls=9, hw=32, seed=23),
             # seeds whose gradients do not hinge on a ReLU within round-off of zero (tools/find_stable_unet_seed.py:
             # the oracle in float32 and in float64 agree to 5e-6 on every tensor; with seed 22 they differ by 3e-2)
             "unet_d": dict(B=2, cin=3, ncls=4, hw=64, seed=37),
             "unet_e": dict(B=2, cin=3, ncls=4, hw=64, seed=42)}
    for name, c in cases.items():
        m = R.UNet(c["cin"], c["ncls"], 0, base_channels=8)
        st = orc.make_state(c["cin"], c["ncls"], base=8, seed=c["seed"])
        m.load_state_dict({k: v.clone() for k, v in st.items()})
        assert list(m.state_dict().keys()) == list(st.keys())
        x = T(dg.hash_normalish((c["B"], c["cin"], c["hw"], c["hw"]), c["seed"] + 100))
        t = T(dg.blob_mask(c["B"], c["hw"], c["hw"], c["ncls"], c["seed"] + 300))
        names = [f"c{i}" for i in range(1, c["ncls"])]
        m.train()
        logits = m(x)
        ld = R.diceloss(names)(logits, t)
        ld["total_loss"].backward()
        out[f"{name}.logits"] = logits.detach().numpy()
        out[f"{name}.loss"] = np.array([float(v) for v in ld.values()], dtype=np.float64)
        keys, norms, samples = [], [], []
        for k, p in m.named_parameters():
            g = p.grad.detach().numpy().reshape(-1)
            keys.append(k)
            norms.append(np.sqrt((g.astype(np.float64) ** 2).sum()))
            samples.append(g[sample_idx(g.size)])
        out[f"{name}.grad_keys"] = np.array(keys)
        out[f"{name}.grad_norms"] = np.array(norms)
        out[f"{name}.grad_samples"] = np.stack(samples)
        out[f"{name}.grad.out_conv.weight"] = m.out_conv.weight.grad.numpy()
        out[f"{name}.grad.encoder1.cbr1.0.weight"] = m.encoder1.cbr1[0].weight.grad.numpy()
        out[f"{name}.running_mean.center_conv.cbr1"] = m.center_conv.cbr1[1].running_mean.numpy().copy()
        out[f"{name}.running_var.center_conv.cbr1"] = m.center_conv.cbr1[1].running_var.numpy().copy()
        m.eval()
        with torch.no_grad():
            out[f"{name}.eval_logits"] = m(x).numpy()


def gold_pool_up(R, out):
    m = R.UNet(1, 2, 0, base_channels=8)
    # max-pool with exact ties: post-ReLU style data (many zeros) + a constant block
    x = dg.hash_normalish((2, 3, 8, 12), 31)
    x = np.maximum(x, 0)
    x[0, 0, :4, :4] = 0.5
    x = T(x).requires_grad_(True)
    gy = T(dg.hash_normalish((2, 3, 4, 6), 32))
    y = m.dowm(x)
    y.backward(gy)
    out["pool.y"] = y.detach().numpy()
    out["pool.dx"] = x.grad.numpy()
    for name, (h, w) in {"up_2": (2, 2), "up_5": (5, 5), "up_3x7": (3, 7)}.items():
        x = T(dg.hash_normalish((2, 3, h, w), 33 + h)).requires_grad_(True)
        skip = T(dg.hash_normalish((2, 2, 2 * h, 2 * w), 43 + h)).requires_grad_(True)
        y = torch.cat([m.up(x), skip], dim=1)          # unet.py:66-67 order: upsampled first
        gy = T(dg.hash_normalish(tuple(y.shape), 53 + h))
        y.backward(gy)
        out[f"{name}.y"] = y.detach().numpy()
        out[f"{name}.dx"] = x.grad.numpy()
        out[f"{name}.dskip"] = skip.grad.numpy()


def loss_cases():
    return {"loss_c4": dict(B=2, C=4, hw=16, seed=51, absent=None),
            "loss_c9": dict(B=2, C=9, hw=16, seed=52, absent=None),
            "loss_absent": dict(B=3, C=4, hw=16, seed=53, absent=2),
            "loss_allbg": dict(B=1, C=4, hw=16, seed=54, absent="all")}


def loss_inputs(c):
    x = dg.hash_normalish((c["B"], c["C"], c["hw"], c["hw"]), c["seed"], std=2.0)
    t = dg.blob_mask(c["B"], c["hw"], c["hw"], c["C"], c["seed"] + 1)
    if c["absent"] == "all":
        t[:] = 0
    elif c["absent"] is not None:
        t[t == c["absent"]] = 0
    return x, t


def gold_loss(R, out):
    for name, c in loss_cases().items():
        x, t = loss_inputs(c)
        names = [f"c{i}" for i in range(1, c["C"])]
        xl = T(x).requires_grad_(True)
        ld = R.diceloss(names)(xl, T(t))
        assert list(ld.keys())[-1] == "total_loss"
        ld["total_loss"].backward()
        out[f"{name}.dice"] = np.array([float(v) for v in ld.values()], dtype=np.float64)
        out[f"{name}.dice_dlogits"] = xl.grad.numpy()
        li = R.IOULoss(names)(T(x), T(t))
        out[f"{name}.iou"] = np.array([float(v) for v in li.values()], dtype=np.float64)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            fl = R.Focal_Loss(names)(T(x), T(t))    # pins the CE map of loss_fn.py:199-203
        out[f"{name}.focal"] = np.array([float(v) for v in fl.values()], dtype=np.float64)
        # the CE expression itself, exactly as written at loss_fn.py:199,203
        ce = torch.nn.CrossEntropyLoss(label_smoothing=1e-7, reduction="none", ignore_index=0)
        out[f"{name}.ce_map"] = ce(T(x), T(t)).numpy()


def metric_cases():
    return {"met_n3": dict(B=3, C=4, n=3, hw=16, seed=61, mode="plain"),
            "met_n8": dict(B=2, C=9, n=8, hw=16, seed=62, mode="plain"),
            "met_allbg": dict(B=2, C=4, n=3, hw=16, seed=63, mode="allbg"),
            "met_nopred": dict(B=2, C=4, n=3, hw=16, seed=64, mode="nopred"),
            "met_ties": dict(B=2, C=4, n=3, hw=16, seed=65, mode="ties")}


def metric_inputs(c):
    x = dg.hash_normalish((c["B"], c["C"], c["hw"], c["hw"]), c["seed"], std=2.0)
    t = dg.blob_mask(c["B"], c["hw"], c["hw"], c["C"], c["seed"] + 1)
    if c["mode"] == "allbg":
        t[:] = 0
    elif c["mode"] == "nopred":
        x[:, 2] = -50.0                       # class 2 is never the argmax
    elif c["mode"] == "ties":
        x = np.round(x)                       # many exact ties between classes
        x[0, :, :4, :] = 1.0                  # all-equal logits -> class 0
    return x, t


def gold_metrics(R, out):
    for name, c in metric_cases().items():
        x, t = metric_inputs(c)
        names = [f"c{i}" for i in range(1, c["n"] + 1)]
        M = R.Metrics(names)
        out[f"{name}.matrix"] = M.update(T(x), T(t))
        # the integer counts of metrics.py:31-34 on the (n+1)-wide one-hot
        pred = torch.argmax(T(x), dim=1)
        K = c["n"] + 1
        p1 = torch.nn.functional.one_hot(pred, K).permute(0, 3, 1, 2).float()
        t1 = torch.nn.functional.one_hot(T(t), K).permute(0, 3, 1, 2).float()
        TP, FN, FP, TN = M.compute_confusion_matrix(p1, t1)
        out[f"{name}.TP"], out[f"{name}.FN"] = TP.numpy(), FN.numpy()
        out[f"{name}.FP"], out[f"{name}.TN"] = FP.numpy(), TN.numpy()


class ListLoader(list):
    """A list of batches in the reference's loader format: data[0][0]=images,
    data[0][1]=masks, data[1]=names (train_and_eval.py:65)."""


def train_case():
    return dict(B=2, cin=1, ncls=4, hw=32, seed=71, steps=3, lr=8e-4)


def train_batches(c):
    out = []
    for s in range(c["steps"]):
        x = dg.hash_normalish((c["B"], c["cin"], c["hw"], c["hw"]), c["seed"] + 10 * s)
        t = dg.blob_mask(c["B"], c["hw"], c["hw"], c["ncls"], c["seed"] + 10 * s + 1)
        out.append((x, t))
    return out


def gold_train(R, out):
    c = train_case()
    m = R.UNet(c["cin"], c["ncls"], 0, base_channels=8)
    st = orc.make_state(c["cin"], c["ncls"], base=8, seed=c["seed"])
    m.load_state_dict({k: v.clone() for k, v in st.items()})
    names = ["a", "b", "c"]
    opt = torch.optim.AdamW(m.parameters(), lr=c["lr"], betas=(0.95, 0.999), weight_decay=1e-4)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        scaler = torch.amp.GradScaler("cuda")          # self-disables without CUDA
        loader = ListLoader([((T(x), T(t)), [f"img{i}"]) for i, (x, t) in enumerate(train_batches(c))])
        losses, mets = R.train_one_epoch(m, opt, 0, loader, torch.device("cpu"), R.diceloss(names), scaler,
                                         R.Metrics(names), sched, names, False, 0.0, 0.0)
        out["train.losses"] = np.array([float(v) for v in losses], dtype=np.float64)
        out["train.metrics"] = np.asarray(mets)
        sd = m.state_dict()
        out["train.keys"] = np.array(list(sd.keys()))
        out["train.param_sums"] = np.array([float(v.double().sum()) for v in sd.values()])
        out["train.param_abs_sums"] = np.array([float(v.double().abs().sum()) for v in sd.values()])
        out["train.out_conv.weight"] = sd["out_conv.weight"].numpy().copy()
        out["train.encoder1.cbr1.0.weight"] = sd["encoder1.cbr1.0.weight"].numpy().copy()
        elosses, emets = R.evaluate(m, torch.device("cpu"), loader, R.diceloss(names), R.Metrics(names), names)
        out["eval.losses"] = np.array([float(v) for v in elosses], dtype=np.float64)
        out["eval.metrics"] = np.asarray(emets)


def gold_dp(R, out):
    """Two reference replicas, each on half the batch, gradients averaged: the
    data-parallel parity target (BatchNorm statistics stay replica-local)."""
    c = dict(B=4, cin=1, ncls=4, hw=32, seed=81)
    st = orc.make_state(c["cin"], c["ncls"], base=8, seed=c["seed"])
    x = dg.hash_normalish((c["B"], c["cin"], c["hw"], c["hw"]), c["seed"] + 100)
    t = dg.blob_mask(c["B"], c["hw"], c["hw"], c["ncls"], c["seed"] + 300)
    names = ["a", "b", "c"]
    grads = None
    losses = []
    for r in range(2):
        m = R.UNet(c["cin"], c["ncls"], 0, base_channels=8)
        m.load_state_dict({k: v.clone() for k, v in st.items()})
        m.train()
        sl = slice(r * 2, r * 2 + 2)
        ld = R.diceloss(names)(m(T(x[sl])), T(t[sl]))
        ld["total_loss"].backward()
        losses.append(float(ld["total_loss"]))
        g = [p.grad.detach().double().numpy() for p in m.parameters()]
        grads = g if grads is None else [a + b for a, b in zip(grads, g)]
    grads = [g / 2 for g in grads]
    out["dp.shard_losses"] = np.array(losses)
    out["dp.grad_norms"] = np.array([np.sqrt((g ** 2).sum()) for g in grads])
    out["dp.grad_samples"] = np.stack([g.reshape(-1)[sample_idx(g.size)] for g in grads])


class StubSegModel(torch.nn.Module):
    """Deterministic per-pixel 'model' for the sliding-window fixture: class logits
    are fixed linear functions of the RGB pixel, so a tile's class map depends only
    on the pixels (not on tile position)."""

    def __init__(self, n_classes=4):
        super().__init__()
        w = dg.hash_normalish((n_classes, 3), 91, std=4.0)
        self.register_buffer("w", T(w))

    def forward(self, x):
        return torch.einsum("kc,bchw->bkhw", self.w, x)


def slide_cases():
    return {"slide_a": dict(h=40, w=56, window=16, stride=8, seed=92),
            "slide_b": dict(h=37, w=45, window=16, stride=8, seed=93),
            "slide_c": dict(h=64, w=64, window=32, stride=16, seed=94)}


def gold_slide(R, out):
    from PIL import Image
    model = StubSegModel()
    for name, c in slide_cases().items():
        img = (dg.hash_u64((c["h"], c["w"], 3), c["seed"]) % np.uint64(256)).astype(np.uint8)
        pred = R.SlidingWindowPredictor(model, torch.device("cpu"), c["window"], c["stride"])
        padded, pads = pred._pad_image(img)
        out[f"{name}.pads"] = np.array(pads)
        out[f"{name}.positions"] = np.array(pred._split_image(padded), dtype=np.int64)
        with tempfile.TemporaryDirectory() as d:
            src, dst = os.path.join(d, "in.png"), os.path.join(d, "out.png")
            Image.fromarray(img).save(src)
            pred.predict(src, dst)
            out[f"{name}.result"] = np.array(Image.open(dst))
    for (h, w, win, s) in [(4096, 4096, 1024, 512), (4096, 4096, 256, 128), (300, 500, 256, 128)]:
        pred = R.SlidingWindowPredictor(None, None, win, s)
        padded, pads = pred._pad_image(np.zeros((h, w, 1), dtype=np.uint8))
        pos = np.array(pred._split_image(padded), dtype=np.int64)
        out[f"slide_pos_{h}x{w}_{win}_{s}.positions"] = pos
        out[f"slide_pos_{h}x{w}_{win}_{s}.pads"] = np.array(pads)


def a_unet_cases():
    return {"aunet_a": dict(B=2, cin=3, ncls=4, base=8, hw=32, seed=41),
            "aunet_b": dict(B=1, cin=1, ncls=5, base=16, hw=48, seed=42)}


def gold_a_unet(R, out):
    """A_UNet (model/a_unet.py) and a stand-alone AMSFN block: parameters from the hash generator
    in state_dict order (oracle.hash_state), train-mode forward + Dice loss + backward."""
    for name, c in a_unet_cases().items():
        m = R.A_UNet(c["cin"], c["ncls"], 0, base_channels=c["base"])
        spec = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
        st = orc.hash_state(spec, c["seed"])
        m.load_state_dict({k: v.clone() for k, v in st.items()})
        m.train()
        x = T(dg.hash_normalish((c["B"], c["cin"], c["hw"], c["hw"]), c["seed"] + 100))
        t = T(dg.blob_mask(c["B"], c["hw"], c["hw"], c["ncls"], c["seed"] + 300))
        names = [f"c{i}" for i in range(1, c["ncls"])]
        logits = m(x)
        ld = R.diceloss(names)(logits, t)
        ld["total_loss"].backward()
        out[f"{name}.keys"] = np.array([k for k, _ in spec])
        out[f"{name}.shapes"] = np.array([",".join(map(str, s)) for _, s in spec])
        out[f"{name}.logits"] = logits.detach().numpy()
        out[f"{name}.loss"] = np.array([float(v) for v in ld.values()], dtype=np.float64)
        params = dict(m.named_parameters())
        out[f"{name}.grad_keys"] = np.array(list(params.keys()))
        out[f"{name}.grad_norms"] = np.array([float(p.grad.double().norm()) for p in params.values()])
        out[f"{name}.running_mean.acpn0.final_conv"] = m.acpn0.final_conv[1].running_mean.numpy().copy()
        m.eval()
        with torch.no_grad():
            out[f"{name}.eval_logits"] = m(x).numpy()
    # one block on its own, with input gradient
    blk = R.AMSFN(8)
    spec = [(k, tuple(v.shape)) for k, v in blk.state_dict().items()]
    st = orc.hash_state(spec, 43)
    blk.load_state_dict({k: v.clone() for k, v in st.items()})
    blk.train()
    x = T(np.maximum(dg.hash_normalish((2, 8, 12, 20), 143), 0)).requires_grad_(True)     # post-ReLU like input
    gy = T(dg.hash_normalish((2, 8, 12, 20), 243))
    y = blk(x)
    y.backward(gy)
    out["amsfn.keys"] = np.array([k for k, _ in spec])
    out["amsfn.shapes"] = np.array([",".join(map(str, s)) for _, s in spec])
    out["amsfn.y"] = y.detach().numpy()
    out["amsfn.dx"] = x.grad.numpy()
    for k, p in blk.named_parameters():
        out[f"amsfn.grad.{k}"] = p.grad.numpy()


def rdam_cases():
    return {"rdam_a": dict(B=2, cin=3, ncls=4, base=32, hw=32, seed=61)}


def rdam_block_cases():
    """stand-alone blocks: (kind, constructor args, input shape, seed)"""
    return {"resd_a": ("resd", (32, 64), (2, 32, 20, 12), 71), "resd_b": ("resd", (64, 32), (1, 64, 9, 17), 72),
            "mdam_a": ("mdam", (32,), (2, 32, 12, 20), 73), "mdam_b": ("mdam", (128,), (1, 128, 7, 5), 74)}


def _block_fixture(blk, shape, seed, name, out, positive_input):
    spec = [(k, tuple(v.shape)) for k, v in blk.state_dict().items()]
    st = orc.hash_state(spec, seed)
    blk.load_state_dict({k: v.clone() for k, v in st.items()})
    blk.train()
    x = dg.hash_normalish(shape, seed + 100)
    x = T(np.maximum(x, 0) if positive_input else x).requires_grad_(True)
    gy = T(dg.hash_normalish(shape[:1] + (blk_out_channels(blk, shape),) + shape[2:], seed + 200))
    y = blk(x)
    y.backward(gy)
    out[f"{name}.keys"] = np.array([k for k, _ in spec])
    out[f"{name}.shapes"] = np.array([",".join(map(str, s)) for _, s in spec])
    out[f"{name}.y"] = y.detach().numpy()
    out[f"{name}.dx"] = x.grad.numpy()
    for k, p in blk.named_parameters():
        out[f"{name}.grad.{k}"] = p.grad.numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
    for k, b in blk.named_buffers():
        out[f"{name}.buf.{k}"] = b.detach().numpy().copy()


def blk_out_channels(blk, shape):
    return blk.c1[0].out_channels if hasattr(blk, "c1") else shape[1]


def gold_rdam(R, out):
    """RDAM_UNet (model/rdam_unet.py) and its two new blocks on their own (ResDConv with input
    and parameter gradients and updated running statistics; MDAM likewise): parameters from the
    hash generator in state_dict order, train-mode forward + backward."""
    for name, (kind, args, shape, seed) in rdam_block_cases().items():
        blk = R.ResDConv(*args) if kind == "resd" else R.MDAM(*args)
        _block_fixture(blk, shape, seed, name, out, positive_input=(kind == "resd"))
    for name, c in rdam_cases().items():
        m = R.RDAM_UNet(c["cin"], c["ncls"], 0, base_channels=c["base"])
        spec = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
        st = orc.hash_state(spec, c["seed"])
        m.load_state_dict({k: v.clone() for k, v in st.items()})
        m.train()
        x = T(dg.hash_normalish((c["B"], c["cin"], c["hw"], c["hw"]), c["seed"] + 100))
        t = T(dg.blob_mask(c["B"], c["hw"], c["hw"], c["ncls"], c["seed"] + 300))
        names = [f"c{i}" for i in range(1, c["ncls"])]
        logits = m(x)
        ld = R.diceloss(names)(logits, t)
        ld["total_loss"].backward()
        out[f"{name}.keys"] = np.array([k for k, _ in spec])
        out[f"{name}.shapes"] = np.array([",".join(map(str, s)) for _, s in spec])
        out[f"{name}.logits"] = logits.detach().numpy()
        out[f"{name}.loss"] = np.array([float(v) for v in ld.values()], dtype=np.float64)
        params = dict(m.named_parameters())
        out[f"{name}.grad_keys"] = np.array(list(params.keys()))
        out[f"{name}.grad_norms"] = np.array([float(p.grad.double().norm()) if p.grad is not None else 0.0
                                              for p in params.values()])
        out[f"{name}.grad_samples"] = np.stack([
            (p.grad if p.grad is not None else torch.zeros_like(p)).detach().numpy().reshape(-1)[sample_idx(p.numel())]
            for p in params.values()])
        out[f"{name}.running_mean.encoder3.cbr3"] = m.encoder3.cbr3[1].running_mean.numpy().copy()
        m.eval()
        with torch.no_grad():
            out[f"{name}.eval_logits"] = m(x).numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default="", help="comma-separated fixture groups (default: all)")
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(1)            # fixtures must not depend on thread count
    R = import_reference(args.ref)
    groups = {"double_conv": gold_double_conv, "unet": gold_unet, "pool_up": gold_pool_up,
              "loss": gold_loss, "metrics": gold_metrics, "train": gold_train, "dp": gold_dp,
              "slide": gold_slide, "a_unet": gold_a_unet, "rdam": gold_rdam}
    only = [g for g in args.only.split(",") if g]
    for gname, fn in groups.items():
        if only and gname not in only:
            continue
        out = {}
        fn(R, out)
        path = os.path.join(HERE, f"{gname}.npz")
        np.savez_compressed(path, **out)
        print(f"{gname}: {len(out)} arrays -> {path} ({os.path.getsize(path)} bytes)")


if __name__ == "__main__":
    main()
