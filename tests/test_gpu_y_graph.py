"""GPU suite: a training step captured in a HIP graph must be the eager step, bit for bit —
the same kernels run in the same order on the same values; only who issues them differs."""
import copy

import numpy as np
import pytest
import torch

from oracle import datagen as dg
from oracle import unet_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(base, seed, dtype):
    from medical_image_segmentation_amd.model.unet import UNet
    m = UNet(1, 4, 0, base_channels=base, compute_dtype=dtype)
    m.load_state_dict({k: v.clone() for k, v in orc.make_state(1, 4, base=base, seed=seed).items()})
    return m.to(DEV).train()


def _batches(n, B, hw):
    out = []
    for i in range(n):
        x = torch.from_numpy(dg.hash_normalish((B, 1, hw, hw), 700 + i)).float().to(DEV)
        y = torch.from_numpy((dg.hash_u64((B, hw, hw), 800 + i) % np.uint64(4)).astype(np.int64)).to(DEV)
        out.append((x, y))
    return out


@pytest.mark.parametrize("dtype,base,hw,B", [(torch.float32, 8, 32, 2), (torch.bfloat16, 32, 64, 2),
                                             (torch.bfloat16, 32, 64, 4)])
def test_graphed_step_equals_eager(dtype, base, hw, B):
    if not torch.cuda.is_available():
        pytest.fail("GPU suite needs a HIP device")
    from medical_image_segmentation_amd.utils.graph_step import GraphedTrainStep
    from medical_image_segmentation_amd.utils.loss_fn import diceloss
    from medical_image_segmentation_amd.utils.optim import clip_and_step
    loss_fn = diceloss(["a", "b", "c"])

    def loss_of(model, images, masks):
        ld = loss_fn(model(images), masks)
        return ld["total_loss"], ld

    batches = _batches(5, B, hw)
    opts = dict(lr=3e-3, betas=(0.9, 0.99), weight_decay=1e-2)
    # eager: two steps on batch 0 (the wrapper's warm-up), then batches 1..4, halving lr before the last
    ma = _model(base, 51, dtype)
    oa = torch.optim.AdamW(ma.parameters(), **opts)
    pa = list(ma.parameters())
    eager_losses = []
    for k, (x, y) in enumerate([batches[0], batches[0]] + batches[1:]):
        if k == 5:
            oa.param_groups[0]["lr"] *= 0.5
        oa.zero_grad()
        loss, _ = loss_of(ma, x, y)
        loss.backward()
        clip_and_step(oa, pa, 1.0)
        eager_losses.append(float(loss))
    # graphed
    mb = _model(base, 51, dtype)
    ob = torch.optim.AdamW(mb.parameters(), **opts)
    step = GraphedTrainStep(mb, ob, loss_of, batches[0][0], batches[0][1], max_norm=1.0, warmup=2)
    graph_losses = []
    for k, (x, y) in enumerate(batches[1:]):
        if k == 3:
            ob.param_groups[0]["lr"] *= 0.5
        loss, aux, norm = step(x, y)
        graph_losses.append(float(loss))
        assert float(norm) > 0 and set(aux) == {"total_loss", "a", "b", "c"} or len(aux) >= 1
    assert step.steps_taken == 6
    assert graph_losses == eager_losses[2:]
    for (n, a), b in zip(ma.state_dict().items(), mb.state_dict().values()):
        assert torch.equal(a, b), n
    for p, q in zip(ma.parameters(), mb.parameters()):
        sa, sb = oa.state[p], ob.state[q]
        assert float(sa["step"]) == float(sb["step"]) == 6.0
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])
    # back to eager steps on the graphed model: the caches notice the weights the graph wrote
    oa.zero_grad(); ob.zero_grad()
    la, _ = loss_of(ma, *batches[0]); lb, _ = loss_of(mb, *batches[0])
    assert float(la) == float(lb)
    la.backward(); lb.backward()
    clip_and_step(oa, pa, 1.0); clip_and_step(ob, list(mb.parameters()), 1.0)
    for a, b in zip(ma.parameters(), mb.parameters()):
        assert torch.equal(a, b)


def test_graphed_step_clears_the_counts_on_every_replay():
    """The confusion counts are accumulated with atomics into a buffer cleared at the start of the fused
    loss pass, so every replay of a captured step must clear them again.  (While hipMemsetAsync did the
    clearing, replays after the first returned garbage in the upper halves of the counts — and a Dice loss
    of exactly 1 per class — in some processes: tools/graph_debug.py showed it, this test's process did
    not; a kernel clears them now.)"""
    from medical_image_segmentation_amd.utils.graph_step import GraphedTrainStep
    from medical_image_segmentation_amd.utils.loss_fn import diceloss
    from medical_image_segmentation_amd.utils.metrics import Metrics
    loss_fn, metric = diceloss(["a", "b", "c"]), Metrics(["a", "b", "c"])
    B, hw = 4, 256
    (x, y), = _batches(1, B, hw)
    m = _model(32, 52, torch.bfloat16)
    o = torch.optim.AdamW(m.parameters(), lr=8e-4, betas=(0.95, 0.999), weight_decay=1e-4)

    def loss_of(model, images, masks):
        logits = model(images)
        ld = loss_fn(logits, masks)
        return ld["total_loss"], (ld, metric.confusion_counts(logits, masks), logits)

    step = GraphedTrainStep(m, o, loss_of, x, y, warmup=2)
    losses = []
    for _ in range(4):
        loss, (ld, (cm, status), logits), _ = step(step.images, step.masks)    # (replays back to back, no copy between)
        assert int(cm.sum()) == B * hw * hw and int(status) == 0
        want = metric.confusion_counts(logits.detach().clone(), y)[0]      # a fresh count of the same logits
        assert torch.equal(cm, want)
        losses.append(float(loss))
    assert all(0.0 < v < 0.75 for v in losses) and len(set(losses)) == 4


def test_graphed_step_rejects_other_shapes():
    from medical_image_segmentation_amd.utils.graph_step import GraphedTrainStep
    from medical_image_segmentation_amd.utils.loss_fn import diceloss
    loss_fn = diceloss(["a", "b", "c"])
    m = _model(8, 52, torch.float32)
    o = torch.optim.AdamW(m.parameters(), lr=1e-3)
    (x, y), = _batches(1, 2, 32)
    step = GraphedTrainStep(m, o, lambda mm, a, b: (loss_fn(mm(a), b)["total_loss"], None), x, y, warmup=1)
    with pytest.raises(ValueError):
        step(x[:1], y[:1])


def test_graphed_step_with_dropout_draws_new_masks():
    """Dropout2d inside the captured step: torch's generator advances with every replay, so two
    replays on the same batch drop different feature maps (different losses), and training mode
    keeps working."""
    from medical_image_segmentation_amd.model.unet import UNet
    from medical_image_segmentation_amd.utils.graph_step import GraphedTrainStep
    from medical_image_segmentation_amd.utils.loss_fn import diceloss
    loss_fn = diceloss(["a", "b", "c"])
    torch.manual_seed(5)
    m = UNet(1, 4, 0.3, base_channels=8, compute_dtype=torch.float32)
    m.load_state_dict({k: v.clone() for k, v in orc.make_state(1, 4, base=8, seed=53).items()})
    m = m.to(DEV).train()
    o = torch.optim.AdamW(m.parameters(), lr=0.0)            # lr 0: the weights stay, only the masks change
    (x, y), = _batches(1, 2, 32)
    step = GraphedTrainStep(m, o, lambda mm, a, b: (loss_fn(mm(a), b)["total_loss"], None), x, y, warmup=1)
    losses = [float(step(x, y)[0]) for _ in range(4)]
    assert all(np.isfinite(losses)) and len(set(losses)) > 1
