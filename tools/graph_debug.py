"""Replays a captured training step a few times and reports the loss and the first non-finite tensors."""
import sys

import torch

sys.path.insert(0, ".")
from medical_image_segmentation_amd import _lib  # noqa: E402
from medical_image_segmentation_amd.model.unet import UNet  # noqa: E402
from medical_image_segmentation_amd.utils.graph_step import GraphedTrainStep  # noqa: E402
from medical_image_segmentation_amd.utils.loss_fn import diceloss  # noqa: E402
from medical_image_segmentation_amd.utils.metrics import Metrics  # noqa: E402

_lib.load()
B, S = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = UNet(1, 4, 0, base_channels=32, compute_dtype=torch.bfloat16).to(dev).train()
opt = torch.optim.AdamW(model.parameters(), lr=8e-4, betas=(0.95, 0.999), weight_decay=1e-4)
loss_fn, metric = diceloss(["a", "b", "c"]), Metrics(["a", "b", "c"])
images = torch.randn(B, 1, S, S).to(dev)
masks = torch.randint(0, 4, (B, S, S)).to(dev)


def loss_of(m, x, y):
    logits = m(x)
    ld = loss_fn(logits, y)
    return ld["total_loss"], (ld, metric.confusion_counts(logits, y), logits)


g = GraphedTrainStep(model, opt, loss_of, images, masks, max_norm=1.0, warmup=2)
for it in range(4):
    loss, (ld, counts, logits), norm = g(g.images, g.masks)
    torch.cuda.synchronize()
    bad = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
    badg = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print(it, {k: round(float(v), 5) for k, v in ld.items()}, "counts", counts[0].sum().item(), counts[0][0].flatten().tolist(),
          "logits mean/std", float(logits.float().mean()), float(logits.float().std()))
    print(it, "loss", float(loss), "norm", float(norm), "logits finite", bool(torch.isfinite(logits.float()).all()),
          "bad params", bad[:4], "bad grads", badg[:6])

