"""Which host-side torch ops launch the small fill/copy kernels of a training step: runs a few
steps under torch.profiler and prints every aten fill/copy/zero op with its input shapes and the
innermost Python frames."""
import collections
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, ".")
from medical_image_segmentation_amd import _lib  # noqa: E402
from medical_image_segmentation_amd.model.unet import UNet  # noqa: E402
from medical_image_segmentation_amd.utils.loss_fn import diceloss  # noqa: E402
from medical_image_segmentation_amd.utils.metrics import Metrics  # noqa: E402
from medical_image_segmentation_amd.utils.optim import clip_and_step  # noqa: E402

_lib.load()
dev = torch.device("cuda:0")
model = UNet(1, 4, 0, base_channels=32, compute_dtype=torch.bfloat16).to(dev).train()
opt = torch.optim.AdamW(model.parameters(), lr=8e-4, betas=(0.95, 0.999), weight_decay=1e-4)
loss_fn, metric = diceloss(["a", "b", "c"]), Metrics(["a", "b", "c"])
images = torch.randn(4, 1, 128, 128).to(dev)
masks = torch.randint(0, 4, (4, 128, 128)).to(dev)
params = list(model.parameters())


def step():
    opt.zero_grad()
    logits = model(images)
    ld = loss_fn(logits, masks)
    metric.confusion_counts(logits, masks)
    ld["total_loss"].backward()
    clip_and_step(opt, params, max_norm=1.0)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
seen = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::cat", "aten::_foreach_copy_"):
        frames = [f for f in ev.stack if "site-packages/torch" not in f and "<built-in" not in f][:3]
        seen[(ev.name, str(ev.input_shapes), " <- ".join(frames))] += 1
for (name, shapes, where), n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(n, name, shapes, where)
