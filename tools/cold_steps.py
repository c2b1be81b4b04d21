"""Wall time of every training step (synchronised one by one) from the start of a process: shows
whether a slow first run on a box is a uniformly slow GPU or single stalls."""
import gc
import os
import sys
import time

import torch

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
images = torch.randn(16, 1, 512, 512).to(dev)
masks = torch.randint(0, 4, (16, 512, 512)).to(dev)
params = list(model.parameters())
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
sync_every = int(sys.argv[2]) if len(sys.argv) > 2 else 1
times = []
issue = []
torch.cuda.synchronize()
t_prev = time.perf_counter()
for i in range(n):
    opt.zero_grad()
    logits = model(images)
    ld = loss_fn(logits, masks)
    metric.confusion_counts(logits, masks)
    ld["total_loss"].backward()
    clip_and_step(opt, params, max_norm=1.0)
    if i == 3 and os.environ.get("COLD_GC_FREEZE"):
        gc.collect()
        gc.freeze()
    issue.append((time.perf_counter() - t_prev) * 1e3)
    if (i + 1) % sync_every == 0:
        torch.cuda.synchronize()
        t = time.perf_counter()
        times.append((t - t_prev) * 1e3 / sync_every)
        t_prev = t
print("step ms :", " ".join("%.2f" % t for t in times))
if sync_every == 1:
    print("issue ms:", " ".join("%.2f" % t for t in issue))
    print("gc:", gc.get_stats())
    print("allocator:", torch.cuda.memory_stats()["num_device_alloc"], "device allocations,",
          torch.cuda.memory_stats()["num_alloc_retries"], "retries")
