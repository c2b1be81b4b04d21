#!/usr/bin/env python3
"""Host time of one training step: the step on a tiny input (GPU work negligible), so that wall time per step is what
Python, ctypes and the launches cost.  Compare with the GPU time of the real step (4.4-4.7 ms): where the host needs
longer than the GPU, the step is host-bound.  usage: host_step_time.py [--dist]   (--dist: a one-rank RCCL group and
the gradient exchange of parallel.GradSync)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from medical_image_segmentation_amd.model.unet import UNet  # noqa: E402
from medical_image_segmentation_amd.parallel import GradSync  # noqa: E402
from medical_image_segmentation_amd.utils.loss_fn import diceloss  # noqa: E402
from medical_image_segmentation_amd.utils.metrics import Metrics  # noqa: E402
from medical_image_segmentation_amd.utils.optim import clip_and_step  # noqa: E402

use_dist = "--dist" in sys.argv
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
sync = None
model = UNet(1, 4, 0, base_channels=32, compute_dtype=torch.bfloat16).to(dev).train()
if use_dist:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    sync = GradSync(model, always_reduce=True)
opt = torch.optim.AdamW(model.parameters(), lr=8e-4)
loss_fn, metric = diceloss(["a", "b", "c"]), Metrics(["a", "b", "c"])
x = torch.randn(1, 1, 64, 64, device=dev)
t = torch.randint(0, 4, (1, 64, 64), device=dev)
params = list(model.parameters())


def step(parts=None):
    t0 = time.perf_counter()
    opt.zero_grad()
    logits = model(x)
    ld = loss_fn(logits, t)
    metric.confusion_counts(logits, t)
    t1 = time.perf_counter()
    if sync is not None:
        sync.begin_step()
    ld["total_loss"].backward()
    if sync is not None:
        sync.finish_step()
    t2 = time.perf_counter()
    clip_and_step(opt, params, max_norm=1.0)
    t3 = time.perf_counter()
    if parts is not None:
        parts[0] += t1 - t0
        parts[1] += t2 - t1
        parts[2] += t3 - t2


for _ in range(10):
    step()
torch.cuda.synchronize()
n = 200
parts = [0.0, 0.0, 0.0]
t0 = time.perf_counter()
for _ in range(n):
    step(parts)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"dist={use_dist} side_stream={os.environ.get('MIS_WGRAD_STREAM', '1')}: {dt / n * 1e3:.3f} ms per step on the host "
      f"(forward+loss {parts[0] / n * 1e3:.3f}, backward {parts[1] / n * 1e3:.3f}, clip+AdamW {parts[2] / n * 1e3:.3f})")
if use_dist:
    dist.destroy_process_group()
