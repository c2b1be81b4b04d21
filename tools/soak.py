#!/usr/bin/env python3
"""Stability run of the benchmark step: many steps on changing batches, watching the loss, the
allocator's reserved memory and the parameters for non-finite values.   python tools/soak.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from medical_image_segmentation_amd.model.unet import UNet  # noqa: E402
from medical_image_segmentation_amd.utils.loss_fn import diceloss  # noqa: E402
from medical_image_segmentation_amd.utils.optim import clip_and_step  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = UNet(1, 4, 0.1, base_channels=32, compute_dtype=torch.bfloat16).to(dev).train()
opt = torch.optim.AdamW(m.parameters(), lr=8e-4, betas=(0.95, 0.999), weight_decay=1e-4)
loss_fn = diceloss(["a", "b", "c"])
params = list(m.parameters())
g = torch.Generator(device=dev).manual_seed(1)
# a learnable synthetic task: the class is a function of the local mean intensity
t0 = time.perf_counter()
first = last = None
for k in range(steps):
    x = torch.randn(16, 1, 512, 512, device=dev, generator=g)
    xs = torch.nn.functional.avg_pool2d(x, 9, 1, 4)
    y = torch.bucketize(xs[:, 0], torch.tensor([-0.1, 0.0, 0.1], device=dev)).long()
    opt.zero_grad()
    loss = loss_fn(m(x), y)["total_loss"]
    loss.backward()
    clip_and_step(opt, params, 1.0)
    if k % 50 == 0 or k == steps - 1:
        lv = float(loss.detach())
        first = lv if first is None else first
        last = lv
        print(f"step {k:4d} loss {lv:.4f} reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB", flush=True)
        assert lv == lv, "loss is NaN"
torch.cuda.synchronize()
assert all(torch.isfinite(p).all() for p in params)
print(f"{steps} steps in {time.perf_counter() - t0:.1f} s; loss {first:.4f} -> {last:.4f}; parameters finite")
