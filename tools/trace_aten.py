#!/usr/bin/env python3
"""Counts the ATen ops one bench-style training step issues (besides the library's own kernels)
and where they come from: finds stray fills/copies.  usage: python tools/trace_aten.py [unet|a_unet|rdam_unet]"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.ops = collections.Counter()
        self.where = collections.defaultdict(collections.Counter)

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        self.ops[name] += 1
        fr = [f for f in traceback.extract_stack()[:-1]
              if "medical_image_segmentation_amd" in f.filename or "trace_aten" in f.filename]
        if fr:
            f = fr[-1]
            self.where[name][f"{os.path.basename(f.filename)}:{f.lineno}"] += 1
        return func(*args, **(kwargs or {}))


def main():
    from medical_image_segmentation_amd.model.unet import UNet
    from medical_image_segmentation_amd.utils.loss_fn import diceloss
    from medical_image_segmentation_amd.utils.metrics import Metrics
    from medical_image_segmentation_amd.utils.optim import clip_and_step
    dev = torch.device("cuda:0")
    which = sys.argv[1] if len(sys.argv) > 1 else "unet"
    if which == "unet":
        model, cin = UNet(1, 4, 0, base_channels=32, compute_dtype=torch.bfloat16), 1
    elif which == "a_unet":
        from medical_image_segmentation_amd.model.a_unet import A_UNet
        model, cin = A_UNet(3, 4, 0, base_channels=32, compute_dtype=torch.bfloat16), 3
    else:
        from medical_image_segmentation_amd.model.rdam_unet import RDAM_UNet
        model, cin = RDAM_UNet(3, 4, 0, base_channels=32, compute_dtype=torch.bfloat16), 3
    model = model.to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=8e-4, betas=(0.95, 0.999), weight_decay=1e-4)
    names = ["a", "b", "c"]
    loss_fn, metric = diceloss(names), Metrics(names)
    x = torch.randn(16, cin, 512, 512, device=dev)
    t = torch.randint(0, 4, (16, 512, 512), device=dev)
    params = list(model.parameters())

    def step():
        opt.zero_grad()
        logits = model(x)
        ld = loss_fn(logits, t)
        metric.confusion_counts(logits, t)
        ld["total_loss"].backward()
        clip_and_step(opt, params, max_norm=1.0)

    step()
    step()
    with Log() as log:
        step()
    for name, n in log.ops.most_common():
        print(f"{n:5d} {name:45s} {dict(log.where[name].most_common(4))}")


if __name__ == "__main__":
    main()
