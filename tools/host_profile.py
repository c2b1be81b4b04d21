#!/usr/bin/env python3
"""cProfile of the host side of a training step on a tiny input (GPU work negligible): where the
Python / ctypes time of the ~230 launches goes.  usage: python tools/host_profile.py"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    from medical_image_segmentation_amd.model.unet import UNet
    from medical_image_segmentation_amd.utils.loss_fn import diceloss
    from medical_image_segmentation_amd.utils.metrics import Metrics
    from medical_image_segmentation_amd.utils.optim import clip_and_step
    dev = torch.device("cuda:0")
    model = UNet(1, 4, 0, base_channels=32, compute_dtype=torch.bfloat16).to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=8e-4)
    loss_fn, metric = diceloss(["a", "b", "c"]), Metrics(["a", "b", "c"])
    x = torch.randn(1, 1, 64, 64, device=dev)
    t = torch.randint(0, 4, (1, 64, 64), device=dev)
    params = list(model.parameters())

    def step():
        opt.zero_grad()
        logits = model(x)
        ld = loss_fn(logits, t)
        metric.confusion_counts(logits, t)
        ld["total_loss"].backward()
        clip_and_step(opt, params, max_norm=1.0)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(50):
        step()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
