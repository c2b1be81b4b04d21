#!/usr/bin/env python3
"""Training step of A_UNet or RDAM_UNet (in=3, classes=4, base=32), p=0, 512x512 (the reference's
model/a_unet.py / model/rdam_unet.py at the U-Net benchmark's image size): forward + Dice loss + confusion counts +
backward + clip + AdamW, synthetic data, one JSON line.

    python tools/bench_aunet.py [--model a_unet|rdam_unet] [--batch 16] [--dtype bf16] [--steps 10]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="a_unet", choices=["a_unet", "rdam_unet"])
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    args = ap.parse_args()
    from medical_image_segmentation_amd.model.a_unet import A_UNet
    from medical_image_segmentation_amd.model.rdam_unet import RDAM_UNet
    from medical_image_segmentation_amd.utils.loss_fn import diceloss
    from medical_image_segmentation_amd.utils.metrics import Metrics
    from medical_image_segmentation_amd.utils.optim import clip_and_step
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    cls = A_UNet if args.model == "a_unet" else RDAM_UNet
    model = cls(3, 4, 0, base_channels=32, compute_dtype=cdt).to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=8e-4, betas=(0.95, 0.999), weight_decay=1e-4)
    names = ["a", "b", "c"]
    loss_fn, metric = diceloss(names), Metrics(names)
    x = torch.randn(args.batch, 3, args.size, args.size, device=dev)
    t = torch.randint(0, 4, (args.batch, args.size, args.size), device=dev)
    params = list(model.parameters())

    def step():
        opt.zero_grad()
        logits = model(x)
        ld = loss_fn(logits, t)
        metric.confusion_counts(logits, t)
        ld["total_loss"].backward()
        clip_and_step(opt, params, max_norm=1.0)
        return ld

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ld = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"workload": f"{cls.__name__}(in=3,classes=4,base=32) train step, batch {args.batch}, {args.size}x{args.size}",
                      "dtype": args.dtype, "ms_per_step": round(dt * 1e3, 2), "images_per_s": round(args.batch / dt, 1),
                      "params": sum(p.numel() for p in params), "final_loss": round(float(ld["total_loss"].detach()), 5),
                      "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}))


if __name__ == "__main__":
    main()
