#!/usr/bin/env python3
"""Sliding-window prediction of a synthetic 4096x4096x3 uint8 slide (randint(0,256)) with
UNet(in=3, classes=4, base=32) in eval mode, at the two window/stride settings the reference
uses (predict.py:86-91: 1024/512 -> 64 tiles, 49 unique; class default 256/128 -> 1024 tiles,
961 unique).  Prints one JSON line per setting: whole call (host pad + upload + tiles + blend +
download) and the device part alone.

    python tools/bench_slide.py [--size 4096] [--dtype bf16] [--reps 3]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np   # noqa: E402
import torch         # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    from medical_image_segmentation_amd.model.unet import UNet
    from medical_image_segmentation_amd.utils.slide_predict import SlidingWindowPredictor
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model = UNet(3, 4, 0, base_channels=32, compute_dtype=cdt).to(dev).eval()
    img = np.random.default_rng(0).integers(0, 256, (args.size, args.size, 3), dtype=np.uint8)
    for ws, st in [(1024, 512), (256, 128)]:
        pred = SlidingWindowPredictor(model, dev, ws, st)
        padded, _ = pred._pad_image(img)
        n_tiles = len(pred._split_image(padded))
        n_unique = len(set(pred._split_image(padded)))
        pred.predict_array(img)                       # warm-up (kernels, Gaussian table)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            out = pred.predict_array(img)
        dt = (time.perf_counter() - t0) / args.reps
        print(json.dumps({"workload": f"slide {args.size}x{args.size}x3, window {ws}, stride {st}",
                          "tiles": n_tiles, "unique_tiles": n_unique, "dtype": args.dtype,
                          "seconds_per_slide": round(dt, 4), "tiles_per_s": round(n_tiles / dt, 1),
                          "megapixels_per_s": round(args.size * args.size / dt / 1e6, 1),
                          "classes_present": int(len(np.unique(out)))}))


if __name__ == "__main__":
    main()
