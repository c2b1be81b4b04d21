#!/usr/bin/env python3
"""Per-layer timing of the conv kernels at the benchmark shapes (B=16, 512^2 input, bf16):
forward conv (+stats), data gradient and weight gradient of every DoubleConv layer.
Prints us, algorithmic TB/s and TFLOP/s, and the roofline time min-bound max(bytes/6.3TB/s, flops/2.5PF)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_segmentation_amd import functional as F_

LAYERS = [("enc1.1", 1, 8, 512), ("enc1.2", 8, 32, 512), ("enc2.1", 32, 16, 256), ("enc2.2", 16, 64, 256),
          ("enc3.1", 64, 32, 128), ("enc3.2", 32, 128, 128), ("enc4.1", 128, 64, 64), ("enc4.2", 64, 256, 64),
          ("cen.1", 256, 512, 32), ("cen.2", 512, 256, 32), ("dec1.1", 512, 32, 64), ("dec1.2", 32, 128, 64),
          ("dec2.1", 256, 16, 128), ("dec2.2", 16, 64, 128), ("dec3.1", 128, 8, 256), ("dec3.2", 8, 32, 256),
          ("dec4.1", 64, 8, 512), ("dec4.2", 8, 32, 512)]


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--no-cat", action="store_true", help="decoder entry convs on a materialised concatenation")
    ap.add_argument("--lazy", action="store_true", help="input is a lazy activation (affine+ReLU applied on load)")
    args = ap.parse_args()
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    es = 2 if dt == torch.bfloat16 else 4
    dev = torch.device("cuda:0")
    B = args.batch
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    ideal = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    print(f"{'layer':8s} {'shape':>18s} | {'fwd us':>8s} {'TB/s':>5s} {'TF':>6s} {'ideal':>6s} | {'dgrad':>8s} {'TB/s':>5s} {'TF':>6s} | {'wgrad':>8s} {'TB/s':>5s} {'TF':>6s}")
    for name, ci, co, hw in LAYERS:
        if args.only and args.only not in name:
            continue
        x = torch.randn(B, ci, hw, hw, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        dz = torch.randn(B, co, hw, hw, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
        b = torch.zeros(co, device=dev)
        wf = F_.pack_weights(w, dt)
        wd = F_.pack_weights(w, dt, transposed=True)
        n = B * hw * hw
        by = n * (ci + co) * es
        fl = 18 * n * ci * co
        t_ideal = max(by / 6.3e12, fl / 2.5e15) * 1e6
        ss = torch.rand(2, ci, device=dev) if args.lazy else None
        if name.startswith("dec") and name.endswith(".1") and not args.no_cat:
            # decoder entry conv: [up | skip] as two tensors (skip lazy), gradient split in two
            xa = x[:, :ci // 2].contiguous(memory_format=torch.channels_last)
            xb = x[:, ci // 2:].contiguous(memory_format=torch.channels_last)
            ssb = torch.rand(2, ci // 2, device=dev) if args.lazy else None
            assert F_.cat_supported(xa, xb, co)
            tf = timeit(lambda: F_.conv3x3_raw(xa, wf, b, co, "partial", None, xb, ssb), args.iters)
            td = timeit(lambda: F_.conv3x3_split_raw(dz, wd, ci // 2, ci // 2), args.iters)
            tw = timeit(lambda: F_.conv3x3_wgrad_raw(xa, dz, None, xb, ssb), args.iters)
        else:
            tf = timeit(lambda: F_.conv3x3_raw(x, wf, b, co, "partial", ss), args.iters)
            td = timeit(lambda: F_.conv3x3_raw(dz, wd, None, ci, False), args.iters) if ci >= 8 else float("nan")
            tw = timeit(lambda: F_.conv3x3_wgrad_raw(x, dz, ss), args.iters)
        f = lambda t: (by / t / 1e6, fl / t / 1e6)
        print(f"{name:8s} {ci:4d}->{co:4d} @{hw:4d} | {tf:8.1f} {f(tf)[0]:5.2f} {f(tf)[1]:6.1f} {t_ideal:6.1f} | "
              f"{td:8.1f} {f(td)[0]:5.2f} {f(td)[1]:6.1f} | {tw:8.1f} {f(tw)[0]:5.2f} {f(tw)[1]:6.1f}")
        tot["fwd"] += tf
        tot["dgrad"] += 0 if td != td else td
        tot["wgrad"] += tw
        for k in ideal:
            ideal[k] += t_ideal
    print("total us:", {k: round(v, 1) for k, v in tot.items()}, "ideal per pass:", round(ideal["fwd"], 1))


if __name__ == "__main__":
    main()
