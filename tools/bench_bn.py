#!/usr/bin/env python3
"""Per-layer timing of the BatchNorm+ReLU backward kernels (reduce, dz) and the pool / upsample
kernels at the benchmark shapes (B=16, 512^2 input, bf16)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_segmentation_amd import functional as F_
from medical_image_segmentation_amd._lib import call, query

LAYERS = [("enc1.1", 8, 512), ("enc1.2", 32, 512), ("enc2.1", 16, 256), ("enc2.2", 64, 256), ("enc3.1", 32, 128),
          ("enc3.2", 128, 128), ("enc4.1", 64, 64), ("enc4.2", 256, 64), ("cen.1", 512, 32), ("cen.2", 256, 32),
          ("dec1.1", 32, 64), ("dec1.2", 128, 64), ("dec2.1", 16, 128), ("dec2.2", 64, 128), ("dec3.1", 8, 256),
          ("dec3.2", 32, 256), ("dec4.1", 8, 512), ("dec4.2", 32, 512)]


def timeit(fn, iters=20):
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
    B, dt, dev = 16, torch.bfloat16, torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    tot = [0.0, 0.0]
    print(f"{'layer':8s} {'C':>4s} {'hw':>4s} {'MB':>6s} | {'reduce us':>9s} {'TB/s':>5s} | {'dz us':>8s} {'TB/s':>5s}")
    for name, C, hw in LAYERS:
        z = torch.randn(B, C, hw, hw, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        da = torch.randn_like(z)
        dz = torch.empty_like(z)
        n = B * hw * hw
        mr = torch.rand(2, C, device=dev) + 0.5
        g, b_ = torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev)
        sums = torch.empty(2, C, device=dev)
        dbias = torch.empty(C, device=dev)
        ws = F_.workspace(query("mis_bn_relu_bwd_workspace_bytes", n, C), dev)
        tr = timeit(lambda: call("mis_bn_relu_bwd_reduce", z.data_ptr(), da.data_ptr(), 1, n, C, mr.data_ptr(),
                                 g.data_ptr(), b_.data_ptr(), sums.data_ptr(), ws.data_ptr(), st))
        td = timeit(lambda: call("mis_bn_relu_bwd_dz", z.data_ptr(), da.data_ptr(), 1, n, C, mr.data_ptr(),
                                 g.data_ptr(), b_.data_ptr(), sums.data_ptr(), dz.data_ptr(), dbias.data_ptr(),
                                 ws.data_ptr(), st))
        mb = z.numel() * 2 / 1e6
        print(f"{name:8s} {C:4d} {hw:4d} {mb:6.1f} | {tr:9.1f} {2 * mb / tr:5.2f} | {td:8.1f} {3 * mb / td:5.2f}")
        tot[0] += tr
        tot[1] += td
    print("total us: reduce %.1f dz %.1f" % tuple(tot))
    # pool / upsample
    for name, C, hw in [("pool1", 32, 512), ("pool2", 64, 256), ("pool3", 128, 128), ("pool4", 256, 64)]:
        x = torch.randn(B, C, hw, hw, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        ss = torch.rand(2, C, device=dev)
        y = torch.empty(B, C, hw // 2, hw // 2, device=dev, dtype=dt).contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(x)
        add = torch.randn_like(x)
        tf = timeit(lambda: call("mis_maxpool2x2_fwd", x.data_ptr(), ss.data_ptr(), None, 1, B, hw, hw, C, y.data_ptr(), st))
        ss4 = torch.rand(4, C, device=dev)
        sums = torch.empty(2, C, device=dev)
        wsp = F_.workspace(query("mis_maxpool2x2_bwd_workspace_bytes", C), dev)
        tb = timeit(lambda: call("mis_maxpool2x2_bwd", x.data_ptr(), ss4.data_ptr(), None, y.data_ptr(), add.data_ptr(), 1, B,
                                 hw, hw, C, dx.data_ptr(), sums.data_ptr(), wsp.data_ptr(), st))
        mb = x.numel() * 2 / 1e6
        print(f"{name:8s} {C:4d} {hw:4d} {mb:6.1f} | fwd {tf:7.1f} us {1.25 * mb / tf:5.2f} TB/s | bwd(+add) {tb:7.1f} us {3.25 * mb / tb:5.2f} TB/s")
    for name, C, h in [("up1", 256, 32), ("up2", 128, 64), ("up3", 64, 128), ("up4", 32, 256)]:
        low = torch.randn(B, C, h, h, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        ss = torch.rand(2, C, device=dev)
        out = torch.empty(B, C, 2 * h, 2 * h, device=dev, dtype=dt).contiguous(memory_format=torch.channels_last)
        dlow = torch.empty_like(low)
        tf = timeit(lambda: call("mis_upsample2x_cat_fwd", low.data_ptr(), ss.data_ptr(), None, None, None, 1, B, h, h, C, 0,
                                 out.data_ptr(), st))
        tb = timeit(lambda: call("mis_upsample2x_cat_bwd", out.data_ptr(), None, 1, B, h, h, C, 0, dlow.data_ptr(), None, st))
        mb = out.numel() * 2 / 1e6
        print(f"{name:8s} {C:4d} {h:4d} {mb:6.1f} | fwd {tf:7.1f} us {1.25 * mb / tf:5.2f} TB/s | bwd {tb:7.1f} us {1.25 * mb / tb:5.2f} TB/s")


if __name__ == "__main__":
    main()
