"""The 3x3 max-pool kernels alone on a lazy bf16 input: python tools/bench_mp3.py [B C H W] [--strips] [--iters N]
Prints microseconds per launch (HIP events) and the rate over the algorithmic bytes (read + write + positions)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from medical_image_segmentation_amd import _lib
from medical_image_segmentation_amd import functional as F_

args = [a for a in sys.argv[1:] if not a.startswith("--")]
B, C, H, W = (int(a) for a in args[:4]) if len(args) >= 4 else (16, 32, 512, 512)
iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 10
dev = "cuda:0"
_lib.query("mis_set_maxpool3_path", 1 if "--strips" in sys.argv else 0)
z = torch.randn(B, C, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
ss = torch.stack([torch.rand(C) + 0.5, torch.randn(C) * 0.2]).to(dev)
gy = torch.randn(B, C, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
y, dx = torch.empty_like(z), torch.empty_like(z)
arg = torch.empty(z.numel(), dtype=torch.uint8, device=dev)
st = F_._stream()


def fwd():
    _lib.call("mis_maxpool3x3s1_fwd", z.data_ptr(), ss.data_ptr(), F_._dt(z), B, H, W, C, y.data_ptr(), arg.data_ptr(), st)


def bwd():
    _lib.call("mis_maxpool3x3s1_bwd", gy.data_ptr(), arg.data_ptr(), F_._dt(z), B, H, W, C, dx.data_ptr(), st)


nbytes = z.numel() * 5
for name, fn in (("fwd", fwd), ("bwd", bwd)):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{name} {B}x{C}x{H}x{W}: {us:.1f} us  {nbytes / us / 1e6:.2f} TB/s")
