#!/usr/bin/env python3
"""GPU time of the first-layer forward (Cin=1 -> 8 @ 512^2, B=16, bf16) over 50 back-to-back launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_segmentation_amd import functional as F_
dev = torch.device("cuda:0")
x = torch.randn(16, 1, 512, 512, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = torch.randn(8, 1, 3, 3, device=dev) * 0.1
b = torch.zeros(8, device=dev)
wf = F_.pack_weights(w, torch.bfloat16)
dz = torch.randn(16, 8, 512, 512, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
for name, fn in (("fwd", lambda: F_.conv3x3_raw(x, wf, b, 8, "partial", None)), ("wgrad", lambda: F_.conv3x3_wgrad_raw(x, dz, None))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(name, round(e0.elapsed_time(e1) * 1e3 / 50, 1), "us per call (host-bound below ~40)")
