#!/usr/bin/env python3
"""Bilinear x2 forward under contention, with the evidence kept: repeats one launch on a fixed input, compares every
result with the first bit for bit and, for the first launches that differ, prints where (lane of the workgroup that
wrote the element), the stored and the expected bit patterns, the four source values and the tap weights.
Start two of these at once (tools/up_diag.sh).   python tools/up_diag.py [iters] [bf16|f32] [C] [h]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from medical_image_segmentation_amd import functional as F_  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dtype = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.bfloat16
C = int(sys.argv[3]) if len(sys.argv) > 3 else 32
h = int(sys.argv[4]) if len(sys.argv) > 4 else 256
B = 16
dev = torch.device("cuda:0")
torch.manual_seed(5)
low = torch.randn(B, C, h, h, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
ss = torch.stack([torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1,
                  torch.zeros(C, device=dev), torch.ones(C, device=dev)]).contiguous()
ref = F_.UpsampleFn.apply(low, ss, None).clone()
torch.cuda.synchronize()
bad_launches, shown = 0, 0
itype = torch.int16 if dtype == torch.bfloat16 else torch.int32
for it in range(iters):
    out = F_.UpsampleFn.apply(low, ss, None)
    d = out != ref
    if bool(d.any()):
        bad_launches += 1
        if shown < 4:
            shown += 1
            idx = torch.nonzero(d.permute(0, 2, 3, 1))[:48].cpu()        # (b, y, x, c)
            print(f"pid {os.getpid()} launch {it}: {int(d.sum())} differing elements")
            sy = (h - 1) / (2 * h - 1)
            for b, y, x, c in idx.tolist():
                got = out[b, c, y, x].view(itype).item() & (0xFFFF if dtype == torch.bfloat16 else 0xFFFFFFFF)
                exp = ref[b, c, y, x].view(itype).item() & (0xFFFF if dtype == torch.bfloat16 else 0xFFFFFFFF)
                y0 = min(int(np.float32(sy) * np.float32(y)), h - 1)
                x0 = min(int(np.float32(sy) * np.float32(x)), h - 1)
                y1, x1 = min(y0 + 1, h - 1), min(x0 + 1, h - 1)
                src = [float(low[b, c, yy, xx]) for yy, xx in ((y0, x0), (y0, x1), (y1, x0), (y1, x1))]
                print(f"   b{b} y{y} x{x} c{c}: got {got:#x} ({float(out[b, c, y, x]):.7g}) expected {exp:#x} "
                      f"({float(ref[b, c, y, x]):.7g}); tile ({y // 16},{x // 16}) in-tile ({y % 16},{x % 16}); "
                      f"sources z {src} scale {float(ss[0, c]):.6g} shift {float(ss[1, c]):.6g}")
torch.cuda.synchronize()
print(f"pid {os.getpid()} {dtype} C={C} h={h}: {bad_launches} of {iters} launches differ from the first")
