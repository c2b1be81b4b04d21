#!/usr/bin/env python3
"""Host cost of the primitives the backward pass uses per layer (microseconds per call, GPU idle)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from medical_image_segmentation_amd import functional as F_  # noqa: E402
from medical_image_segmentation_amd._lib import query  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
main = torch.cuda.current_stream(dev)
side = torch.cuda.Stream(device=dev)
ev = torch.cuda.Event()
x = torch.empty(16, device=dev)


def bench(name, fn, n=20000):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{name:55s} {dt / n * 1e6:7.2f} us")


bench("torch.cuda.current_stream(dev)", lambda: torch.cuda.current_stream(dev))
bench("functional._stream()", F_._stream)
bench("side.wait_stream(main)", lambda: side.wait_stream(main))


def rec_wait():
    ev.record(main)
    side.wait_event(ev)


bench("ev.record(main); side.wait_event(ev)", rec_wait)
bench("torch.cuda.is_current_stream_capturing()", torch.cuda.is_current_stream_capturing)
bench("torch.empty(1<<20, uint8)", lambda: torch.empty(1 << 20, dtype=torch.uint8, device=dev))
bench("query(wgrad_workspace_bytes)", lambda: query("mis_conv3x3_wgrad_workspace_bytes", 16, 64, 64, 128, 64))
bench("torch._C._current_graph_task_id()", torch._C._current_graph_task_id)
bench("x.data_ptr()", x.data_ptr)
bench("x.untyped_storage()", x.untyped_storage)
with torch.cuda.stream(side):
    pass
def ctx():
    with torch.cuda.stream(side):
        pass
bench("with torch.cuda.stream(side): pass", ctx)
