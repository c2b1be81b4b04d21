#!/usr/bin/env python3
"""Host time inside the backward pass by call site (tiny input, GPU work negligible): wraps the library calls and the
stream waits of medical_image_segmentation_amd.functional with timers.  Microseconds per step."""
import collections
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from medical_image_segmentation_amd import functional as F_  # noqa: E402
from medical_image_segmentation_amd.model.unet import UNet  # noqa: E402
from medical_image_segmentation_amd.utils.loss_fn import diceloss  # noqa: E402
from medical_image_segmentation_amd.utils.optim import clip_and_step  # noqa: E402
from medical_image_segmentation_amd import parallel  # noqa: E402
import torch.distributed as dist  # noqa: E402

dev = torch.device("cuda:0")
model = UNet(1, 4, 0, base_channels=32, compute_dtype=torch.bfloat16).to(dev).train()
opt = torch.optim.AdamW(model.parameters(), lr=8e-4)
loss_fn = diceloss(["a", "b", "c"])
x = torch.randn(1, 1, 64, 64, device=dev)
t = torch.randint(0, 4, (1, 64, 64), device=dev)
params = list(model.parameters())
sync = None
if "--dist" in sys.argv:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29545")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    sync = parallel.GradSync(model, always_reduce=True)
acc = collections.defaultdict(lambda: [0.0, 0])
on = [False]

orig_call = F_.call


def timed_call(name, *a):
    t0 = time.perf_counter()
    r = orig_call(name, *a)
    if on[0]:
        e = acc["call " + name]
        e[0] += time.perf_counter() - t0
        e[1] += 1
    return r


F_.call = timed_call
for fname in ("_wgrad_deferred", "flush_wgrad_reduces", "conv3x3_wgrad_raw"):
    orig = getattr(F_, fname)

    def wrap(*a, _o=orig, _n=fname, **k):
        t0 = time.perf_counter()
        r = _o(*a, **k)
        if on[0]:
            e = acc[_n + " (whole)"]
            e[0] += time.perf_counter() - t0
            e[1] += 1
        return r
    setattr(F_, fname, wrap)
orig_ws = torch.cuda.Stream.wait_stream


def ws(self, other):
    t0 = time.perf_counter()
    orig_ws(self, other)
    if on[0]:
        e = acc["Stream.wait_stream"]
        e[0] += time.perf_counter() - t0
        e[1] += 1


torch.cuda.Stream.wait_stream = ws
for fname in ("_on_grad", "_launch", "finish_step", "begin_step"):
    orig = getattr(parallel.GradSync, fname)

    def wrapm(self, *a, _o=orig, _n=fname, **k):
        t0 = time.perf_counter()
        r = _o(self, *a, **k)
        if on[0]:
            e = acc["GradSync." + _n + " (whole)"]
            e[0] += time.perf_counter() - t0
            e[1] += 1
        return r
    setattr(parallel.GradSync, fname, wrapm)
if sync is not None:      # the hooks were registered with the unwrapped bound method
    sync.remove()
    sync = parallel.GradSync(model, always_reduce=True)
orig_ar = dist.all_reduce


def ar(*a, **k):
    t0 = time.perf_counter()
    r = orig_ar(*a, **k)
    if on[0]:
        e = acc["dist.all_reduce"]
        e[0] += time.perf_counter() - t0
        e[1] += 1
    return r


dist.all_reduce = ar
parallel.dist.all_reduce = ar
for cls in (F_.ConvBnReluFn,):
    ob = cls.backward

    def bw(ctx, *g, _o=ob):
        t0 = time.perf_counter()
        r = _o(ctx, *g)
        if on[0]:
            e = acc["ConvBnReluFn.backward (whole)"]
            e[0] += time.perf_counter() - t0
            e[1] += 1
        return r
    cls.backward = staticmethod(bw)


def step():
    opt.zero_grad()
    ld = loss_fn(model(x), t)
    t0 = time.perf_counter()
    if sync is not None:
        sync.begin_step()
    ld["total_loss"].backward()
    if on[0]:
        e = acc["backward() (whole)"]
        e[0] += time.perf_counter() - t0
        e[1] += 1
    if sync is not None:
        sync.finish_step()
    clip_and_step(opt, params, max_norm=1.0)


for _ in range(10):
    step()
torch.cuda.synchronize()
on[0] = True
n = 100
for _ in range(n):
    step()
torch.cuda.synchronize()
print(f"side stream {F_.WGRAD_SIDE_STREAM}")
for k, (s, c) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:50s} {s / n * 1e6:9.1f} us per step  {c / n:6.1f} calls  {s / max(c, 1) * 1e6:7.2f} us each")
if getattr(F_, "_T", None):
    d = collections.defaultdict(float)
    T = F_._T
    for (a, ta), (b, tb) in zip(T, T[1:]):
        if b != "L0":
            d[a + "->" + b] += tb - ta
    for k, v in sorted(d.items(), key=lambda kv: -kv[1])[:8]:
        print(f"  lines {k:10s} {v / (n + 10) * 1e6:8.1f} us per step")
