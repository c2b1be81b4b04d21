#!/usr/bin/env python3
"""Do the kernels depend on LDS or vector registers they never wrote?  Runs train steps of the benchmark model twice from
the same state: plain, and with launches that fill every CU's LDS and every vector register with a NaN bit pattern before every
library call.  Every reduction is fixed-order, so the two runs must agree bit for bit; a difference names
the parameters whose gradients changed.   python tools/lds_pollution_check.py [--only substring]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from medical_image_segmentation_amd import _lib, functional as F_  # noqa: E402
from medical_image_segmentation_amd.model.unet import UNet  # noqa: E402
from medical_image_segmentation_amd.utils.loss_fn import diceloss  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--only", default="")
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--pattern", default="0xFFFFFFFF")
args = ap.parse_args()

pol = ctypes.CDLL(os.path.join(ROOT, "tools", "micro", "libpollute.so"))
pol.pollute_lds.argtypes = [ctypes.c_uint32, ctypes.c_void_p]
pol.pollute_regs.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
F_.WGRAD_SIDE_STREAM = False
real_call = F_.call
pattern = int(args.pattern, 0)
polluting = [False]
hit = set()


def call(name, *a):
    if polluting[0] and args.only in name:
        rc = pol.pollute_lds(pattern, F_._stream()) or pol.pollute_regs(F_._stream())
        if rc:
            raise RuntimeError(f"polluter launch failed: {rc}")
        hit.add(name)
    return real_call(name, *a)


F_.call = call


def run(pollute):
    torch.manual_seed(1234)
    model = UNet(1, 4, 0, base_channels=32, compute_dtype=torch.bfloat16).to(dev).train()
    g = torch.Generator(device="cpu").manual_seed(100)
    x = torch.randn(args.batch, 1, args.size, args.size, generator=g).to(dev)
    t = torch.randint(0, 4, (args.batch, args.size, args.size), generator=g).to(dev)
    loss_fn = diceloss(["a", "b", "c"])
    polluting[0] = pollute
    logits = model(x)
    ld = loss_fn(logits, t)
    ld["total_loss"].backward()
    torch.cuda.synchronize()
    polluting[0] = False
    return float(ld["total_loss"]), logits.detach().float().clone(), {n: p.grad.clone() for n, p in model.named_parameters()}


l0, lg0, g0 = run(False)
l1, lg1, g1 = run(True)
print("loss plain", l0, "polluted", l1, "logits equal", bool(torch.equal(lg0, lg1)),
      "logits finite", bool(torch.isfinite(lg1).all()))
bad = [n for n in g0 if not torch.equal(g0[n], g1[n])]
for n in bad:
    print("  differs:", n, "finite" if bool(torch.isfinite(g1[n]).all()) else "NON-FINITE",
          float((g0[n] - g1[n]).abs().max()))
print(len(bad), "of", len(g0), "gradients differ; polluted calls:", sorted(hit))
