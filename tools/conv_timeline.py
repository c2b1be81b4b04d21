#!/usr/bin/env python3
"""Phase timeline of the conv forward kernel from in-kernel cycle stamps (library built with
-DMIS_TIMELINE=1 and selected through MISAMD_LIBRARY).  usage: conv_timeline.py <layer> [--dgrad] [--lazy]"""
import argparse
import ctypes
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from medical_image_segmentation_amd import functional as F_, _lib
from bench_conv import LAYERS

NAMES = {1: "entry", 2: "prologue done", 10: "stage begin", 11: "barrier A passed", 16: "loads arrived", 12: "halo(+w) stored",
         13: "barrier B passed", 14: "next loads issued", 15: "MFMAs done", 20: "epilogue done", 30: "kernel end"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("layer")
    ap.add_argument("--dgrad", action="store_true")
    ap.add_argument("--wgrad", action="store_true")
    ap.add_argument("--lazy", action="store_true")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--dump", type=int, default=0)
    a = ap.parse_args()
    name, ci, co, hw = next(l for l in LAYERS if l[0] == a.layer)
    if a.dgrad:
        ci, co = co, ci
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    x = torch.randn(a.batch, ci, hw, hw, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    wf = F_.pack_weights(w, dt)
    b = torch.zeros(co, device=dev)
    ss = torch.rand(2, ci, device=dev) if a.lazy else None
    run = lambda: F_.conv3x3_raw(x, wf, b, co, False if a.dgrad else "partial", ss)
    if a.wgrad:
        dz = torch.randn(a.batch, co, hw, hw, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        run = lambda: F_.conv3x3_wgrad_raw(x, dz, ss)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record()
    torch.cuda.synchronize()
    buf = np.zeros((4, 1024), dtype=np.uint64)
    lib = ctypes.CDLL(os.environ["MISAMD_LIBRARY"])
    rc = lib.mis_debug_timeline(buf.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, rc
    print(f"{name} {ci}->{co}@{hw} {'wgrad' if a.wgrad else 'dgrad' if a.dgrad else 'fwd'} lazy={a.lazy}: {e0.elapsed_time(e1) * 1e3:.1f} us (event)")
    for slot in range(3):
        n = int(buf[slot, 1023])
        if n == 0:
            continue
        ids = (buf[slot, :n] >> np.uint64(56)).astype(int)
        t = (buf[slot, :n] & np.uint64((1 << 56) - 1)).astype(np.int64)
        total = t[-1] - t[0]
        acc = defaultdict(lambda: [0, 0])
        for i in range(1, n):
            k = (ids[i - 1], ids[i])
            acc[k][0] += int(t[i] - t[i - 1])
            acc[k][1] += 1
        print(f" workgroup slot {slot}: {n} stamps, {total} ticks total")
        for (p, q), (s, c) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
            print(f"   {NAMES.get(p, p)!s:36s} -> {NAMES.get(q, q)!s:36s} {s:9d} ticks ({100.0 * s / total:5.1f} %)  x{c:4d}  avg {s / c:8.1f}")
        if a.dump and slot == 0:
            for i in range(min(n, a.dump)):
                print("     ", ids[i], int(t[i] - t[0]))


if __name__ == "__main__":
    main()
