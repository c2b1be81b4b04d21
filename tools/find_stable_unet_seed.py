#!/usr/bin/env python3
"""Picks seeds for whole-model gradient fixtures whose gradients are insensitive to last-bit forward differences:
runs the CPU oracle's forward + Dice loss + backward in float32 and in float64 on the same hash-generated weights and
batch and ranks the seeds by the worst per-tensor gradient deviation between the two (a ReLU whose pre-activation is
within round-off of zero at a pixel that carries a large Dice gradient moves whole tensors by per cents in the
2-channel BatchNorm layers of the base-8 fixture networks).   python tools/find_stable_unet_seed.py [first] [count]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from oracle import datagen as dg  # noqa: E402
from oracle import unet_oracle as orc  # noqa: E402
import make_golden as mg  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 30
count = int(sys.argv[2]) if len(sys.argv) > 2 else 24
B, cin, ncls, hw = 2, 3, 4, 64
torch.set_num_threads(8)
rows = []
for seed in range(first, first + count):
    x = torch.from_numpy(dg.hash_normalish((B, cin, hw, hw), seed + 100))
    t = torch.from_numpy(dg.blob_mask(B, hw, hw, ncls, seed + 300))
    names = [f"c{i}" for i in range(1, ncls)]
    grads = {}
    for dt in (torch.float32, torch.float64):
        st = orc.make_state(cin, ncls, base=8, seed=seed, dtype=dt)
        keys = orc.param_keys(st)
        for k in keys:
            st[k].requires_grad_(True)
        lo = orc.unet_forward(x.to(dt), st, training=True)
        orc.dice_loss(lo, t, names)["total_loss"].backward()
        grads[dt] = {k: st[k].grad.double() for k in keys}
    worst_n, worst_s = 0.0, 0.0
    for k in grads[torch.float32]:
        if k.endswith(".0.bias") and k != "out_conv.bias":
            continue
        a, b = grads[torch.float32][k], grads[torch.float64][k]
        worst_n = max(worst_n, abs(float(a.norm()) - float(b.norm())) / (float(b.norm()) + 1e-12))
        idx = mg.sample_idx(a.numel())
        worst_s = max(worst_s, float((a.reshape(-1)[idx] - b.reshape(-1)[idx]).abs().max() / (b.abs().max() + 1e-30)))
    rows.append((max(worst_n, worst_s), worst_n, worst_s, seed))
    print(f"seed {seed}: norm deviation {worst_n:.2e}, sample deviation {worst_s:.2e}", flush=True)
rows.sort()
print("most stable:", [(r[3], f"{r[0]:.1e}") for r in rows[:6]])
