#!/usr/bin/env python3
"""Which gradients change when the weight-gradient slab launches run on the second stream?  Every step
computes the gradients twice from the same parameters and batch (second stream off / on) — all
reductions are fixed-order, so they must agree bit for bit — and reports the first differences."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from medical_image_segmentation_amd import functional as F_  # noqa: E402
from medical_image_segmentation_amd.model.unet import UNet  # noqa: E402
from medical_image_segmentation_amd.utils.loss_fn import diceloss  # noqa: E402
from medical_image_segmentation_amd.utils.optim import clip_and_step  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = UNet(1, 4, 0, base_channels=32, compute_dtype=torch.bfloat16).to(dev).train()
opt = torch.optim.AdamW(model.parameters(), lr=8e-4, betas=(0.95, 0.999), weight_decay=1e-4)
g = torch.Generator(device="cpu").manual_seed(100)
x = torch.randn(16, 1, 512, 512, generator=g).to(dev)
t = torch.randint(0, 4, (16, 512, 512), generator=g).to(dev)
loss_fn = diceloss(["a", "b", "c"])
params = list(model.parameters())
names = [n for n, _ in model.named_parameters()]


def grads(side):
    F_.WGRAD_SIDE_STREAM = side
    opt.zero_grad()
    # batch statistics must not move between the two passes
    state = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}
    ld = loss_fn(model(x), t)
    ld["total_loss"].backward()
    model.load_state_dict(state, strict=False)
    return [p.grad for p in params]


bad_steps = 0
for step in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    a = [v.clone() for v in grads(False)]
    b = grads(True)
    torch.cuda.synchronize()
    diff = [(n, u, v) for n, u, v in zip(names, a, b) if not torch.equal(u, v)]
    if diff:
        bad_steps += 1
        print(f"step {step}: {len(diff)} gradients differ")
        for n, u, v in diff[:6]:
            d = (u - v).abs()
            idx = torch.nonzero(d.flatten() > 0).flatten()
            print(f"   {n} shape {tuple(u.shape)}: {idx.numel()} of {u.numel()} elements, finite={bool(torch.isfinite(v).all())}, "
                  f"max|d|={float(d.max()):.3e}, first flat indices {idx[:8].tolist()}, span {int(idx.min())}..{int(idx.max())}")
        if bad_steps >= 3:
            break
    F_.WGRAD_SIDE_STREAM = False
    for p, u in zip(params, a):
        p.grad = u
    clip_and_step(opt, params, max_norm=1.0)
print("steps with differences:", bad_steps)
