#!/usr/bin/env python3
"""HBM roofs as torch sees them: fill (write-only), copy (read+write), sum (read-only)."""
import torch

def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

for mb in (64, 268, 537, 1074):
    n = mb * 1000 * 1000 // 2
    x = torch.empty(n, dtype=torch.bfloat16, device="cuda").normal_()
    y = torch.empty_like(x)
    tf = timeit(lambda: y.fill_(1.0))
    tc = timeit(lambda: y.copy_(x))
    ts = timeit(lambda: x.float().sum()) if mb <= 268 else float("nan")
    tr = timeit(lambda: torch.relu_(y))
    print(f"{mb:5d} MB: fill {tf:7.1f} us {mb/tf*1e-0/1e3*1e3/1e3:5.2f} TB/s | copy {tc:7.1f} us {2*mb/tc/1e3:5.2f} TB/s | relu_ (r+w) {tr:7.1f} us {2*mb/tr/1e3:5.2f} TB/s")
