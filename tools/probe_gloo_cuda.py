"""Probe: does this torch build's gloo backend all-reduce HIP tensors (two ranks on one GPU)?"""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def run(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.full((4,), float(rank + 1), device="cuda:0")
    dist.all_reduce(t)
    h = dist.all_reduce(t, async_op=True)
    h.wait()
    torch.cuda.synchronize()
    g = torch.empty(2 * 3, dtype=torch.int64, device="cuda:0")
    dist.all_gather_into_tensor(g, torch.full((3,), rank, dtype=torch.int64, device="cuda:0"))
    print(rank, t.tolist(), g.tolist(), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(run, args=(2, 29611), nprocs=2)
    print("gloo on HIP tensors: ok")
