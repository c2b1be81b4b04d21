"""CPU suite: the data-parallel gradient exchange (parallel.GradSync) with world size 2 on the
gloo backend, and the host-side metric post-math against the golden matrices."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import make_golden as mg
from oracle import unet_oracle as orc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_model():
    torch.manual_seed(7)
    return torch.nn.Sequential(torch.nn.Conv2d(2, 4, 3, padding=1), torch.nn.BatchNorm2d(4), torch.nn.ReLU(),
                               torch.nn.Conv2d(4, 3, 1))


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from medical_image_segmentation_amd.parallel import GradSync, all_reduce_dice_sums
    model = _make_model()
    if rank == 1:                      # rank 1 starts from different weights: broadcast must fix that
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    sync = GradSync(model, n_buckets=2)
    sync.broadcast_parameters()
    g = torch.Generator().manual_seed(100)
    x = torch.randn(4, 2, 8, 8, generator=g)
    xs = x[2 * rank:2 * rank + 2]
    for step in range(2):              # two steps: bucket state must reset
        model.zero_grad()
        sync.begin_step()
        model(xs).square().mean().backward()
        sync.finish_step()
    grads = [p.grad.clone() for p in model.parameters()]
    sums = torch.full((3, 4), float(rank + 1))
    all_reduce_dice_sums(sums)
    torch.save({"grads": grads, "params": [p.detach().clone() for p in model.parameters()], "sums": sums},
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_gradsync_world2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    # single-process emulation: two replicas on the two shards, gradients averaged
    g = torch.Generator().manual_seed(100)
    x = torch.randn(4, 2, 8, 8, generator=g)
    ref = None
    for r in range(2):
        m = _make_model()
        m(x[2 * r:2 * r + 2]).square().mean().backward()
        gr = [p.grad for p in m.parameters()]
        ref = gr if ref is None else [a + b for a, b in zip(ref, gr)]
    ref = [a / 2 for a in ref]
    for a, b, c in zip(r0["grads"], r1["grads"], ref):
        assert torch.equal(a, b)                       # both ranks hold the same averaged gradient
        assert torch.allclose(a, c, rtol=1e-5, atol=1e-7)
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)                       # broadcast made the replicas identical
    assert torch.equal(r0["sums"], torch.full((3, 4), 3.0)) and torch.equal(r1["sums"], r0["sums"])


def test_bucket_order_is_backward_order():
    from medical_image_segmentation_amd.parallel import GradSync
    model = _make_model()
    sync = GradSync(model, n_buckets=2)
    flat = [p for b in sync.buckets for p in b]
    assert flat == list(reversed(list(model.parameters())))
    assert sum(f.numel() for f in sync.flat) == sum(p.numel() for p in model.parameters())
    sync.remove()


@pytest.mark.parametrize("name", list(mg.metric_cases()))
def test_metrics_host_postmath_matches_reference(golden, name):
    """Metrics.from_counts (the host half of the product) on oracle counts == the reference's matrix."""
    from medical_image_segmentation_amd.utils.metrics import Metrics
    g = golden("metrics")
    c = mg.metric_cases()[name]
    x, t = mg.metric_inputs(c)
    cm = orc.confusion_counts(torch.from_numpy(x), torch.from_numpy(t))
    m = Metrics([f"c{i}" for i in range(1, c["n"] + 1)]).from_counts(torch.from_numpy(cm))
    assert np.array_equal(m, g[f"{name}.matrix"])


def test_finish_metrics_averages_batches(golden):
    from medical_image_segmentation_amd.utils.metrics import Metrics
    from medical_image_segmentation_amd.utils.train_and_eval import _finish_metrics
    c = mg.metric_cases()["met_n3"]
    x, t = mg.metric_inputs(c)
    cm = torch.from_numpy(orc.confusion_counts(torch.from_numpy(x), torch.from_numpy(t)))
    M = Metrics(["a", "b", "c"])
    ok = torch.zeros(1, dtype=torch.int32)
    total = _finish_metrics(M, [(cm, ok), (cm[:1], ok)], 4)
    assert np.allclose(total, M.from_counts(cm) + M.from_counts(cm[:1]))
    with pytest.raises(RuntimeError):
        _finish_metrics(M, [(cm, torch.ones(1, dtype=torch.int32))], 4)


def test_device_batches_order_and_format():
    """The batch iterator keeps the loader's order and the reference's batch format
    (data[0][0] = images, data[0][1] = masks, data[1] = names) on any device."""
    from medical_image_segmentation_amd.utils.train_and_eval import _device_batches
    batches = [((torch.full((2, 1, 4, 4), float(i)), torch.full((2, 4, 4), i)), [f"n{i}"]) for i in range(5)]
    got = list(_device_batches(batches, "cpu"))
    assert len(got) == 5
    for i, (im, mk) in enumerate(got):
        assert float(im.mean()) == i and int(mk.max()) == i
    assert list(_device_batches([], "cpu")) == []
