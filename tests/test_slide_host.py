"""CPU suite: host logic of the sliding-window predictor (padding, window grid, Gaussian
weights, tile sharding and the world-size-2 class-map exchange on gloo)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import make_golden as mg
from oracle import datagen as dg
from oracle import unet_oracle as orc
from medical_image_segmentation_amd.utils import slide_predict as sp


def test_pad_and_windows_match_reference(golden):
    g = golden("slide")
    for name, c in mg.slide_cases().items():
        img = (dg.hash_u64((c["h"], c["w"], 3), c["seed"]) % np.uint64(256)).astype(np.uint8)
        pred = sp.SlidingWindowPredictor(None, "cpu", c["window"], c["stride"])
        padded, pads = pred._pad_image(img)
        assert tuple(pads) == tuple(g[f"{name}.pads"])
        ref_padded, _ = orc.slide_pad(img, c["window"], c["stride"])
        assert np.array_equal(padded, ref_padded)
        assert np.array_equal(np.array(pred._split_image(padded)), g[f"{name}.positions"])
    for (h, w, win, s) in [(4096, 4096, 1024, 512), (4096, 4096, 256, 128), (300, 500, 256, 128)]:
        pred = sp.SlidingWindowPredictor(None, "cpu", win, s)
        padded, pads = pred._pad_image(np.zeros((h, w, 1), dtype=np.uint8))
        assert tuple(pads) == tuple(g[f"slide_pos_{h}x{w}_{win}_{s}.pads"])
        assert np.array_equal(np.array(pred._split_image(padded)), g[f"slide_pos_{h}x{w}_{win}_{s}.positions"])


def test_gaussian_window():
    for ws in (4, 16, 31, 64):
        g = sp.gaussian_window(ws)
        assert g.dtype == np.float64 and g.shape == (ws, ws)
        # same formula evaluated by numpy: agreement to the last bits (libm vs numpy exp)
        np.testing.assert_allclose(g, orc.slide_gaussian(ws), rtol=4e-16, atol=0)
        assert g[ws // 2, ws // 2] == 1.0
    # the reference's own double loop (slide_predict.py:95-100), element by element
    import math
    ws = 8
    sigma = ws / 4
    for i in range(ws):
        for j in range(ws):
            d = ((i - ws // 2) ** 2 + (j - ws // 2) ** 2) ** 0.5
            assert sp.gaussian_window(ws)[i, j] == math.exp(-d ** 2 / (2 * sigma ** 2))


def test_shard_merge_roundtrip():
    for n, world in [(49, 8), (7, 2), (3, 4), (961, 8), (1, 1)]:
        maps = torch.arange(n, dtype=torch.uint8).reshape(n, 1, 1).expand(n, 2, 3).contiguous()
        longest = (n + world - 1) // world
        gathered = []
        seen = []
        for r in range(world):
            idx = sp.shard_tiles(n, r, world)
            seen += idx
            buf = torch.zeros((longest, 2, 3), dtype=torch.uint8)
            if idx:
                buf[:len(idx)] = maps[idx]
            gathered.append(buf)
        assert sorted(seen) == list(range(n))
        assert torch.equal(sp.merge_shards(gathered, n, world), maps)


def test_cpu_device_is_refused():
    pred = sp.SlidingWindowPredictor(lambda x: x, "cpu", 16, 8)
    try:
        pred.predict_array(np.zeros((32, 32, 3), dtype=np.uint8))
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("a CPU device must be refused")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 7                                  # odd: the shards differ in length
    g = torch.Generator().manual_seed(5)
    all_maps = torch.randint(0, 4, (n, 6, 6), generator=g, dtype=torch.uint8)
    mine = sp.shard_tiles(n, rank, world)
    merged = sp.gather_class_maps(all_maps[mine], n)
    torch.save(merged, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_gather_class_maps_world2_gloo(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    g = torch.Generator().manual_seed(5)
    all_maps = torch.randint(0, 4, (7, 6, 6), generator=g, dtype=torch.uint8)
    for r in range(2):
        assert torch.equal(torch.load(tmp_path / f"r{r}.pt"), all_maps)
