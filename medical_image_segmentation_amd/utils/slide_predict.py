"""SlidingWindowPredictor — drop-in for the reference's utils/slide_predict.py:7-117.

Same constructor, `_pad_image`, `_split_image` and `predict(image_path, save_path)`; the work
between reading and writing the image runs on the GPU:

  * the padded uint8 image is uploaded once; tiles are cropped, scaled to [0,1] and laid out
    channels-last by one kernel per batch of tiles (`mis_slide_gather_tiles`), so the model sees
    a batch instead of one window at a time (slide_predict.py:81-92 runs them one by one);
  * duplicate windows (the clamped last row/column repeats the previous one whenever the
    padded size is a multiple of the stride, slide_predict.py:41-56) are predicted once;
  * `torch.argmax(pred, 1)` becomes a uint8 class map on the device (`mis_argmax_classes`);
  * the Gaussian-weighted blend (slide_predict.py:94-112) is a per-pixel gather that visits
    the covering tiles in the reference's order (duplicates included), reproducing numpy's
    float32 `+=` of float64 products bit for bit (`mis_slide_blend`);
  * with a process group, the unique tiles are dealt round-robin to the ranks and the uint8
    class maps are all-gathered; every rank then holds the same blended map.

`predict_array(image)` is the same computation on an in-memory [H, W, 3] uint8 array.
"""
import math

import numpy as np
import torch

from .. import functional as F_
from .._lib import call

_GAUSS = {}


def gaussian_window(window_size):
    """The per-tile weight of slide_predict.py:95-100 as a float64 [window, window] array,
    evaluated with the same Python float operations — `d = (di**2 + dj**2) ** 0.5`,
    `exp(-d**2 / (2*sigma**2))`, sigma = window/4 — once per distinct squared distance
    instead of once per pixel per tile."""
    g = _GAUSS.get(window_size)
    if g is None:
        off = np.arange(window_size, dtype=np.int64) - window_size // 2
        r2 = off[:, None] ** 2 + off[None, :] ** 2
        uniq, inv = np.unique(r2, return_inverse=True)
        sigma = window_size / 4
        vals = np.empty(len(uniq), dtype=np.float64)
        for k, u in enumerate(uniq.tolist()):
            d = u ** 0.5
            vals[k] = math.exp(-d ** 2 / (2 * sigma ** 2))
        g = vals[inv.reshape(-1)].reshape(window_size, window_size)
        _GAUSS[window_size] = g
    return g


def shard_tiles(n_unique, rank, world):
    """Indices of the unique tiles rank `rank` predicts: round-robin, so every rank gets
    ceil or floor of n/world tiles whatever the grid shape."""
    return list(range(rank, n_unique, world))


def merge_shards(gathered, n_unique, world):
    """Inverse of `shard_tiles`: `gathered[r]` holds rank r's class maps (padded to the longest
    shard) in its own tile order; returns them stacked in unique-tile order."""
    first = gathered[0]
    out = first.new_empty((n_unique,) + tuple(first.shape[1:]))
    for r in range(world):
        idx = shard_tiles(n_unique, r, world)
        if idx:
            out[torch.as_tensor(idx, device=out.device)] = gathered[r][:len(idx)]
    return out


def gather_class_maps(local, n_unique, group=None):
    """All-gathers the ranks' class maps (`local` = this rank's shard per `shard_tiles`, uint8
    [n_local, window, window]) and returns all `n_unique` maps in unique-tile order on every
    rank.  The one exchange step of the multi-GPU path: <= 1 byte per tile pixel."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    longest = (n_unique + world - 1) // world
    send = local.new_zeros((longest,) + tuple(local.shape[1:]))
    send[:local.shape[0]] = local
    gathered = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(gathered, send, group=group)
    return merge_shards(gathered, n_unique, world)


class SlidingWindowPredictor:
    def __init__(self, model, device, window_size=256, stride=128):
        self.model = model
        self.device = device
        self.window_size = window_size
        self.stride = stride
        self.pad_value = 0
        self.batch_tiles = None       # tiles per model call; None = as many as ~4 Mpixel

    # -- host geometry (slide_predict.py:15-58) -------------------------------------------
    def _pad_image(self, image):
        """Reflect-pads bottom/right by (window - size % stride) % window."""
        h, w = image.shape[:2]
        ws, st = self.window_size, self.stride
        pad_h, pad_w = (ws - h % st) % ws, (ws - w % st) % ws
        if pad_h or pad_w:
            image = np.pad(image, ((0, pad_h), (0, pad_w), (0, 0)), "reflect")
        return image, (pad_h, pad_w)

    def _starts(self, size):
        """Window starts along one axis: k*stride for k = 0..num, clamped so the window ends
        inside the image (the clamped last entry may repeat the one before it)."""
        ws, st = self.window_size, self.stride
        num = (size - ws) // st + 1
        return [min(k * st, size - ws) for k in range(num + 1)]

    def _split_image(self, image):
        h, w = image.shape[:2]
        ws = self.window_size
        return [(y, y + ws, x, x + ws) for y in self._starts(h) for x in self._starts(w)]

    # -- device path -----------------------------------------------------------------------
    def _tiles_per_call(self):
        if self.batch_tiles:
            return int(self.batch_tiles)
        return max(1, (4 << 20) // (self.window_size * self.window_size))

    def _class_maps(self, image_dev, corners):
        """uint8 [n, window, window] class maps of the windows at `corners` ([(y, x), ...])."""
        ws = self.window_size
        H, W, Cc = image_dev.shape
        n = len(corners)
        maps = torch.empty((n, ws, ws), dtype=torch.uint8, device=image_dev.device)
        if n == 0:
            return maps
        yx = torch.tensor(corners, dtype=torch.int32).to(image_dev.device)
        dtype = getattr(self.model, "compute_dtype", None) or torch.float32
        if dtype not in F_._DT:
            dtype = torch.float32
        step = self._tiles_per_call()
        st = F_._stream()
        with torch.no_grad():
            for b0 in range(0, n, step):
                nb = min(step, n - b0)
                x = torch.empty((nb, Cc, ws, ws), dtype=dtype, device=image_dev.device,
                                memory_format=torch.channels_last)
                call("mis_slide_gather_tiles", image_dev.data_ptr(), H, W, Cc, yx[b0:].data_ptr(), nb, ws,
                     F_._DT[dtype], x.data_ptr(), st)
                pred = self.model(x)
                if pred.dim() != 4 or pred.shape[0] != nb or tuple(pred.shape[2:]) != (ws, ws):
                    raise ValueError(f"model returned {tuple(pred.shape)} for a {tuple(x.shape)} batch")
                if pred.dtype not in F_._DT:
                    pred = pred.float()
                sb, sc, sh, sw = pred.stride()
                if sh != ws * sw:                       # pixels must be addressable by one stride
                    pred = pred.contiguous()
                    sb, sc, sh, sw = pred.stride()
                call("mis_argmax_classes", pred.data_ptr(), F_._DT[pred.dtype], sb, sc, sw, nb, pred.shape[1],
                     ws * ws, maps[b0:].data_ptr(), F_._stream())
        return maps

    def predict_array(self, image, group=None):
        """[H, W, C] uint8 array -> [H, W] uint8 class map (slide_predict.py:66-112).
        `group`: a torch.distributed process group (or True for the default group) to split the
        tiles over its ranks; every rank must call with the same image."""
        image = np.ascontiguousarray(image)
        if image.ndim != 3 or image.dtype != np.uint8:
            raise ValueError(f"expected a [H, W, C] uint8 array, got {image.shape} {image.dtype}")
        device = torch.device(self.device)
        if device.type != "cuda":
            raise RuntimeError(f"SlidingWindowPredictor needs a CUDA/HIP device, got {device} "
                               "(there is no CPU fallback)")
        h, w = image.shape[:2]
        padded, _ = self._pad_image(image)
        ph, pw = padded.shape[:2]
        ws = self.window_size
        if ph < ws or pw < ws:
            raise ValueError(f"padded image {ph}x{pw} is smaller than the window {ws}")
        ys, xs = self._starts(ph), self._starts(pw)
        # unique windows, in first-appearance order, and each tile's index into them
        uniq, index = {}, []
        for y in ys:
            for x in xs:
                index.append(uniq.setdefault((y, x), len(uniq)))
        corners = list(uniq.keys())

        with torch.cuda.device(device):
            image_dev = torch.from_numpy(padded).to(device)
            if group is None:
                maps = self._class_maps(image_dev, corners)
            else:
                import torch.distributed as dist
                pg = None if group is True else group
                mine = shard_tiles(len(corners), dist.get_rank(pg), dist.get_world_size(pg))
                local = self._class_maps(image_dev, [corners[k] for k in mine])
                maps = gather_class_maps(local, len(corners), pg)
            gauss = torch.from_numpy(gaussian_window(ws)).to(device)
            idx = torch.tensor(index, dtype=torch.int32).to(device)
            ys_d = torch.tensor(ys, dtype=torch.int32).to(device)
            xs_d = torch.tensor(xs, dtype=torch.int32).to(device)
            out = torch.empty((h, w), dtype=torch.uint8, device=device)
            call("mis_slide_blend", maps.data_ptr(), idx.data_ptr(), ys_d.data_ptr(), xs_d.data_ptr(), len(ys),
                 len(xs), ws, gauss.data_ptr(), h, w, out.data_ptr(), F_._stream())
            return out.cpu().numpy()

    def predict(self, image_path, save_path):
        from PIL import Image
        original = np.array(Image.open(image_path).convert("RGB"))
        result = self.predict_array(original)
        Image.fromarray(result).save(save_path)
        print(f"Prediction saved to {save_path}")
