"""TEST INFRASTRUCTURE — deterministic input/weight generator shared by the
golden-vector script and the parity tests.

Values come from a counter-based integer hash (splitmix64 finaliser), so a
fixture only has to store the *outputs* of the reference: every test regenerates
the exact same inputs from (shape, seed) on any machine, independent of any
library's RNG stream.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return x ^ (x >> np.uint64(31))


def hash_u64(shape, seed):
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + (np.uint64(seed) << np.uint64(32))
        return _splitmix64(_splitmix64(idx)).reshape(shape)


def hash_uniform(shape, seed, lo=-1.0, hi=1.0):
    """float32 uniform in [lo, hi) on a 2^-24 grid (exactly representable)."""
    u = (hash_u64(shape, seed) >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(np.float32)


def hash_normalish(shape, seed, std=1.0):
    """Sum of 4 uniforms, rescaled to unit variance: bell-shaped, bounded."""
    acc = np.zeros(shape, dtype=np.float64)
    for k in range(4):
        acc += hash_uniform(shape, seed * 4 + k + 1000003).astype(np.float64)
    return (acc * (std * np.sqrt(3.0) / 2.0)).astype(np.float32)


def hash_int(shape, seed, n):
    """int64 in [0, n)."""
    return (hash_u64(shape, seed) >> np.uint64(33)).astype(np.int64) % int(n)


def blob_mask(b, h, w, n_classes, seed):
    """Segmentation-like int64 mask: blocky regions (8x8 cells) of random class,
    so every class forms contiguous areas the way real masks do."""
    ch, cw = max(h // 8, 1), max(w // 8, 1)
    cells = hash_int((b, (h + ch - 1) // ch, (w + cw - 1) // cw), seed, n_classes)
    m = np.repeat(np.repeat(cells, ch, axis=1), cw, axis=2)[:, :h, :w]
    return np.ascontiguousarray(m)
