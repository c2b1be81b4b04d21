"""Tensor-level wrappers and autograd Functions over the C-ABI (include/misamd.h).

PyTorch is used for device memory, streams and the autograd graph only: every op
below launches hand-written gfx950 kernels through `_lib.call`, on torch's current
stream, on tensors owned by torch's caching allocator.

Layout contract: an activation is a torch tensor of logical shape [B, C, H, W]
stored channels-last (`memory_format=torch.channels_last`), dtype float32 or
bfloat16.  Parameters are float32 in the reference's own layouts.
"""
import ctypes
import os
import weakref

import torch

from . import _lib
from ._lib import call, query

_DT = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16}
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _stream():
    """Raw handle of torch's current stream on the current device (the fast private accessor;
    constructing a torch.cuda.Stream object costs ~10 us per launch)."""
    try:
        return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())
    except AttributeError:        # accessor renamed/removed: the public route
        return torch.cuda.current_stream().cuda_stream


def _dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported activation dtype {t.dtype}: use float32 or bfloat16") from None


def _check_act(t, name="x"):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: the MI355X kernels need a CUDA/HIP tensor, got {t.device} "
                           "(there is no CPU fallback)")
    if t.dim() != 4:
        raise ValueError(f"{name}: expected [B, C, H, W], got {tuple(t.shape)}")
    if not t.is_contiguous(memory_format=torch.channels_last):
        raise ValueError(f"{name}: activation must be channels-last")
    return t


def to_act(t, dtype=None):
    """Any [B,C,H,W] tensor -> channels-last activation of `dtype` (default: keep)."""
    dtype = dtype or t.dtype
    tracked = t.requires_grad and torch.is_grad_enabled()      # autograd must see the conversion then
    if t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda and dtype in _DT and t.shape[1] > 1 and not tracked:
        B, Cc, H, W = t.shape
        y = torch.empty((B, Cc, H, W), dtype=dtype, device=t.device, memory_format=torch.channels_last)
        call("mis_nchw_f32_to_nhwc", t.data_ptr(), B, Cc, H, W, _DT[dtype], y.data_ptr(), _stream())
        return y
    return t.to(dtype).contiguous(memory_format=torch.channels_last)


def _empty_act(B, Cc, H, W, like):
    return torch.empty((B, Cc, H, W), dtype=like.dtype, device=like.device, memory_format=torch.channels_last)


_WS = {}


def workspace(nbytes, device):
    """One growing scratch buffer per (device, stream).  Every kernel that uses it runs on that
    stream, so consecutive users are ordered by the stream itself."""
    key = (device.type, device.index, _stream() if device.type == "cuda" else 0)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


class private_workspaces:
    """`with private_workspaces() as held:` — scratch buffers requested inside the block are fresh
    ones that the cache forgets again on exit (`held` keeps them alive for the caller).  For graph
    capture: the captured kernels' scratch then comes from the graph's own pool and belongs to the
    captured object only — eager launches that later run on a stream with the same handle, or a
    second capture, get buffers of their own, and a captured kernel never points into a cached
    buffer that a later, larger request would replace and free."""

    def __enter__(self):
        self._saved = dict(_WS)
        _WS.clear()
        self.held = {}
        return self.held

    def __exit__(self, *exc):
        self.held.update(_WS)
        _WS.clear()
        _WS.update(self._saved)
        return False


# ---------------------------------------------------------------------------
# optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg)
# ---------------------------------------------------------------------------
PROFILE = None        # None = off; a dict = collect {family: [(start_event, end_event, flops, bytes), ...]}


class _prof:
    __slots__ = ("name", "flops", "nbytes", "tag", "e0")

    def __init__(self, name, flops, nbytes, tag=None):
        self.name, self.flops, self.nbytes, self.tag = name, flops, nbytes, tag

    def __enter__(self):
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if PROFILE is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            PROFILE.setdefault(self.name, []).append((self.e0, e1, self.flops, self.nbytes, self.tag))
        return False


STREAM_COPY_GBS = 6290.0     # what a float4 device copy reaches on an MI355X (read + write bytes per second; 79 % of the 8 TB/s peak)


def profile_summary(peak_hbm_gbs, peak_tflops, steps=1):
    """Per kernel family: time per step, algorithmic GB/s and TFLOP/s, and the roofline
    object of the family with the largest time share."""
    fams, layers = {}, {}
    for name, recs in (PROFILE or {}).items():
        times = [a.elapsed_time(b) for a, b, _, _, _ in recs]
        L = len(recs) // steps
        if steps >= 3 and L * steps == len(recs):
            # every step issues the same launches in the same order: a launch counts with its MEDIAN over the
            # steps, so that a host stall inside one bracket (the GPU then idles between the two events:
            # tens of ms were seen in a single bracket) does not pass for kernel time
            med = [sorted(times[s_ * L + i] for s_ in range(steps))[steps // 2] for i in range(L)]
            times = med * steps
        ms = sum(times)
        fl = sum(r[2] for r in recs)
        by = sum(r[3] for r in recs)
        for t, r in zip(times, recs):
            if r[4] is not None:            # per-shape lines of the conv families
                e = layers.setdefault(f"{name} {r[4]}", [0.0, 0, 0, 0])
                e[0] += t; e[1] += 1; e[2] += r[2]; e[3] += r[3]
        gbps = by / (ms * 1e-3) / 1e9 if ms > 0 else None
        fams[name] = {"ms_per_step": round(ms / steps, 4), "launches_per_step": len(recs) // steps,
                      "GBps": round(gbps, 1) if gbps is not None else None,
                      "TFLOPs": round(fl / (ms * 1e-3) / 1e12, 2) if ms > 0 and fl else None,
                      # against the HBM peak and against what a plain streaming copy reaches on this part
                      "frac_of_hbm_peak": round(gbps / peak_hbm_gbs, 3) if gbps is not None else None,
                      "frac_of_copy_rate": round(gbps / STREAM_COPY_GBS, 3) if gbps is not None else None,
                      "alg_bytes_per_step": by // steps, "alg_flops_per_step": fl // steps}
    per_layer = {}
    for key, (t, n, fl, by) in layers.items():
        # each shape against ITS roof: the larger of bytes / HBM peak and flops / matrix-core peak
        t_hbm, t_mfma = by / (peak_hbm_gbs * 1e9), fl / (peak_tflops * 1e12)
        per_layer[key] = {"us": round(1e3 * t / n, 1), "launches_per_step": n // steps,
                          "GBps": round(by / (t * 1e-3) / 1e9, 1), "TFLOPs": round(fl / (t * 1e-3) / 1e12, 1),
                          "bound": "mfma" if t_mfma > t_hbm else "hbm",
                          "roofline_frac": round(max(t_hbm, t_mfma) / (t * 1e-3), 3)}
    if not fams:
        return {"dominant": None, "kernels": {}, "conv_layers": {}}
    dom = max(fams, key=lambda k: fams[k]["ms_per_step"])
    d = fams[dom]
    t_hbm = d["alg_bytes_per_step"] / (peak_hbm_gbs * 1e9)
    t_mfma = d["alg_flops_per_step"] / (peak_tflops * 1e12)
    if t_mfma > t_hbm:
        roof = {"bound": "mfma", "achieved": d["TFLOPs"], "peak": peak_tflops, "unit": "TFLOP/s",
                "frac": round(d["TFLOPs"] / peak_tflops, 4)}
    else:
        roof = {"bound": "hbm", "achieved": d["GBps"], "peak": peak_hbm_gbs, "unit": "GB/s",
                "frac": round(d["GBps"] / peak_hbm_gbs, 4)}
    # the same family with every layer shape against ITS OWN roof (matrix cores for the deep layers, HBM for
    # the others): sum of the layers' roof times over the sum of their measured times
    own = [(v["roofline_frac"] * v["us"] * v["launches_per_step"], v["us"] * v["launches_per_step"])
           for k, v in per_layer.items() if k.startswith(dom + " ")]
    if own:
        roof["frac_per_layer_roofs"] = round(sum(a for a, _ in own) / sum(b for _, b in own), 4)
    roof.update({"kernel": dom, "avg_launch_ms": round(d["ms_per_step"] / max(d["launches_per_step"], 1), 4),
                 "launches_per_step": d["launches_per_step"],
                 "algorithmic_bytes_per_launch": d["alg_bytes_per_step"] // max(d["launches_per_step"], 1),
                 "traffic": None})
    return {"dominant": roof, "kernels": fams, "conv_layers": per_layer}


# ---------------------------------------------------------------------------
# raw ops (no autograd)
# ---------------------------------------------------------------------------
def pack_weights(w, dtype, transposed=False):
    """float32 [Cout, Cin, 3, 3] -> the library's private operand image (a byte buffer)."""
    Co, Ci = w.shape[0], w.shape[1]
    nbytes = query("mis_conv3x3_packed_weight_bytes", Co, Ci, int(transposed), _DT[dtype])
    out = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    call("mis_pack_conv3x3_weights", w.data_ptr(), Co, Ci, int(transposed), _DT[dtype], out.data_ptr(), _stream())
    return out


class WeightPacker:
    """Operand images (forward + transposed) of a set of 3x3 conv weights, re-packed by ONE launch
    whenever a weight changed (every optimizer step) instead of one launch per layer; in
    inference nothing is packed after the first call.  The images live in persistent buffers and
    are handed to `ConvBnReluFn` through `packed_for`."""

    def __init__(self, weights, dtype, by_ptr=()):
        """by_ptr: those of `weights` that reach the convs as aliases (see _PACKED_PTR)."""
        from . import _lib
        self.weights = [w for w in weights]
        self.by_ptr = {id(w) for w in by_ptr}
        self.dtype = dtype
        self.epoch = _lib.PATH_EPOCH
        self.versions = None
        self.images = {}
        dev = self.weights[0].device
        jobs, first = [], 0
        for w in self.weights:
            Co, Ci = w.shape[0], w.shape[1]
            nf = query("mis_conv3x3_packed_weight_bytes", Co, Ci, 0, _DT[dtype])
            nt = query("mis_conv3x3_packed_weight_bytes", Co, Ci, 1, _DT[dtype])
            buf = torch.empty(align_up(nf, 256) + nt, dtype=torch.uint8, device=dev)
            wf, wt = buf[:nf], buf[align_up(nf, 256):]
            job = _lib.PackJob()
            nb = query("mis_pack_conv3x3_job", w.data_ptr(), Co, Ci, _DT[dtype], wf.data_ptr(), wt.data_ptr(), first,
                       ctypes.addressof(job))
            if nb > 0:                       # else: this layer is packed on its own by its conv call
                jobs.append(job)
                first += nb
                self.images[w.data_ptr()] = (w, wf, wt)
        self.n_jobs, self.total_blocks = len(jobs), first
        self.table = None
        if jobs:
            raw = b"".join(bytes(j) for j in jobs)
            self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)

    def valid_for(self, weights, dtype):
        from . import _lib
        return (self.dtype == dtype and self.epoch == _lib.PATH_EPOCH and len(weights) == len(self.weights)
                and all(a is b for a, b in zip(weights, self.weights))
                and all(k == w.data_ptr() for k, (w, _, _) in self.images.items()))

    def refresh(self):
        if self.table is None:
            return
        versions = [w._version for w in self.weights]
        if versions == self.versions:
            return
        call("mis_pack_conv3x3_weights_multi", self.table.data_ptr(), self.n_jobs, self.total_blocks,
             _DT[self.dtype], _stream())
        self.versions = versions
        for w, wf, wt in self.images.values():
            table = _PACKED_PTR if id(w) in self.by_ptr else _PACKED
            key = (w.data_ptr() if id(w) in self.by_ptr else id(w), self.dtype)
            table[key] = (weakref.ref(w, lambda _r, k=key, t=table: t.pop(k, None)), w.data_ptr(), w._version, wf, wt,
                          self.epoch)


# (id(weight), dtype) -> (weak reference, data pointer, version, forward image, transposed image,
# path epoch); an entry is used only for the very same tensor object, unchanged since it was packed
_PACKED = {}


# (data pointer, dtype) -> the same record, for the persistent buffers of a ComposedWeights plan: the convs receive
# ALIASES of those buffers (a fresh tensor object per step, sharing memory and version counter with the buffer)
_PACKED_PTR = {}


def packed_for(weight, dtype):
    """The current operand images of `weight` if a WeightPacker produced them, else None."""
    from . import _lib
    hit = _PACKED.get((id(weight), dtype))
    if (hit is not None and hit[0]() is weight and hit[1] == weight.data_ptr() and hit[2] == weight._version
            and hit[5] == _lib.PATH_EPOCH):
        return hit[3], hit[4]
    hit = _PACKED_PTR.get((weight.data_ptr(), dtype))
    if hit is not None:
        base = hit[0]()
        if (base is not None and base.shape == weight.shape and base.data_ptr() == weight.data_ptr()
                and hit[2] == weight._version == base._version and hit[5] == _lib.PATH_EPOCH):
            return hit[3], hit[4]
    return None


# pconv.weight / 1x1 weight data pointer -> (weak reference to the plan, index of the layer in plan.entries)
_COMPOSED = {}
_COMPOSED_W = {}             # data pointer of a plan's W buffer -> the same
DEFER_COMPOSED = True        # (module attribute: the tests compare) gradients of composed weights wait for the end of the pass
BATCH_WEIGHT_PREP = True     # (module attribute: the tests compare with the per-layer launches)


class ComposedWeights:
    """The dense 3x3 weights of every DWConv (W = P*D, b = bp + P bd; modules.py:52-60) and of every 1x1 conv that
    runs on the 3x3 kernels (P at the centre tap), composed by ONE launch whenever a parameter changed, into
    persistent float32 buffers; their operand images — and those of the plain 3x3 convs given in `leaves` — come
    from one more launch (WeightPacker).  DwComposeFn / PointwisePadFn return aliases of the buffers when the
    parameters still have the versions the buffers were composed from, else they compute per layer as before.
    dw: [(dconv, pconv)] nn.Conv2d pairs; pw: [1x1 nn.Conv2d]; leaves: [3x3 nn.Conv2d]."""

    def __init__(self, dw, pw, leaves, dtype):
        from . import _lib
        self.dtype, self.epoch = dtype, _lib.PATH_EPOCH
        self.dw, self.pw, self.leaves = list(dw), list(pw), list(leaves)
        self.params = self._params()
        dev = self.params[0].device
        self.entries = []            # (key pointer, parameter tuple, W, b)
        jobs, first = [], 0
        for dconv, pconv in self.dw:
            Co, Ci = pconv.weight.shape[0], pconv.weight.shape[1]
            W = torch.empty((Co, Ci, 3, 3), dtype=torch.float32, device=dev)
            b = torch.empty(Co, dtype=torch.float32, device=dev)
            nb = max(1, min(64, (Co * Ci * 9 + Co + 255) // 256))
            jobs.append(_lib.ComposeJob(pconv.weight.data_ptr(), dconv.weight.data_ptr(), dconv.bias.data_ptr(),
                                        pconv.bias.data_ptr(), W.data_ptr(), b.data_ptr(), Co, Ci, first, nb))
            first += nb
            self.entries.append((pconv.weight.data_ptr(), (pconv.weight, dconv.weight, dconv.bias, pconv.bias), W, b))
        for conv in self.pw:
            Co, Ci = conv.weight.shape[0], conv.weight.shape[1]
            W = torch.empty((Co, Ci, 3, 3), dtype=torch.float32, device=dev)
            nb = max(1, min(64, (Co * Ci * 9 + 255) // 256))
            jobs.append(_lib.ComposeJob(conv.weight.data_ptr(), None, None, None, W.data_ptr(), None, Co, Ci, first, nb))
            first += nb
            self.entries.append((conv.weight.data_ptr(), (conv.weight,), W, None))
        self.n_jobs, self.total_blocks = len(jobs), first
        self.table = None
        if jobs:
            raw = b"".join(bytes(j) for j in jobs)
            self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        bufs = [e[2] for e in self.entries]
        self.packer = WeightPacker(bufs + [c.weight for c in self.leaves], dtype, by_ptr=bufs)
        self.versions = None

    def _params(self):
        ps = []
        for dconv, pconv in self.dw:
            ps += [pconv.weight, dconv.weight, dconv.bias, pconv.bias]
        return ps + [c.weight for c in self.pw] + [c.weight for c in self.leaves]

    def valid_for(self, dtype):
        from . import _lib
        now = self._params()
        return (self.dtype == dtype and self.epoch == _lib.PATH_EPOCH and len(now) == len(self.params)
                and all(a is b for a, b in zip(now, self.params))
                and all(k == ps[0].data_ptr() for k, ps, _, _ in self.entries))

    def refresh(self):
        versions = [p._version for p in self.params]
        if versions == self.versions:
            return
        if self.table is not None:
            call("mis_dwconv_compose_multi", self.table.data_ptr(), self.n_jobs, self.total_blocks, _stream())
            torch._C._increment_version([e[2] for e in self.entries])
        self.packer.refresh()
        self.versions = versions
        ref = weakref.ref(self)
        # (the registries hold the plan weakly and an index into it: a dead model's buffers and parameters go with it)
        self.composed_from = [(tuple(p._version for p in ps), tuple(p.data_ptr() for p in ps)) for _, ps, _, _ in self.entries]
        for i, (key, ps, W, b) in enumerate(self.entries):
            _COMPOSED[key] = (ref, i)
            _COMPOSED_W[W.data_ptr()] = (ref, i)

    def __del__(self):
        entries = getattr(self, "entries", ())
        if _COMPOSED is None or _COMPOSED_W is None:        # interpreter shutdown: the module's globals are gone
            return
        for table, keys in ((_COMPOSED, [e[0] for e in entries]), (_COMPOSED_W, [e[2].data_ptr() for e in entries])):
            for k in keys:
                hit = table.get(k)
                if hit is not None and hit[0]() in (None, self):
                    table.pop(k, None)


def _composed_for(*params):
    """(W, b) aliases from a live plan whose buffers were composed from exactly these parameter versions, else None."""
    if not BATCH_WEIGHT_PREP:
        return None
    hit = _COMPOSED.get(params[0].data_ptr())
    plan = hit[0]() if hit is not None else None
    if plan is None:
        return None
    versions, ptrs = plan.composed_from[hit[1]]
    if versions != tuple(p._version for p in params) or ptrs != tuple(p.data_ptr() for p in params):
        return None
    _, _, W, b = plan.entries[hit[1]]
    return W.view_as(W), (b.view_as(b) if b is not None else None)


def prepare_weights(model, dtype, dw, pw, leaves):
    """Called at the top of a model's forward: composes / packs all conv weights in two launches (only when a
    parameter changed since the last call).  The plan is kept on the module."""
    if not BATCH_WEIGHT_PREP:
        return
    plan = model.__dict__.get("_mis_weight_plan")
    if plan is None or not plan.valid_for(dtype):
        dwl, pwl, lv = dw(), pw(), leaves()
        ok = lambda t: t is not None and t.is_cuda and t.dtype == torch.float32
        dwl = [(d, p) for d, p in dwl if ok(d.weight) and ok(p.weight) and ok(d.bias) and ok(p.bias)]
        pwl = [c for c in pwl if ok(c.weight)]
        lv = [c for c in lv if ok(c.weight)]
        if not (dwl or pwl or lv):
            return
        plan = ComposedWeights(dwl, pwl, lv, dtype)
        model.__dict__["_mis_weight_plan"] = plan
    plan.refresh()


def align_up(n, a):
    return (n + a - 1) // a * a


def pack_weights_pair(w, dtype):
    """Both operand images of one weight tensor — (forward, transposed = data-gradient operand) —
    packed by one launch."""
    Co, Ci = w.shape[0], w.shape[1]
    nf = query("mis_conv3x3_packed_weight_bytes", Co, Ci, 0, _DT[dtype])
    nt = query("mis_conv3x3_packed_weight_bytes", Co, Ci, 1, _DT[dtype])
    nf_al = (nf + 255) // 256 * 256
    buf = torch.empty(nf_al + nt, dtype=torch.uint8, device=w.device)
    wf, wt = buf[:nf], buf[nf_al:]
    call("mis_pack_conv3x3_weights_pair", w.data_ptr(), Co, Ci, _DT[dtype], wf.data_ptr(), wt.data_ptr(), _stream())
    return wf, wt


def _p(t):
    return t.data_ptr() if t is not None else None


def cat_supported(x, x2, Cout):
    """Whether conv / data gradient / weight gradient can treat [x | x2] as one concatenated input
    without materialising it (else the caller concatenates)."""
    B, Ca, H, W = x.shape
    return bool(query("mis_conv3x3_cat_supported", _dt(x), B, H, W, Ca, x2.shape[1], Cout))


def conv3x3_raw(x, w_packed, bias, Cout, want_stats, x_ss=None, x2=None, x2_ss=None, dilation=1):
    """z = conv3x3(act([x | x2])) + bias; optionally the per-channel (sum, sumsq) of z.
    act(t) = relu(t*scale + shift) per tensor when its ss ([2, C]) is given (lazy activation),
    else t.  x2 (optional) supplies the input channels after x's: the concatenation is virtual."""
    B, Ca, H, W = x.shape
    Cb = x2.shape[1] if x2 is not None else 0
    Cin = Ca + Cb
    z = _empty_act(B, Cout, H, W, x)
    ws = workspace(query("mis_conv3x3_workspace_bytes", B, H, W, Cin, Cout), x.device)
    if dilation != 1 and x2 is not None:
        raise ValueError("dilated convs take one input tensor")
    if want_stats == "partial":
        # un-reduced per-workgroup rows stay at the head of the workspace for bn_finalize, which must
        # be the next user of the workspace on this stream: (tensor, rows)
        stats = ws
        rows = (query("mis_conv3x3_stats_rows", _dt(x), B, H, W, Cin, Cout) if dilation == 1 else
                query("mis_conv3x3_dilated_stats_rows", _dt(x), B, H, W, Cin, Cout, dilation))
    else:
        stats = torch.empty((2, Cout), dtype=torch.float32, device=x.device) if want_stats else None
    es = x.element_size()
    with _prof("conv3x3_fwd", 18 * B * H * W * Cin * Cout, (B * H * W * (Cin + Cout) + 9 * Cin * Cout) * es,
               f"{Cin}->{Cout}@{H}x{W}" + (f" d{dilation}" if dilation != 1 else "")):
        if dilation != 1:
            call("mis_conv3x3_dilated_fwd", x.data_ptr(), _p(x_ss), w_packed.data_ptr(), _p(bias), _dt(x), B, H, W, Cin,
                 Cout, dilation, z.data_ptr(), _p(stats), ws.data_ptr(), _stream())
        elif x2 is None:
            call("mis_conv3x3_fwd", x.data_ptr(), _p(x_ss), w_packed.data_ptr(), _p(bias),
                 _dt(x), B, H, W, Cin, Cout, z.data_ptr(), _p(stats), ws.data_ptr(), _stream())
        else:
            call("mis_conv3x3_cat_fwd", x.data_ptr(), _p(x_ss), x2.data_ptr(), _p(x2_ss), w_packed.data_ptr(),
                 _p(bias), _dt(x), B, H, W, Ca, Cb, Cout, z.data_ptr(), _p(stats), ws.data_ptr(), _stream())
    return z, ((stats, rows) if want_stats == "partial" else stats)


def conv3x3_split_raw(x, w_packed, Ca, Cb):
    """The conv of conv3x3_raw (no bias, no statistics) with its output channels written as two
    tensors [0, Ca) and [Ca, Ca+Cb): the data gradient of a conv over a virtual concatenation."""
    B, Cin, H, W = x.shape
    za, zb = _empty_act(B, Ca, H, W, x), _empty_act(B, Cb, H, W, x)
    ws = workspace(query("mis_conv3x3_workspace_bytes", B, H, W, Cin, Ca + Cb), x.device)
    es = x.element_size()
    with _prof("conv3x3_fwd", 18 * B * H * W * Cin * (Ca + Cb),
               (B * H * W * (Cin + Ca + Cb) + 9 * Cin * (Ca + Cb)) * es, f"{Cin}->{Ca}+{Cb}@{H}x{W}"):
        call("mis_conv3x3_split_fwd", x.data_ptr(), None, w_packed.data_ptr(), None, _dt(x), B, H, W, Cin, Ca, Cb,
             za.data_ptr(), zb.data_ptr(), ws.data_ptr(), _stream())
    return za, zb


# Weight gradients inside a backward pass are needed only when the pass is over (optimizer, gradient
# exchange, user code after backward()): each layer writes its partial slabs into a buffer of its
# own and ONE launch at the end of the pass (or when a gradient bucket is about to be read) reduces
# the slabs of all layers — instead of one 4-10 us reduction launch per layer.
DEFER_WGRAD_REDUCE = True


class _WgradJob(ctypes.Structure):          # mis_wgrad_reduce_job
    _fields_ = [("slabs", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("bias_rows", ctypes.c_void_p), ("dbias", ctypes.c_void_p),
                ("n_slabs", ctypes.c_int), ("Cin", ctypes.c_int), ("Cout", ctypes.c_int), ("n_bias_rows", ctypes.c_int)]


# (device index, stream handle) -> [(job fields, tensors kept alive until the reduction has been enqueued)]:
# a reduction runs on the device and stream its slabs were written on, whatever is current when it is flushed
_wgrad_pending = {}


# The slab launches of a backward pass go to a second stream of the device: nothing on the chain of data
# gradients reads them before the pass ends, so they fill the gaps that chain leaves on the GPU (the
# launch boundaries of its ~150 dependent kernels, their tails, the few-microsecond reductions between
# them).  The inputs of a slab launch stay referenced until the reduction is enqueued, and the
# reduction's stream first waits for the second stream, so the allocator cannot hand their memory out
# while the second stream still reads it.  Off while kernels are being timed one by one (PROFILE) and
# under stream capture (a forked branch inside a HIP graph measured slower than the single chain:
# 4.62 vs 4.52 ms per replayed step).
# (3540 -> 3610 images/s on the benchmark step; MIS_WGRAD_STREAM=0 switches it off.)  History worth keeping: with
# this on, one process in four used to end with non-finite parameters.  The cause was not the streams but a kernel
# of the pass itself: the data-gradient conv with fused BatchNorm sums read its table of (scale, shift, mean, rstd)
# in a leading load-only iteration BEFORE the first barrier of its loop and multiplied what it read by a zero
# gradient — harmless while the LDS still held the same table from an earlier workgroup of the same kernel, NaN
# (0 * NaN) once workgroups of another kernel had used that CU in between, which only happens when two kernels
# run side by side (a second stream, or a second process on the GPU).  Fixed in csrc/conv_mfma.hip (a barrier
# after the tables are filled; a select instead of the product); tools/two_procs.sh and tools/repeat_bench.sh
# reproduce the old failure within seconds on a build without the fix.
WGRAD_SIDE_STREAM = os.environ.get("MIS_WGRAD_STREAM", "1") != "0"
_side_streams = {}         # device index -> torch.cuda.Stream


def _side_stream(dev_index):
    s = _side_streams.get(dev_index)
    if s is None:
        s = _side_streams[dev_index] = torch.cuda.Stream(device=dev_index)
    return s


def flush_wgrad_reduces(on_second_stream=False):
    """Enqueues the pending weight-gradient reductions (one launch per device and stream in use).  Runs by
    itself at the end of every backward pass; call it before reading a weight gradient INSIDE a
    backward pass (a gradient hook, a bucket copy).  on_second_stream: the caller reads the gradients on the
    device's second stream (current, and already waiting for the stream of the pass): the stream of the pass
    then waits for nothing here, and the caller owns the returned references — that memory was allocated on the
    stream of the pass and must stay referenced until that stream has waited for the second one."""
    held = []
    while _wgrad_pending:
        key, pending = _wgrad_pending.popitem()
        dev, st = key
        jobs = (_WgradJob * len(pending))()
        nbytes = 0
        main = None
        n_jobs = 0
        for f, keep in pending:
            if len(keep) > 4:
                main = keep[4]
            if f is None:                   # reduced on the second stream already, right behind its slabs
                continue
            j = jobs[n_jobs]
            n_jobs += 1
            j.slabs, j.dw, j.bias_rows, j.dbias, j.n_slabs, j.Cin, j.Cout, j.n_bias_rows = f
            nbytes += 36 * f[5] * f[6] * (f[4] + 1)
        with torch.cuda.device(dev):
            if on_second_stream:
                st = _side_stream(dev).cuda_stream
                held.append(pending)
            elif main is not None:          # slabs were written on the device's second stream
                main.wait_stream(_side_stream(dev))
            if n_jobs:
                with _prof("wgrad_reduce_multi", 0, nbytes):
                    call("mis_wgrad_reduce_multi", ctypes.addressof(jobs), n_jobs, st)
            held.extend(_flush_compose_bwd(key, st, on_second_stream))
    for key in list(_compose_bwd_pending):          # (none expected: their weight gradients were pending above)
        with torch.cuda.device(key[0]):
            held.extend(_flush_compose_bwd(key, key[1], False))
    _pending_dw.clear()
    return held


# Gradients of COMPOSED weights (ComposedWeights: W = P*D of a DWConv, a 1x1 weight at the centre tap): when the
# weight gradient of such a layer is pending, the chain rule back to the parameters is pending with it — one launch
# per 24 layers right behind the reductions, instead of one per layer inside the pass.
_pending_dw = {}              # data pointer of a pending dw / dbias -> (device index, stream handle) of its reduction
_compose_bwd_pending = {}     # (device index, stream handle) -> [(job fields, tensors kept alive)]


def _flush_compose_bwd(key, st, keep):
    pending = _compose_bwd_pending.pop(key, None)
    if not pending:
        return []
    jobs = (_lib.ComposeBwdJob * len(pending))()
    for j, (f, _) in zip(jobs, pending):
        j.P, j.D, j.bd, j.dW, j.db, j.dP, j.dD, j.dbd, j.Cout, j.Cin = f
    call("mis_dwconv_compose_bwd_multi", ctypes.addressof(jobs), len(pending), st)
    return [pending] if keep else []


def _in_backward_pass():
    try:
        return torch._C._current_graph_task_id() != -1
    except AttributeError:
        return False


def _wgrad_deferred(x, dz, x_ss, x2, x2_ss, bias_rows, dilation, dw, Ca, Cb, Cout, B, H, W, side_ok=True):
    """The slabs-only launch + a pending reduction job; False if the shape is not taken by the
    matrix-core kernels (the caller then uses the one-call form)."""
    Cin = Ca + Cb
    if not query("mis_conv3x3_wgrad_slabs_supported", _dt(x), B, H, W, Ca, Cb, Cout, dilation):
        return None
    slabs = torch.empty(query("mis_conv3x3_wgrad_workspace_bytes", B, H, W, Cin, Cout), dtype=torch.uint8, device=x.device)
    n_slabs = ctypes.c_int(0)
    st = _stream()
    extra = ()
    launch_st = st
    if side_ok and WGRAD_SIDE_STREAM and PROFILE is None and not torch.cuda.is_current_stream_capturing():
        main = torch.cuda.current_stream(x.device)
        side = _side_stream(x.device.index)
        side.wait_stream(main)              # x and dz were written on the current stream
        launch_st = side.cuda_stream
        extra = (main, x, x_ss, x2, x2_ss, dz)
    call("mis_conv3x3_wgrad_slabs", x.data_ptr(), _p(x_ss), _p(x2), _p(x2_ss), dz.data_ptr(), _dt(x), B, H, W, Ca, Cb,
         Cout, dilation, slabs.data_ptr(), ctypes.addressof(n_slabs), launch_st)
    rows, n_rows, dbias = None, 0, None
    if bias_rows is not None:
        rows, n_rows = bias_rows
        dbias = torch.empty(Cout, dtype=torch.float32, device=x.device)
    if not _wgrad_pending:
        torch.autograd.Variable._execution_engine.queue_callback(flush_wgrad_reduces)
    fields = (slabs.data_ptr(), dw.data_ptr(), _p(rows), _p(dbias), n_slabs.value, Cin, Cout, n_rows)
    if extra:
        # On the second stream the reduction follows its slabs at once: nothing waits for it there, and what the
        # stream of the pass meets at the end of the pass is the last layer's reduction instead of all of them.
        job = (_WgradJob * 1)()
        j = job[0]
        j.slabs, j.dw, j.bias_rows, j.dbias, j.n_slabs, j.Cin, j.Cout, j.n_bias_rows = fields
        call("mis_wgrad_reduce_multi", ctypes.addressof(job), 1, launch_st)
        fields = None
    # The pending entry keeps the MEMORY of dw / dbias alive (their storages), not the tensors: autograd
    # takes a returned gradient as .grad without copying only while nobody else refers to the tensor.
    _pending_dw[dw.data_ptr()] = (x.device.index, st)
    if dbias is not None:
        _pending_dw[dbias.data_ptr()] = (x.device.index, st)
    _wgrad_pending.setdefault((x.device.index, st), []).append(
        (fields,
         (slabs, dw.untyped_storage(), rows, dbias.untyped_storage() if dbias is not None else None) + extra))
    return dbias if bias_rows is not None else True


DEFER_SAME_STREAM = 2        # (truthy) defer the reduction, but keep the slab launch on the stream of the pass


def _hw_queues():
    try:
        return int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        return 4


# Parameters with gradient hooks that flush (parallel.GradSync): whether their slab launches go to the second stream
# as well.  With HIP's default of 4 hardware queues per process the streams of such a run (the pass, the second stream,
# the collective library's own) share queues and the second stream then costs time instead of saving it (one-rank
# RCCL group on one GPU: 4.92 vs 4.70 ms per step); with 8 queues it saves as it does without hooks (4.61 vs 4.70).
# Default: on from 8 queues up (the HIP runtime reads GPU_MAX_HW_QUEUES from the environment when it starts);
# MIS_DP_SIDE_STREAM=0/1 overrides.
DP_SIDE_STREAM = (os.environ["MIS_DP_SIDE_STREAM"] != "0") if "MIS_DP_SIDE_STREAM" in os.environ else _hw_queues() >= 8


# Parameters whose gradients are pending in the backward pass `_deferred_pass` (their data pointers).
_deferred_params = set()
_deferred_pass = -1


def can_defer_wgrad(weight, bias=None):
    """Whether nothing reads the gradient of `weight` (and `bias`) before the backward pass ends: a leaf
    without a gradient to accumulate into (autograd then stores the returned tensor as .grad without
    touching its contents) and without hooks — except hooks that flush first (parallel.GradSync marks
    its parameters).  With such hooks the answer is DEFER_SAME_STREAM unless DP_SIDE_STREAM is set (see there).

    A parameter that feeds SEVERAL conv nodes of one pass (a module called twice, siamese branches,
    loss(model(x1)) + loss(model(x2))) is deferred once only: autograd adds the gradients the nodes return as soon
    as the second one arrives, so on a second sighting the pending reductions are enqueued first (the stream of
    the pass then waits for the second stream) and this node computes its gradient at once."""
    global _deferred_pass
    mode = True
    tensors = (weight, bias)
    if not weight.is_leaf:
        # a composed weight (ComposedWeights): its gradient goes through DwComposeFn / PointwisePadFn to the
        # parameters it was composed from — those decide; the slab launch stays on the stream of the pass (the
        # networks that compose weights lose time on the second stream: A_UNet 31.96 vs 31.63 ms)
        hit = _COMPOSED_W.get(weight.data_ptr()) if (DEFER_COMPOSED and BATCH_WEIGHT_PREP) else None
        plan = hit[0]() if hit is not None else None
        if plan is None:
            return False
        tensors = tuple(plan.entries[hit[1]][1]) + ((bias,) if (bias is not None and bias.is_leaf) else ())
        mode = DEFER_SAME_STREAM
    for t in tensors:
        if t is None:
            continue
        if not t.is_leaf or t.grad is not None or t._backward_hooks:
            return False
        if getattr(t, "_post_accumulate_grad_hooks", None):
            if not getattr(t, "_mis_flushing_hooks", False):
                return False
            if not DP_SIDE_STREAM:
                mode = DEFER_SAME_STREAM
    if not weight.is_leaf:
        mode = DEFER_SAME_STREAM
    try:
        task = torch._C._current_graph_task_id()
    except AttributeError:
        return False
    if task == -1:
        return False
    if task != _deferred_pass:
        _deferred_pass = task
        _deferred_params.clear()
    keys = [t.data_ptr() for t in (weight,) + tuple(tensors) if t is not None]
    if any(k in _deferred_params for k in keys):
        flush_wgrad_reduces()
        return False
    _deferred_params.update(keys)
    return mode


def conv3x3_wgrad_raw(x, dz, x_ss=None, x2=None, x2_ss=None, bias_rows=None, dilation=1, defer=False):
    """dW of the conv; with `bias_rows` = (rows tensor [n, Cout], n) also dbias = column sums of
    the rows, reduced inside the same launch.  Returns dw, or (dw, dbias).  With `defer` (and inside
    a backward pass) the reduction is deferred (see DEFER_WGRAD_REDUCE): the returned tensors are
    complete once flush_wgrad_reduces() has run."""
    B, Ca, H, W = x.shape
    Cb = x2.shape[1] if x2 is not None else 0
    Cin = Ca + Cb
    Cout = dz.shape[1]
    dw = torch.empty((Cout, Cin, 3, 3), dtype=torch.float32, device=x.device)
    if defer and DEFER_WGRAD_REDUCE and _in_backward_pass():
        with _prof("conv3x3_wgrad", 18 * B * H * W * Cin * Cout,
                   B * H * W * (Cin + Cout) * x.element_size() + 36 * Cin * Cout,
                   f"{Cin}->{Cout}@{H}x{W}" + (f" d{dilation}" if dilation != 1 else "")):
            r = _wgrad_deferred(x, dz, x_ss, x2, x2_ss, bias_rows, dilation, dw, Ca, Cb, Cout, B, H, W,
                                side_ok=(defer != DEFER_SAME_STREAM))
        if r is not None:
            return dw if bias_rows is None else (dw, r)
    ws = workspace(query("mis_conv3x3_wgrad_workspace_bytes", B, H, W, Cin, Cout), x.device)
    rows, n_rows, dbias = None, 0, None
    if bias_rows is not None:
        rows, n_rows = bias_rows
        dbias = torch.empty(Cout, dtype=torch.float32, device=x.device)
    with _prof("conv3x3_wgrad", 18 * B * H * W * Cin * Cout,
               B * H * W * (Cin + Cout) * x.element_size() + 36 * Cin * Cout,
               f"{Cin}->{Cout}@{H}x{W}" + (f" d{dilation}" if dilation != 1 else "")):
        if dilation != 1:
            call("mis_conv3x3_dilated_wgrad", x.data_ptr(), _p(x_ss), dz.data_ptr(), _dt(x), B, H, W, Cin, Cout,
                 dilation, dw.data_ptr(), _p(rows), n_rows, _p(dbias), ws.data_ptr(), _stream())
        elif x2 is None:
            call("mis_conv3x3_wgrad", x.data_ptr(), _p(x_ss), dz.data_ptr(), _dt(x), B, H, W, Cin, Cout,
                 dw.data_ptr(), _p(rows), n_rows, _p(dbias), ws.data_ptr(), _stream())
        else:
            call("mis_conv3x3_cat_wgrad", x.data_ptr(), _p(x_ss), x2.data_ptr(), _p(x2_ss), dz.data_ptr(), _dt(x),
                 B, H, W, Ca, Cb, Cout, dw.data_ptr(), _p(rows), n_rows, _p(dbias), ws.data_ptr(), _stream())
    return dw if bias_rows is None else (dw, dbias)


def channel_sum_raw(x):
    B, Cc, H, W = x.shape
    out = torch.empty(Cc, dtype=torch.float32, device=x.device)
    ws = workspace(query("mis_channel_sum_workspace_bytes", B * H * W, Cc), x.device)
    with _prof("channel_sum", 0, x.numel() * x.element_size()):
        call("mis_channel_sum", x.data_ptr(), _dt(x), B * H * W, Cc, out.data_ptr(), ws.data_ptr(), _stream())
    return out


# ---------------------------------------------------------------------------
# Conv3x3 + BatchNorm2d + ReLU   (model/utils/modules.py:86-94)
# ---------------------------------------------------------------------------
FUSE_BN_SUMS = True      # producers of `da` that hold z in registers also emit the BatchNorm-backward sums


def attach_bn_sums(da, sums, z):
    """Tags the gradient tensor `da` (gradient with respect to the activation of the lazy tensor
    z) with the BatchNorm-backward sums computed alongside it.  The tag is honoured only if the
    very same tensor object arrives at the producing layer's backward with the version counter it
    had here: when z has several consumers autograd sums their gradients IN PLACE into the first
    one it received (same object, same pointer, tag still attached — but the counter moved), and
    the sums of one consumer alone would be wrong."""
    da._mis_bn_sums = (sums, z.data_ptr(), z._version, da.data_ptr(), da._version)
    return da


def take_bn_sums(da, z):
    tag = getattr(da, "_mis_bn_sums", None)
    if tag is None:
        return None
    sums, zp, zv, dp, dv = tag
    if zp != z.data_ptr() or zv != z._version or dp != da.data_ptr() or dv != da._version:
        return None
    return sums


BN_SUMS_MIN_NUMEL = 15_000_000   # below this the layer's own small reduction pass is as cheap (measured on the full
                              # step: 50e6 -> 15e6 gains 0.4 %, lower changes nothing; tests set it to 0)


def _wants_bn_sums(x_ss, numel):
    return (FUSE_BN_SUMS and x_ss is not None and x_ss.dim() == 2 and x_ss.shape[0] == 4
            and numel >= BN_SUMS_MIN_NUMEL)


# The matrix-core data gradient can also emit the sums (mis_conv3x3_dgrad_bn_sums).  Off: on the full step the conv's
# longer epilogue costs more than the layer's own reduction pass, which reads a gradient that is still in the
# memory-side cache (measured, one process, alternating blocks: U-Net 4.43 -> 4.35 ms, A_UNet 32.65 -> 32.06 ms).
FUSE_DGRAD_BN_SUMS = False


def _dgrad_with_bn_sums(dz, wd, x, x_ss):
    """Data gradient of a conv whose input x is a lazy activation with batch statistics riding
    along (x_ss = [scale, shift, mean, rstd]): the kernel also accumulates the BatchNorm-backward
    sums of x's layer, and the gradient tensor carries them to that layer's backward (which then
    skips its reduction pass).  None when the shape is not taken by the matrix-core kernels."""
    if x_ss is None or x_ss.dim() != 2 or x_ss.shape[0] != 4:
        return None
    B, Cout, H, W = x.shape                 # the data gradient produces x's channels
    Cin = dz.shape[1]
    if not FUSE_DGRAD_BN_SUMS:
        return None
    if not query("mis_conv3x3_dgrad_bn_sums_supported", _dt(x), B, H, W, Cin, Cout):
        return None
    dx = torch.empty_like(x)
    sums = torch.empty((2, Cout), dtype=torch.float32, device=x.device)
    ws = workspace(query("mis_conv3x3_workspace_bytes", B, H, W, Cin, Cout), x.device)
    with _prof("conv3x3_fwd", 18 * B * H * W * Cin * Cout, B * H * W * (Cin + 2 * Cout) * x.element_size(),
               f"{Cin}->{Cout}@{H}x{W} +bn sums"):
        call("mis_conv3x3_dgrad_bn_sums", dz.data_ptr(), wd.data_ptr(), _dt(x), B, H, W, Cin, Cout, dx.data_ptr(),
             x.data_ptr(), x_ss.data_ptr(), sums.data_ptr(), ws.data_ptr(), _stream())
    return attach_bn_sums(dx, sums, x)


class HeadFusion:
    """Couples a conv+BN+ReLU layer with the 1x1 conv that is the ONLY consumer of its output
    (UNet.decoder4 -> out_conv, unet.py:82-84): pass one object as `head=` to both Functions.  In the
    backward pass the 1x1 conv then does not write its data gradient (a full-resolution tensor):
    it leaves its output gradient and weights here, and the layer's BatchNorm backward forms that
    gradient on the fly (mis_bn_relu_bwd_dz_head).  The tensor autograd carries between the two
    nodes is an uninitialised placeholder that must arrive untouched — if the lazy output had
    another consumer autograd would add to it, and the layer's backward raises."""
    __slots__ = ("dy", "w", "sums", "ptr", "version")

    def __init__(self):
        self.dy = self.w = self.sums = None
        self.ptr = self.version = -1


class BranchGradLink:
    """Couples a conv+BN+ReLU layer with the AMSFN fusion node that is the ONLY consumer of its output
    (modules.py:846-868): pass one object as `link=` to the layer and in `links=` to AmsfnFuseFn.  The gradient of
    a branch is dout * w_k + ds_k / HW — the fusion's output gradient and two per-(sample, channel) tables — so
    in the backward pass the fusion writes no branch gradient: it leaves dout, the tables and the layer's
    BatchNorm-backward sums here, and the layer's dz kernel forms the gradient on load
    (mis_bn_relu_bwd_dz_scaled).  The tensor autograd hands the layer is dout itself, as a placeholder that
    must arrive untouched: with another consumer of the branch autograd would add to it, and the layer's
    backward raises."""
    __slots__ = ("gamma", "beta", "g", "scale_ptr", "shift_ptr", "ld", "sums", "keep", "ptr", "version", "want_rows",
                 "dz", "rows")

    def __init__(self):
        self.gamma = self.beta = self.g = self.sums = self.keep = None
        self.scale_ptr = self.shift_ptr = self.ld = 0
        self.ptr = self.version = -1
        self.want_rows = False        # the layer wants its conv-bias gradient (known in its forward pass)
        self.dz = self.rows = None    # FUSE_AMSFN_DZ4: the fusion node has already run the layer's dz kernel


CENTER_TAP_CONVS = True  # 1x1 convs carried as 3x3 multiply their centre tap only (module attribute: the tests compare)
FUSE_AMSFN_BWD = True    # (module attribute: the tests compare with the form that writes the four branch gradients)
FUSE_AMSFN_DZ4 = True    # the four branch layers' dz kernels as ONE launch from the fusion node (the tests compare)


# Order of a layer's two backward convs: weight gradient first.  The data gradient is then the last kernel of the
# layer, and what consumes its output — the next layer's BatchNorm backward, the upsampling or pooling backward —
# runs right behind it, while that tensor is still in the memory-side cache (measured 4.50 -> 4.48 ms per step
# against the other order, identical results).
FUSE_HEAD_DZ = True      # (module attribute: the tests compare the fused and the two-kernel form)


class ConvBnReluFn(torch.autograd.Function):
    """(x, x_ss) -> (z, ss): the activation relu(bn(conv(act(x)))) in LAZY form — the conv output z
    plus the BatchNorm affine ss = [scale, shift]; consumers apply relu(z*scale + shift) when they
    load it, so the normalised activation is never written to memory.  The gradient that flows
    into `z` is the gradient with respect to the ACTIVATION."""

    @staticmethod
    def forward(ctx, x, x_ss, x2, x2_ss, weight, bias, gamma, beta, running_mean, running_var, nbt, training, eps,
                momentum, relu=True, dilation=1, head=None, link=None):
        """x2 (optional, lazy or plain like x): the input is the virtual concatenation [x | x2].
        dilation 0: the weight is a 1x1 conv embedded in a 3x3 (pointwise_as_3x3): forward and data gradient then
        multiply the centre tap only.
        relu=False: BatchNorm without a ReLU of its own (ResDConv's `c1` and `cbr3`, modules.py:432-433,
        444-445): ss is then the affine of bn(z) and the incoming gradient is that of the BatchNorm
        output; only a consumer that knows (functional.add_bn_relu) may read such a pair.
        dilation: 1, 2 or 3 (padding = dilation), modules.py:440-445."""
        _check_act(x)
        B, Ca, H, W = x.shape
        Cin = Ca
        if x2 is not None:
            _check_act(x2, "x2")
            if x2.shape[0] != B or x2.shape[2:] != x.shape[2:] or x2.dtype != x.dtype:
                raise ValueError(f"x2 {tuple(x2.shape)} does not match x {tuple(x.shape)}")
            Cin += x2.shape[1]
        Cout = weight.shape[0]
        if weight.shape != (Cout, Cin, 3, 3):
            raise ValueError(f"weight {tuple(weight.shape)} does not match input channels {Cin}")
        n_pix = B * H * W
        ctx.set_materialize_grads(False)
        pre = packed_for(weight, x.dtype)
        if pre is not None:
            wf, ctx.wt = pre
        elif ctx.needs_input_grad[0] or ctx.needs_input_grad[2]:    # a data gradient will be asked for
            wf, ctx.wt = pack_weights_pair(weight.detach(), x.dtype)
        else:
            wf, ctx.wt = pack_weights(weight.detach(), x.dtype), None
        if dilation == 0 and (x2 is not None or not CENTER_TAP_CONVS
                              or not query("mis_conv3x3_dilated_supported", _dt(x), B, H, W, Cin, Cout, 0)):
            dilation = 1            # (0 is a statement about the weights — only the centre tap is non-zero — not a need)
        if dilation > 1 and not query("mis_conv3x3_dilated_supported", _dt(x), B, H, W, Cin, Cout, dilation):
            raise RuntimeError(f"dilated conv {Cin}->{Cout} (dilation {dilation}) is not taken by the matrix-core kernels")
        z, stats = conv3x3_raw(x, wf, bias.detach() if bias is not None else None, Cout,
                               "partial" if training else False, x_ss, x2, x2_ss, dilation)
        # the lazy handle: rows (scale, shift) are what consumers apply; rows (mean, rstd) ride along
        # for the backward pass (this layer's, possibly fused into the consumer's backward kernel)
        ss = torch.empty((4, Cout), dtype=torch.float32, device=x.device)
        mr = ss[2:]
        st = _stream()
        if training:
            if n_pix <= 1:
                raise ValueError("Expected more than 1 value per channel when training")
            call("mis_bn_finalize", stats[0].data_ptr(), stats[1], Cout, n_pix, gamma.data_ptr(), beta.data_ptr(), eps,
                 momentum,
                 _p(running_mean), _p(running_var), _p(nbt), ss.data_ptr(), mr.data_ptr(), st)
        else:
            call("mis_bn_eval_affine", running_mean.data_ptr(), running_var.data_ptr(), gamma.data_ptr(),
                 beta.data_ptr(), Cout, eps, ss.data_ptr(), st)
            mr[0].copy_(running_mean)
            mr[1].copy_(torch.rsqrt(running_var + eps))
        ctx.save_for_backward(x, x_ss, x2, x2_ss, z, weight, gamma, beta, ss)
        ctx.training = training
        ctx.has_bias = bias is not None
        ctx.bias_ref = weakref.ref(bias) if bias is not None else None
        ctx.relu, ctx.dilation = bool(relu), int(dilation)
        ctx.head = head if (head is not None and relu and FUSE_HEAD_DZ) else None
        ctx.link = link if (link is not None and relu and training and FUSE_AMSFN_BWD) else None
        if ctx.link is not None:
            link.gamma, link.beta, link.g = gamma.detach(), beta.detach(), None
            link.want_rows = bias is not None and ctx.needs_input_grad[5]
            link.dz = link.rows = None
        ctx.mark_non_differentiable(ss)
        return z, ss

    @staticmethod
    def backward(ctx, da, _dss):
        if da is None:
            return (None,) * 18
        x, x_ss, x2, x2_ss, z, weight, gamma, beta, ss = ctx.saved_tensors
        head = ctx.head if (ctx.head is not None and ctx.head.dy is not None) else None
        if head is not None and (da.data_ptr() != head.ptr or da._version != head.version):
            raise RuntimeError("the output of a head-fused conv+BN+ReLU layer had a consumer besides its 1x1 conv")
        link = ctx.link if (ctx.link is not None and ctx.link.g is not None) else None
        if link is not None and (da.data_ptr() != link.ptr or da._version != link.version):
            raise RuntimeError("the output of an AMSFN branch layer had a consumer besides the fusion node")
        mr = ss[2:]
        B, Ca, H, W = x.shape
        Cout, Cin = weight.shape[0], weight.shape[1]
        n_pix = B * H * W
        st = _stream()
        # (sum dy, sum dy*xhat): already produced by the kernel that wrote `da` when that kernel had
        # both z and da in registers (max-pool backward, 1x1-conv backward), else one pass over (z, da)
        relu, dil = ctx.relu, ctx.dilation
        if head is not None:
            sums = head.sums
        elif link is not None:
            sums = link.sums
        else:
            sums = take_bn_sums(da, z) if relu else None       # (the fused sums carry the ReLU mask)
            da = to_act(da, x.dtype)
        ws = workspace(query("mis_bn_relu_bwd_workspace_bytes", n_pix, Cout), x.device)
        if sums is None:
            sums = torch.empty((2, Cout), dtype=torch.float32, device=x.device)
            with _prof("bn_relu_bwd_reduce", 0, 2 * z.numel() * z.element_size()):
                call("mis_bn_relu_bwd_reduce" if relu else "mis_bn_bwd_reduce", z.data_ptr(), da.data_ptr(), _dt(x),
                     n_pix, Cout, mr.data_ptr(), gamma.data_ptr(), beta.data_ptr(), sums.data_ptr(), ws.data_ptr(), st)
        dgamma, dbeta = sums[1], sums[0]          # rows of one buffer; nothing below writes it
        # eval mode: statistics are constants, so the two projection terms vanish
        proj = sums if ctx.training else torch.zeros_like(sums)
        dz = link.dz if (link is not None and link.dz is not None) else torch.empty_like(z)
        want_db = ctx.has_bias and ctx.needs_input_grad[5]
        want_dw = ctx.needs_input_grad[4]
        # The conv-bias gradient (sum of dz) leaves this kernel as per-workgroup rows.  When a weight
        # gradient follows, its reduction launch sums them too (`rows` gets a buffer of its own: the
        # shared workspace is overwritten by the weight-gradient slabs); else they are reduced here.
        dbias = rows = None
        dz_ws = ws
        if link is not None and link.dz is not None:
            # the fusion node ran this layer's dz kernel with its three siblings'; the bias rows came with it
            if want_db and want_dw:
                rows, n_rows = link.rows, link.rows.shape[0]
            elif want_db:
                dbias = link.rows.sum(0)
        elif want_db and want_dw:
            n_rows = query("mis_bn_relu_bwd_dz_rows", _dt(x), n_pix, Cout)
            rows = torch.empty((n_rows, Cout), dtype=torch.float32, device=x.device)
            dz_ws = rows
        elif want_db:
            dbias = torch.empty(Cout, dtype=torch.float32, device=x.device)
        if head is not None:
            hco = head.w.shape[0]
            with _prof("bn_relu_bwd_dz", 0, (2 * z.numel() + n_pix * hco) * z.element_size()):
                call("mis_bn_relu_bwd_dz_head", z.data_ptr(), head.dy.data_ptr(), head.w.data_ptr(), hco, _dt(x), n_pix,
                     Cout, mr.data_ptr(), gamma.data_ptr(), beta.data_ptr(), proj.data_ptr(), dz.data_ptr(),
                     rows.data_ptr() if rows is not None else _p(dbias), dz_ws.data_ptr(), st)
            head.dy = head.w = head.sums = None
        elif link is not None:
            if link.dz is None:
                with _prof("bn_relu_bwd_dz", 0, 3 * z.numel() * z.element_size()):
                    call("mis_bn_relu_bwd_dz_scaled", z.data_ptr(), link.g.data_ptr(), link.scale_ptr, link.shift_ptr,
                         link.ld, H * W, _dt(x), n_pix, Cout, mr.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                         proj.data_ptr(), dz.data_ptr(), rows.data_ptr() if rows is not None else _p(dbias),
                         dz_ws.data_ptr(), st)
            link.g = link.sums = link.keep = link.dz = link.rows = None
        else:
            with _prof("bn_relu_bwd_dz", 0, 3 * z.numel() * z.element_size()):
                call("mis_bn_relu_bwd_dz" if relu else "mis_bn_bwd_dz", z.data_ptr(), da.data_ptr(), _dt(x), n_pix, Cout,
                     mr.data_ptr(),
                     gamma.data_ptr(), beta.data_ptr(), proj.data_ptr(), dz.data_ptr(),
                     rows.data_ptr() if rows is not None else _p(dbias), dz_ws.data_ptr(), st)
        dx = dx2 = None

        def data_gradient():
            nonlocal dx, dx2
            if ctx.needs_input_grad[0] or (x2 is not None and ctx.needs_input_grad[2]):
                wd = ctx.wt if ctx.wt is not None else pack_weights(weight.detach(), x.dtype, transposed=True)
                if x2 is None:
                    dx = _dgrad_with_bn_sums(dz, wd, x, x_ss) if (FUSE_BN_SUMS and dil == 1) else None
                    if dx is None:
                        dx, _ = conv3x3_raw(dz, wd, None, Cin, False, dilation=dil)
                else:
                    dx, dx2 = conv3x3_split_raw(dz, wd, Ca, Cin - Ca)

        # weight gradient first, data gradient last: what consumes the data gradient runs right behind it
        dw = None
        wdil = dil                  # (0: only the centre tap of the weight gradient is computed, the others are zeros)
        if rows is not None:
            # (the bias of a composed weight is a per-step alias that may be gone by now: its parameters decide)
            b_now = ctx.bias_ref() if ctx.bias_ref is not None else None
            dw, dbias = conv3x3_wgrad_raw(x, dz, x_ss, x2, x2_ss, (rows, n_rows), dilation=wdil,
                                          defer=((b_now is not None or not weight.is_leaf)
                                                 and can_defer_wgrad(weight, b_now)))
        elif want_dw:
            dw = conv3x3_wgrad_raw(x, dz, x_ss, x2, x2_ss, dilation=wdil, defer=can_defer_wgrad(weight))
        data_gradient()
        return dx, None, dx2, None, dw, dbias, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None


class MaterializeFn(torch.autograd.Function):
    """Lazy activation (z, ss) -> the plain tensor relu(z*scale + shift).  Identity in activation
    space, so the backward passes the gradient through unchanged."""

    @staticmethod
    def forward(ctx, z, ss):
        B, Cc, H, W = z.shape
        a = torch.empty_like(z)
        with _prof("bn_relu_apply", 0, 2 * z.numel() * z.element_size()):
            call("mis_bn_relu_apply", z.data_ptr(), _dt(z), B * H * W, Cc, ss.data_ptr(), a.data_ptr(), _stream())
        return a

    @staticmethod
    def backward(ctx, da):
        return da, None


def materialize(t, ss):
    return t if ss is None else MaterializeFn.apply(t, ss)


def conv_bn_relu(x, x_ss, conv, bn, training, x2=None, x2_ss=None, head=None):
    """(x, x_ss) [| (x2, x2_ss)] -> lazy (z, ss) with the parameters of an nn.Conv2d /
    nn.BatchNorm2d pair; with x2 the conv input is the virtual concatenation [x | x2].
    head: a HeadFusion shared with the 1x1 conv that is the only consumer of (z, ss)."""
    momentum = BN_MOMENTUM if bn.momentum is None else bn.momentum
    return ConvBnReluFn.apply(x, x_ss, x2, x2_ss, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean,
                              bn.running_var, bn.num_batches_tracked, bool(training), float(bn.eps), float(momentum),
                              True, 1, head)


# ---------------------------------------------------------------------------
# Down / Up   (model/unet.py:19-20, 48-81)
# ---------------------------------------------------------------------------
def dropout2d_noise(layer, x):
    """The multiplier of an nn.Dropout2d `layer` for activation x: None when it is the identity
    (p == 0 or eval mode), else a float32 [B, C, 1, 1] tensor of 0 / 1/(1-p) per (sample, channel),
    drawn exactly as torch's feature dropout draws it (bernoulli_(1-p) on the device generator),
    so a run seeded like a torch run drops the same feature maps.  The Down/Up kernels apply it."""
    p = float(layer.p)
    if p == 0.0 or not layer.training:
        return None
    B, Cc = x.shape[0], x.shape[1]
    if p >= 1.0:
        return torch.zeros((B, Cc, 1, 1), dtype=torch.float32, device=x.device)
    if DROPOUT_RNG == "philox":
        return dropout2d_noise_philox(B, Cc, p, x.device)
    return torch.empty((B, Cc, 1, 1), dtype=torch.float32, device=x.device).bernoulli_(1.0 - p).div_(1.0 - p)


# "torch": the multipliers come from torch's generator exactly as nn.Dropout2d draws them (same seed ->
# the same dropped maps as the reference's module on this device).  "philox": from the library's
# counter-based generator (mis_dropout2d_noise): one launch instead of two, and the value of an element
# depends only on (seed, offset, sample, channel).
DROPOUT_RNG = os.environ.get("MIS_DROPOUT_RNG", "torch")


def dropout2d_noise_philox(B, Cc, p, device, seed=None, offset=None):
    """float32 [B, C, 1, 1] multipliers of one Dropout2d call from the counter-based generator.  Without
    explicit (seed, offset) they are taken from the device's torch generator: its seed, and its Philox
    offset, which is advanced by 4 — so torch.manual_seed reproduces a run and no later torch draw
    reuses the counter."""
    if seed is None:
        if torch.cuda.is_current_stream_capturing():
            raise NotImplementedError("counter-based Dropout2d inside a captured graph needs an explicit (seed, offset)")
        gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
        seed, offset = gen.initial_seed(), gen.get_offset()
        gen.set_offset(offset + 4)
    noise = torch.empty((B, Cc, 1, 1), dtype=torch.float32, device=device)
    call("mis_dropout2d_noise", seed & 0xFFFFFFFFFFFFFFFF, int(offset), B, Cc, float(p), noise.data_ptr(), _stream())
    return noise


def _check_noise(noise, B, Cc):
    if noise is not None and (noise.dtype != torch.float32 or noise.numel() != B * Cc or not noise.is_contiguous()):
        raise ValueError(f"dropout noise must be a contiguous float32 tensor of {B}x{Cc} elements")
    return noise


def _maxpool_fwd(x, x_ss, noise=None):
    _check_act(x)
    B, Cc, H, W = x.shape
    if H % 2 or W % 2:
        raise ValueError(f"max-pool 2x2 needs even H and W, got {H}x{W}")
    _check_noise(noise, B, Cc)
    y = _empty_act(B, Cc, H // 2, W // 2, x)
    with _prof("maxpool2x2_fwd", 0, (x.numel() + y.numel()) * x.element_size()):
        call("mis_maxpool2x2_fwd", x.data_ptr(), _p(x_ss), _p(noise), _dt(x), B, H, W, Cc, y.data_ptr(), _stream())
    return y


def _maxpool_bwd(x, x_ss, dy, dx_add=None, noise=None):
    """Gradient of the pooled activation routed to each window's first maximum, plus `dx_add`
    (a second gradient of the same tensor); autograd's gradient buffers are never written."""
    B, Cc, H, W = x.shape
    dy = to_act(dy, x.dtype)
    if dx_add is not None:
        dx_add = to_act(dx_add, x.dtype)
    dx = torch.empty_like(x)
    nbytes = (2 * x.numel() + dy.numel() + (x.numel() if dx_add is not None else 0)) * x.element_size()
    sums = ws = None
    if _wants_bn_sums(x_ss, x.numel()):
        sums = torch.empty((2, Cc), dtype=torch.float32, device=x.device)
        ws = workspace(query("mis_maxpool2x2_bwd_workspace_bytes", Cc), x.device)
    with _prof("maxpool2x2_bwd", 0, nbytes):
        call("mis_maxpool2x2_bwd", x.data_ptr(), _p(x_ss), _p(noise), dy.data_ptr(), _p(dx_add), _dt(x), B, H, W, Cc,
             dx.data_ptr(), _p(sums), _p(ws), _stream())
    return attach_bn_sums(dx, sums, x) if sums is not None else dx


class MaxPool2x2Fn(torch.autograd.Function):
    """nn.MaxPool2d(2, 2) of a lazy or plain activation, optionally followed by nn.Dropout2d
    (`noise` from `dropout2d_noise`) — unet.py:48-61."""

    @staticmethod
    def forward(ctx, x, x_ss, noise=None):
        y = _maxpool_fwd(x, x_ss, noise)
        ctx.save_for_backward(x, x_ss, noise)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, x_ss, noise = ctx.saved_tensors
        return _maxpool_bwd(x, x_ss, dy, None, noise), None, None


class PoolSkipFn(torch.autograd.Function):
    """An encoder output used twice: max-pooled (then Dropout2d, `noise`) for the next level,
    and (as the very same tensor) as the decoder's skip connection — unet.py:47-60,67-81.
    (x, x_ss) -> (pooled, skip) with skip an alias of x; the backward receives both gradients at
    once and sums them inside the pooling backward kernel."""

    @staticmethod
    def forward(ctx, x, x_ss, noise=None):
        y = _maxpool_fwd(x, x_ss, noise)
        ctx.save_for_backward(x, x_ss, noise)
        ctx.set_materialize_grads(False)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        x, x_ss, noise = ctx.saved_tensors
        if dy is None:
            return dskip, None, None
        return _maxpool_bwd(x, x_ss, dy, dskip, noise), None, None


class UpsampleCatFn(torch.autograd.Function):
    """cat([bilinear_x2(dropout2d(low)), skip], dim=1) — upsampled channels first
    (unet.py:66-67); either input may be a lazy activation, `noise` is the Dropout2d multiplier
    of `low` (unet.py:64,69,74) or None."""

    @staticmethod
    def forward(ctx, low, low_ss, skip, skip_ss, noise=None):
        _check_act(low, "low")
        _check_act(skip, "skip")
        B, Cup, h, w = low.shape
        Bs, Cskip, Hs, Ws = skip.shape
        if (Bs, Hs, Ws) != (B, 2 * h, 2 * w) or skip.dtype != low.dtype:
            raise ValueError(f"skip {tuple(skip.shape)} does not match upsampled {(B, Cup, 2 * h, 2 * w)}")
        _check_noise(noise, B, Cup)
        out = _empty_act(B, Cup + Cskip, 2 * h, 2 * w, low)
        with _prof("upsample2x_cat_fwd", 0, (low.numel() + skip.numel() + out.numel()) * low.element_size()):
            call("mis_upsample2x_cat_fwd", low.data_ptr(), _p(low_ss), _p(noise), skip.data_ptr(), _p(skip_ss),
                 _dt(low), B, h, w, Cup, Cskip, out.data_ptr(), _stream())
        ctx.dims = (B, h, w, Cup, Cskip)
        ctx.save_for_backward(noise)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, h, w, Cup, Cskip = ctx.dims
        noise, = ctx.saved_tensors
        dout = to_act(dout)
        dlow = _empty_act(B, Cup, h, w, dout)
        dskip = _empty_act(B, Cskip, 2 * h, 2 * w, dout)
        with _prof("upsample2x_cat_bwd", 0, (dout.numel() + dlow.numel() + dskip.numel()) * dout.element_size()):
            call("mis_upsample2x_cat_bwd", dout.data_ptr(), _p(noise), _dt(dout), B, h, w, Cup, Cskip,
                 dlow.data_ptr(), dskip.data_ptr(), _stream())
        return dlow, None, dskip, None, None


class UpsampleFn(torch.autograd.Function):
    """bilinear x2 (align_corners=True) of a lazy or plain activation (after its Dropout2d, when
    `noise` is given) — unet.py:20,66 — for the decoder convs that read [up | skip] as two tensors."""

    @staticmethod
    def forward(ctx, low, low_ss, noise=None):
        _check_act(low, "low")
        B, Cup, h, w = low.shape
        _check_noise(noise, B, Cup)
        out = _empty_act(B, Cup, 2 * h, 2 * w, low)
        with _prof("upsample2x_cat_fwd", 0, (low.numel() + out.numel()) * low.element_size()):
            call("mis_upsample2x_cat_fwd", low.data_ptr(), _p(low_ss), _p(noise), None, None, _dt(low), B, h, w, Cup,
                 0, out.data_ptr(), _stream())
        ctx.dims = (B, h, w, Cup)
        ctx.save_for_backward(noise)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, h, w, Cup = ctx.dims
        noise, = ctx.saved_tensors
        dout = to_act(dout)
        dlow = _empty_act(B, Cup, h, w, dout)
        with _prof("upsample2x_cat_bwd", 0, (dout.numel() + dlow.numel()) * dout.element_size()):
            call("mis_upsample2x_cat_bwd", dout.data_ptr(), _p(noise), _dt(dout), B, h, w, Cup, 0, dlow.data_ptr(),
                 None, _stream())
        return dlow, None, None


# ---------------------------------------------------------------------------
# AMSFN pieces (model/utils/modules.py:791-871)
# ---------------------------------------------------------------------------
def conv_bn_relu_w(x, x_ss, weight, bias, bn, training, link=None, pointwise=False):
    """Like `conv_bn_relu` with an explicit [Cout, Cin, 3, 3] weight / bias (any autograd tensors).
    link: a BranchGradLink shared with the AmsfnFuseFn that is the only consumer of (z, ss).
    pointwise: `weight` came from pointwise_as_3x3 (only its centre tap is non-zero)."""
    momentum = BN_MOMENTUM if bn.momentum is None else bn.momentum
    return ConvBnReluFn.apply(x, x_ss, None, None, weight, bias, bn.weight, bn.bias, bn.running_mean,
                              bn.running_var, bn.num_batches_tracked, bool(training), float(bn.eps), float(momentum),
                              True, 0 if pointwise else 1, None, link)


class PointwisePadFn(torch.autograd.Function):
    """[Cout, Cin, 1, 1] -> the 3x3 weight with that matrix at the centre tap, taken from the model's
    ComposedWeights plan (no launch here); the gradient is the centre tap of the 3x3 gradient."""

    @staticmethod
    def forward(ctx, weight, W):
        return W.view_as(W)

    @staticmethod
    def backward(ctx, dW):
        if dW is None:
            return None, None
        key = _pending_dw.get(dW.data_ptr()) if dW.is_contiguous() else None
        if key is None:
            return dW[:, :, 1:2, 1:2].contiguous(), None
        Cout, Cin = dW.shape[0], dW.shape[1]
        dP = torch.empty((Cout, Cin, 1, 1), dtype=torch.float32, device=dW.device)
        _compose_bwd_pending.setdefault(key, []).append(
            ((None, None, None, dW.data_ptr(), None, dP.data_ptr(), None, None, Cout, Cin), (dW, dP.untyped_storage())))
        return dP, None


def pointwise_as_3x3(weight):
    """A 1x1 conv weight [Cout, Cin, 1, 1] as the 3x3 weight whose only non-zero tap is the centre:
    the 1x1 conv runs on the matrix-core 3x3 kernels (same sums, in the same order per tap)."""
    hit = _composed_for(weight) if (weight.is_cuda and weight.dtype == torch.float32) else None
    if hit is not None:
        return PointwisePadFn.apply(weight, hit[0])
    return torch.nn.functional.pad(weight, (1, 1, 1, 1))


class DwComposeFn(torch.autograd.Function):
    """(pconv.weight, dconv.weight, dconv.bias, pconv.bias) -> (W [Cout,Cin,3,3], b [Cout]) of the
    equivalent dense conv, and the chain rule back: two small launches instead of a dozen ATen ones."""

    @staticmethod
    def forward(ctx, P, D, bd, bp):
        Cout, Cin = P.shape[0], P.shape[1]
        Pc, Dc = P.detach().reshape(Cout, Cin).contiguous(), D.detach().reshape(Cin, 9).contiguous()
        bdc, bpc = bd.detach().contiguous(), bp.detach().contiguous()
        hit = _composed_for(P, D, bd, bp)
        if hit is not None:                  # composed with every other layer's, at the top of the model's forward
            W, b = hit
        else:
            W = torch.empty((Cout, Cin, 3, 3), dtype=torch.float32, device=P.device)
            b = torch.empty(Cout, dtype=torch.float32, device=P.device)
            call("mis_dwconv_compose_fwd", Pc.data_ptr(), Dc.data_ptr(), bdc.data_ptr(), bpc.data_ptr(), Cout, Cin,
                 W.data_ptr(), b.data_ptr(), _stream())
        ctx.save_for_backward(Pc, Dc, bdc)
        ctx.shapes = (P.shape, D.shape)
        return W, b

    @staticmethod
    def backward(ctx, dW, db):
        Pc, Dc, bdc = ctx.saved_tensors
        Cout, Cin = Pc.shape
        dev = Pc.device
        dW = torch.zeros((Cout, Cin, 3, 3), device=dev) if dW is None else dW.contiguous()
        db = torch.zeros(Cout, device=dev) if db is None else db.contiguous()
        dP = torch.empty((Cout, Cin), dtype=torch.float32, device=dev)
        dD = torch.empty((Cin, 9), dtype=torch.float32, device=dev)
        dbd = torch.empty(Cin, dtype=torch.float32, device=dev)
        key = _pending_dw.get(dW.data_ptr())
        if key is not None:                 # dW (and db) are complete only after the pending reductions: wait with them
            _compose_bwd_pending.setdefault(key, []).append(
                ((Pc.data_ptr(), Dc.data_ptr(), bdc.data_ptr(), dW.data_ptr(), db.data_ptr(), dP.data_ptr(),
                  dD.data_ptr(), dbd.data_ptr(), Cout, Cin),
                 # (storages, not tensors, of what autograd receives: it takes a gradient as .grad without a copy
                 # only while nobody else refers to the tensor — and a copy made now would copy unwritten memory)
                 (Pc, Dc, bdc, dW, db.untyped_storage(), dP.untyped_storage(), dD.untyped_storage(),
                  dbd.untyped_storage())))
        else:
            call("mis_dwconv_compose_bwd", Pc.data_ptr(), Dc.data_ptr(), bdc.data_ptr(), dW.data_ptr(), db.data_ptr(),
                 Cout, Cin, dP.data_ptr(), dD.data_ptr(), dbd.data_ptr(), _stream())
        ps, ds = ctx.shapes
        return dP.reshape(ps), dD.reshape(ds), dbd, db


def dwconv_as_3x3(dconv, pconv):
    """DWConv (modules.py:52-60) = pointwise(depthwise(x)): the composition is ONE dense 3x3 conv
    with weight W[o,i,ky,kx] = P[o,i] * D[i,ky,kx] and bias P @ bd + bp (the depthwise bias is
    added at every pixel, padding or not, so it folds into a constant).  The products are formed
    from the parameters (a few thousand elements) with autograd attached; the activation-sized
    work is a single pass of the matrix-core conv kernels instead of two memory-bound passes."""
    if (dconv.bias is not None and pconv.bias is not None and pconv.weight.is_cuda
            and pconv.weight.dtype == torch.float32 and dconv.weight.dtype == torch.float32):
        return DwComposeFn.apply(pconv.weight, dconv.weight, dconv.bias, pconv.bias)
    P = pconv.weight[:, :, 0, 0]
    W = P[:, :, None, None] * dconv.weight[:, 0][None]
    b = pconv.bias
    if dconv.bias is not None:
        fold = P @ dconv.bias
        b = fold if b is None else b + fold
    return W, b


class MaxPool3x3Fn(torch.autograd.Function):
    """nn.MaxPool2d(kernel_size=3, stride=1, padding=1) of a lazy or plain activation."""

    @staticmethod
    def forward(ctx, x, x_ss):
        _check_act(x)
        B, Cc, H, W = x.shape
        y = torch.empty_like(x)
        arg = torch.empty(x.numel(), dtype=torch.uint8, device=x.device)
        with _prof("maxpool3x3_fwd", 0, 2 * x.numel() * x.element_size() + x.numel()):
            call("mis_maxpool3x3s1_fwd", x.data_ptr(), _p(x_ss), _dt(x), B, H, W, Cc, y.data_ptr(), arg.data_ptr(),
                 _stream())
        ctx.save_for_backward(arg)
        ctx.dims = (B, Cc, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        arg, = ctx.saved_tensors
        B, Cc, H, W = ctx.dims
        dy = to_act(dy)
        dx = torch.empty_like(dy)
        with _prof("maxpool3x3_bwd", 0, 2 * dy.numel() * dy.element_size() + dy.numel()):
            call("mis_maxpool3x3s1_bwd", dy.data_ptr(), arg.data_ptr(), _dt(dy), B, H, W, Cc, dx.data_ptr(), _stream())
        return dx, None


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() if t is not None else None for t in tensors])


class FanOutFn(torch.autograd.Function):
    """One activation (lazy or plain) read by n consumers: returns n aliases of it and, in the backward pass, sums the
    consumers' gradients in ONE kernel instead of autograd's pairwise additions (a full pass each) — the four branches
    of an AMSFN block read its input (model/utils/modules.py:846-856), an encoder output of A_UNet feeds its block and
    the decoder's skip (model/a_unet.py:66-105).  With `bn_sums` and a training-mode lazy input the same kernel returns
    the BatchNorm-backward sums of the producing layer (the summed gradient is in registers, z is read alongside), and
    the gradient tensor carries them to that layer's backward, which then skips its reduction pass; pass
    bn_sums=False when the activation has consumers outside this fan-out (their gradients are added later)."""

    @staticmethod
    def forward(ctx, x, x_ss, n, bn_sums):
        ctx.set_materialize_grads(False)
        ctx.bn_sums = bool(bn_sums)
        ctx.save_for_backward(x, x_ss)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        x, x_ss = ctx.saved_tensors
        gs = [to_act(g, x.dtype) for g in grads if g is not None]
        if not gs:
            return None, None, None, None
        if len(gs) == 1:
            return gs[0], None, None, None
        if len(gs) > 5 or any(g.shape != x.shape for g in gs):
            out = gs[0]
            for g in gs[1:]:
                out = out + g
            return out, None, None, None
        B, Cc, H, W = x.shape
        out = torch.empty_like(x)
        # (any size: the extra read of z costs one pass of this kernel, the reduction pass it replaces two and a launch)
        want = (ctx.bn_sums and FUSE_BN_SUMS and x_ss is not None and x_ss.dim() == 2 and x_ss.shape[0] == 4
                and Cc % (8 if x.dtype == torch.bfloat16 else 4) == 0)
        sums = torch.empty((2, Cc), dtype=torch.float32, device=x.device) if want else None
        ws = workspace(query("mis_sum_gradients_workspace_bytes", Cc), x.device) if want else None
        es = x.element_size()
        with _prof("sum_gradients", 0, (len(gs) + 1 + (1 if want else 0)) * x.numel() * es):
            call("mis_sum_gradients", _ptr_array(gs), len(gs), _dt(x), B * H * W, Cc, x.data_ptr() if want else None,
                 _p(x_ss) if want else None, out.data_ptr(), _p(sums), _p(ws), _stream())
        if want:
            attach_bn_sums(out, sums, x)
        return out, None, None, None


FUSE_FAN_OUT = True      # (module attribute: the tests compare with autograd's own pairwise accumulation)


def fan_out(x, x_ss, n, bn_sums=True):
    """n aliases of the activation (x, x_ss) for n consumers; see FanOutFn."""
    if n <= 1 or not FUSE_FAN_OUT or not (torch.is_grad_enabled() and x.requires_grad):
        return [x] * n
    return list(FanOutFn.apply(x, x_ss, n, bn_sums))


class AmsfnFuseFn(torch.autograd.Function):
    """Global average pools of the four branch outputs -> squeeze MLP -> sigmoid -> weighted sum
    (modules.py:846-868) in one autograd node: three launches forward, five backward.
    links: four BranchGradLink objects shared with the layers that produced the branches (each branch read by
    this node only): the backward then writes no branch gradients (see BranchGradLink)."""

    @staticmethod
    def forward(ctx, z1, s1, z2, s2, z3, s3, z4, s4, W1, b1, W2, b2, links=None):
        zs, sss = [z1, z2, z3, z4], [s1, s2, s3, s4]
        for t in zs:
            _check_act(t)
            if t.shape != z1.shape or t.dtype != z1.dtype:
                raise ValueError("AMSFN branches must have equal shapes and dtypes")
        B, Cc, H, W = z1.shape
        if W1.shape[:2] != (Cc, 4 * Cc) or W2.shape[:2] != (4 * Cc, Cc):
            raise ValueError(f"MLP weights {tuple(W1.shape)}, {tuple(W2.shape)} do not match {Cc} channels")
        dev = z1.device
        W1c, W2c = W1.detach().reshape(Cc, 4 * Cc).contiguous(), W2.detach().reshape(4 * Cc, Cc).contiguous()
        s = torch.empty((B, 4 * Cc), dtype=torch.float32, device=dev)
        h = torch.empty((B, Cc), dtype=torch.float32, device=dev)
        w = torch.empty((B, 4 * Cc), dtype=torch.float32, device=dev)
        out = torch.empty_like(z1)
        ws = workspace(query("mis_amsfn_workspace_bytes", B, H * W, Cc), dev)
        with _prof("amsfn_fuse_fwd", 0, 9 * z1.numel() * z1.element_size()):
            call("mis_amsfn_fuse_fwd", _ptr_array(zs), _ptr_array(sss), _dt(z1), B, H * W, Cc, W1c.data_ptr(),
                 b1.detach().contiguous().data_ptr(), W2c.data_ptr(), b2.detach().contiguous().data_ptr(),
                 s.data_ptr(), h.data_ptr(), w.data_ptr(), out.data_ptr(), ws.data_ptr(), _stream())
        ctx.save_for_backward(z1, s1, z2, s2, z3, s3, z4, s4, W1c, W2c, s, h, w)
        ctx.shapes = (W1.shape, W2.shape)
        ctx.links = links
        return out

    @staticmethod
    def backward(ctx, dout):
        z1, s1, z2, s2, z3, s3, z4, s4, W1c, W2c, s, h, w = ctx.saved_tensors
        zs, sss = [z1, z2, z3, z4], [s1, s2, s3, s4]
        B, Cc, H, W = z1.shape
        dev = z1.device
        dout = to_act(dout, z1.dtype)
        dW1 = torch.empty((Cc, 4 * Cc), dtype=torch.float32, device=dev)
        dW2 = torch.empty((4 * Cc, Cc), dtype=torch.float32, device=dev)
        db1 = torch.empty(Cc, dtype=torch.float32, device=dev)
        db2 = torch.empty(4 * Cc, dtype=torch.float32, device=dev)
        s1s, s2s = ctx.shapes
        links = ctx.links
        if (FUSE_AMSFN_BWD and links is not None and len(links) == 4 and all(l is not None and l.gamma is not None for l in links)
                and all(t is not None and t.dim() == 2 and t.shape[0] == 4 for t in sss) and B * H * W < 2 ** 31):
            bn_sums = torch.empty((4, 2, Cc), dtype=torch.float32, device=dev)
            shift = torch.empty((B, 4 * Cc), dtype=torch.float32, device=dev)
            ws = workspace(query("mis_amsfn_lazy_workspace_bytes", B, H * W, Cc), dev)
            with _prof("amsfn_fuse_bwd", 0, 5 * z1.numel() * z1.element_size()):
                call("mis_amsfn_fuse_bwd_lazy", _ptr_array(zs), _ptr_array(sss), _ptr_array([l.gamma for l in links]),
                     _ptr_array([l.beta for l in links]), dout.data_ptr(), _dt(z1), B, H * W, Cc, W1c.data_ptr(),
                     W2c.data_ptr(), s.data_ptr(), h.data_ptr(), w.data_ptr(), bn_sums.data_ptr(), shift.data_ptr(),
                     dW1.data_ptr(), db1.data_ptr(), dW2.data_ptr(), db2.data_ptr(), ws.data_ptr(), _stream())
            for k, l in enumerate(links):
                l.g, l.sums, l.keep = dout, bn_sums[k], (w, shift)
                l.scale_ptr, l.shift_ptr, l.ld = w.data_ptr() + 4 * k * Cc, shift.data_ptr() + 4 * k * Cc, 4 * Cc
                l.ptr, l.version = dout.data_ptr(), dout._version
            if FUSE_AMSFN_DZ4:
                # the four layers' dz kernels read the same dout: one launch, dout fetched once
                n_pix = B * H * W
                n_rows = query("mis_bn_relu_bwd_dz_rows", _dt(z1), n_pix, Cc)
                dzs = [torch.empty_like(z1) for _ in range(4)]
                rows = [torch.empty((n_rows, Cc), dtype=torch.float32, device=dev) if l.want_rows else None for l in links]
                with _prof("bn_relu_bwd_dz", 0, 9 * z1.numel() * z1.element_size()):
                    call("mis_bn_relu_bwd_dz_scaled4", _ptr_array(zs), dout.data_ptr(), w.data_ptr(), shift.data_ptr(),
                         4 * Cc, H * W, _dt(z1), n_pix, Cc, _ptr_array([t[2:] for t in sss]),
                         _ptr_array([l.gamma for l in links]), _ptr_array([l.beta for l in links]), bn_sums.data_ptr(),
                         _ptr_array(dzs), _ptr_array(rows), _stream())
                for k, l in enumerate(links):
                    l.dz, l.rows = dzs[k], rows[k]
            return (dout, None, dout, None, dout, None, dout, None, dW1.reshape(s1s), db1, dW2.reshape(s2s), db2, None)
        dact = [torch.empty_like(z1) for _ in range(4)]
        ws = workspace(query("mis_amsfn_workspace_bytes", B, H * W, Cc), dev)
        with _prof("amsfn_fuse_bwd", 0, 10 * z1.numel() * z1.element_size()):
            call("mis_amsfn_fuse_bwd", _ptr_array(zs), _ptr_array(sss), dout.data_ptr(), _dt(z1), B, H * W, Cc,
                 W1c.data_ptr(), W2c.data_ptr(), s.data_ptr(), h.data_ptr(), w.data_ptr(), _ptr_array(dact),
                 dW1.data_ptr(), db1.data_ptr(), dW2.data_ptr(), db2.data_ptr(), ws.data_ptr(), _stream())
        return (dact[0], None, dact[1], None, dact[2], None, dact[3], None, dW1.reshape(s1s), db1, dW2.reshape(s2s),
                db2, None)


class Conv1x1Fn(torch.autograd.Function):
    """nn.Conv2d(Cin, n_classes, 1) — unet.py:44,84; the input may be a lazy activation."""

    @staticmethod
    def forward(ctx, x, x_ss, weight, bias, head=None):
        _check_act(x)
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        # head: see HeadFusion (needs the batch statistics of a training-mode lazy input, few classes)
        ctx.head = head if (head is not None and FUSE_HEAD_DZ and x_ss is not None and x_ss.dim() == 2
                            and x_ss.shape[0] == 4 and Cout <= 8) else None
        y = _empty_act(B, Cout, H, W, x)
        with _prof("conv1x1_fwd", 2 * B * H * W * Cin * Cout, (x.numel() + y.numel()) * x.element_size()):
            call("mis_conv1x1_fwd", x.data_ptr(), _p(x_ss), _dt(x), B * H * W, Cin, Cout, weight.data_ptr(),
                 _p(bias), y.data_ptr(), _stream())
        ctx.save_for_backward(x, x_ss, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, x_ss, weight = ctx.saved_tensors
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        dy = to_act(dy, x.dtype)
        head = ctx.head if ctx.needs_input_grad[0] else None
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty((Cout, Cin, 1, 1), dtype=torch.float32, device=x.device)
        db = torch.empty(Cout, dtype=torch.float32, device=x.device)
        ws = workspace(query("mis_conv1x1_bwd_workspace_bytes", B * H * W, Cin, Cout), x.device)
        sums = None
        if dx is not None and (head is not None or _wants_bn_sums(x_ss, x.numel())):
            sums = torch.empty((2, Cin), dtype=torch.float32, device=x.device)
        nbytes = ((1 if head is not None else 2) * x.numel() + dy.numel()) * x.element_size()
        with _prof("conv1x1_bwd", 4 * B * H * W * Cin * Cout, nbytes):
            call("mis_conv1x1_bwd", x.data_ptr(), _p(x_ss), dy.data_ptr(), _dt(x), B * H * W, Cin, Cout,
                 weight.data_ptr(), None if head is not None else _p(dx), dw.data_ptr(), db.data_ptr(), _p(sums),
                 ws.data_ptr(), _stream())
        if head is not None:
            # dx stays uninitialised: the producing layer's backward forms it from (dy, weight)
            head.dy, head.w, head.sums = dy, weight.detach().reshape(Cout, Cin), sums
            head.ptr, head.version = dx.data_ptr(), dx._version
        elif sums is not None:
            attach_bn_sums(dx, sums, x)
        return dx, None, dw, (db if ctx.has_bias else None), None


# ---------------------------------------------------------------------------
# loss + metrics   (utils/loss_fn.py:63-88,199-203; utils/metrics.py:24-36)
# ---------------------------------------------------------------------------
def _logit_strides(logits):
    """(stride_b, stride_c, stride_p) in elements for a [B,C,H,W] tensor whose H,W are
    jointly contiguous (NCHW-contiguous or channels-last); else a contiguous copy."""
    B, Cc, H, W = logits.shape
    sb, sc, sh, sw = logits.stride()
    if sh == W * sw and (B == 1 or sb >= 0):
        return logits, sb, sc, sw
    logits = logits.contiguous()
    sb, sc, sh, sw = logits.stride()
    return logits, sb, sc, sw


def loss_metrics_raw(logits, target, want_sums=True, want_ce=False, want_cm=True):
    """One pass over the logits: pooled Dice sums [3,C], CE map [B,H,W], confusion
    counts [B,C,C] (int64) and the device status word (non-zero: target out of range)."""
    if not logits.is_cuda:
        raise RuntimeError("loss/metrics kernels need a CUDA/HIP tensor (there is no CPU fallback)")
    if logits.dim() != 4:
        raise ValueError(f"logits: expected [B, C, H, W], got {tuple(logits.shape)}")
    if logits.dtype not in _DT:
        logits = logits.float()
    B, Cc, H, W = logits.shape
    if target.shape != (B, H, W):
        raise ValueError(f"target {tuple(target.shape)} does not match logits {(B, H, W)}")
    target = target.to(torch.int64).contiguous()
    logits, sb, sc, sp = _logit_strides(logits)
    dev = logits.device
    sums = torch.empty((3, Cc), dtype=torch.float32, device=dev) if want_sums else None
    ce = torch.empty((B, H, W), dtype=torch.float32, device=dev) if want_ce else None
    cm = torch.empty((B, Cc, Cc), dtype=torch.int64, device=dev) if want_cm else None
    status = torch.empty(1, dtype=torch.int32, device=dev)
    ws = workspace(query("mis_loss_metrics_workspace_bytes", B, Cc, H * W), dev)
    p = lambda t: t.data_ptr() if t is not None else None
    with _prof("loss_metrics_fwd", 0, B * H * W * (Cc * logits.element_size() + 8 + (4 if want_ce else 0))):
        call("mis_loss_metrics_fwd", logits.data_ptr(), _dt(logits), sb, sc, sp, target.data_ptr(), B, Cc, H * W,
             p(sums), p(ce), p(cm), status.data_ptr(), ws.data_ptr(), _stream())
    return sums, ce, cm, status, logits, target


LOSS_DICE, LOSS_IOU = 0, 1


class DiceLossFn(torch.autograd.Function):
    """losses[C] = (per-foreground-class Dice — or IoU, `kind` — losses..., total) and optionally
    the CE map.

    `group` (a torch.distributed process group, or True for the default group): GLOBAL-batch
    loss of a data-parallel run.  The reference pools its three sums over the whole batch before
    forming the ratio (utils/loss_fn.py:79-80), so the mean of per-shard losses is a different
    number; here the [3, C] partial sums of all ranks are added (one all-reduce of 3*C floats)
    between the fused pass and the ratio, every rank returns the same global loss, and the
    backward uses the global sums.  Each rank then holds the gradient of the global loss with
    respect to ITS logits; a gradient exchange that AVERAGES over ranks (GradSync) would return
    1/world of the true parameter gradient, so the backward scales by the world size."""

    @staticmethod
    def forward(ctx, logits, target, smooth, want_ce, kind=LOSS_DICE, group=None):
        sums, ce, cm, status, lg, tg = loss_metrics_raw(logits, target, True, want_ce, True)
        Cc = lg.shape[1]
        ctx.grad_scale = 1.0
        if group is not None:
            import torch.distributed as dist
            pg = None if group is True else group
            if dist.is_initialized() and dist.get_world_size(pg) > 1:
                dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=pg)
                ctx.grad_scale = float(dist.get_world_size(pg))
        losses = torch.empty(Cc, dtype=torch.float32, device=lg.device)
        call("mis_dice_from_sums", sums.data_ptr(), Cc, smooth, kind, losses.data_ptr(), _stream())
        ctx.save_for_backward(lg, tg, sums)
        ctx.smooth = smooth
        ctx.kind = kind
        ctx.in_dtype = logits.dtype
        ctx.mark_non_differentiable(cm, status)
        ctx.set_materialize_grads(False)      # (no zero tensors for the outputs nobody differentiates)
        ctx.has_ce = ce is not None
        return losses, ce, cm, status

    @staticmethod
    def backward(ctx, dlosses, dce, _dcm, _dstatus):
        lg, tg, sums = ctx.saved_tensors
        if dlosses is None and dce is None:
            return None, None, None, None, None, None
        B, Cc, H, W = lg.shape
        _, sb, sc, sp = _logit_strides(lg)
        w = (dlosses if dlosses is not None else torch.zeros(Cc, device=lg.device)).to(torch.float32).contiguous()
        if ctx.grad_scale != 1.0:
            w = w * ctx.grad_scale
        dce_p = None
        if ctx.has_ce and dce is not None:
            dce = dce.to(torch.float32).contiguous()
            dce_p = dce.data_ptr()
        dlogits = torch.empty_like(lg)
        with _prof("loss_bwd", 0, B * H * W * (2 * Cc * lg.element_size() + 8)):
            call("mis_loss_bwd", lg.data_ptr(), _dt(lg), sb, sc, sp, tg.data_ptr(), B, Cc, H * W, sums.data_ptr(),
                 w.data_ptr(), ctx.smooth, ctx.kind, dce_p, dlogits.data_ptr(), _stream())
        if dlogits.dtype != ctx.in_dtype:
            dlogits = dlogits.to(ctx.in_dtype)
        return dlogits, None, None, None, None, None


# ---------------------------------------------------------------------------
# RDAM_UNet pieces (model/rdam_unet.py, model/utils/modules.py:426-453, model/utils/attention.py:147-190)
# ---------------------------------------------------------------------------
def conv_bn(x, x_ss, weight, bias, bn, training, relu=True, dilation=1, x2=None, x2_ss=None):
    """conv3x3 (any dilation 1..3, explicit [Cout, Cin, 3, 3] weight) + BatchNorm2d [+ ReLU] -> (z, ss)."""
    momentum = BN_MOMENTUM if bn.momentum is None else bn.momentum
    return ConvBnReluFn.apply(x, x_ss, x2, x2_ss, weight, bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                              bn.num_batches_tracked, bool(training), float(bn.eps), float(momentum), bool(relu),
                              int(dilation))


class Conv3x3Fn(torch.autograd.Function):
    """nn.Conv2d(Cin, Cout, 3, padding=1) with bias and nothing after it (RDAM_UNet.encoder1 /
    decoder4, rdam_unet.py:22,52): (x, x_ss) [| (x2, x2_ss)] -> plain z."""

    @staticmethod
    def forward(ctx, x, x_ss, x2, x2_ss, weight, bias):
        _check_act(x)
        Cout = weight.shape[0]
        pre = packed_for(weight, x.dtype)
        need_dx = ctx.needs_input_grad[0] or ctx.needs_input_grad[2]
        if pre is not None:
            wf, ctx.wt = pre
        elif need_dx:
            wf, ctx.wt = pack_weights_pair(weight.detach(), x.dtype)
        else:
            wf, ctx.wt = pack_weights(weight.detach(), x.dtype), None
        z, _ = conv3x3_raw(x, wf, bias.detach() if bias is not None else None, Cout, False, x_ss, x2, x2_ss)
        ctx.save_for_backward(x, x_ss, x2, x2_ss, weight)
        ctx.has_bias = bias is not None
        return z

    @staticmethod
    def backward(ctx, dz):
        x, x_ss, x2, x2_ss, weight = ctx.saved_tensors
        B, Ca, H, W = x.shape
        Cout, Cin = weight.shape[0], weight.shape[1]
        dz = to_act(dz, x.dtype)
        dx = dx2 = None
        if ctx.needs_input_grad[0] or (x2 is not None and ctx.needs_input_grad[2]):
            wd = ctx.wt if ctx.wt is not None else pack_weights(weight.detach(), x.dtype, transposed=True)
            if x2 is None:
                dx, _ = conv3x3_raw(dz, wd, None, Cin, False)
            else:
                dx, dx2 = conv3x3_split_raw(dz, wd, Ca, Cin - Ca)
        dw = conv3x3_wgrad_raw(x, dz, x_ss, x2, x2_ss, defer=can_defer_wgrad(weight)) if ctx.needs_input_grad[4] else None
        db = channel_sum_raw(dz) if (ctx.has_bias and ctx.needs_input_grad[5]) else None
        return dx, None, dx2, None, dw, db


class AddBnReluFn(torch.autograd.Function):
    """relu(bn_a(za) + bn_b(zb)) from two (conv output, BatchNorm affine) pairs produced with
    relu=False — the tail of ResDConv.forward (modules.py:447-453).  Plain output; the backward
    hands dout*[out > 0] to both BatchNorm outputs."""

    @staticmethod
    def forward(ctx, za, ssa, zb, ssb):
        _check_act(za, "za")
        _check_act(zb, "zb")
        if za.shape != zb.shape or za.dtype != zb.dtype:
            raise ValueError(f"shapes differ: {tuple(za.shape)} vs {tuple(zb.shape)}")
        B, Cc, H, W = za.shape
        out = torch.empty_like(za)
        with _prof("add_bn_relu_fwd", 0, 3 * za.numel() * za.element_size()):
            call("mis_add_bn_relu_fwd", za.data_ptr(), ssa.data_ptr(), zb.data_ptr(), ssb.data_ptr(), _dt(za),
                 B * H * W, Cc, out.data_ptr(), _stream())
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, dout):
        out, = ctx.saved_tensors
        B, Cc, H, W = out.shape
        dout = to_act(dout, out.dtype)
        g = torch.empty_like(out)
        with _prof("relu_mask_bwd", 0, 3 * out.numel() * out.element_size()):
            call("mis_relu_mask_bwd", out.data_ptr(), dout.data_ptr(), _dt(out), B * H * W, Cc, g.data_ptr(), _stream())
        return g, None, g, None


def add_bn_relu(a, b):
    """a, b: (z, ss) pairs from conv_bn(..., relu=False)."""
    return AddBnReluFn.apply(a[0], a[1], b[0], b[1])


class WindowSumsFn(torch.autograd.Function):
    """x [B, C, H, W] -> S float32 [B, C, 9]: per (sample, channel) the sum of x over the pixels
    each tap of a zero-padded 3x3 conv reads, so that avgpool(conv3x3(x)) = bias + W.S/(H*W)
    without running the conv (MDAM's squeeze path, attention.py:184-185)."""

    @staticmethod
    def forward(ctx, x):
        _check_act(x)
        B, Cc, H, W = x.shape
        S = torch.empty((B, Cc, 9), dtype=torch.float32, device=x.device)
        ws = workspace(query("mis_mdam_workspace_bytes", B, Cc), x.device)
        with _prof("mdam_window_sums", 0, x.numel() * x.element_size()):
            call("mis_mdam_window_sums", x.data_ptr(), _dt(x), B, H, W, Cc, S.data_ptr(), ws.data_ptr(), _stream())
        ctx.dims = (B, Cc, H, W)
        ctx.meta = (x.dtype, x.device)
        return S

    @staticmethod
    def backward(ctx, dS):
        B, Cc, H, W = ctx.dims
        dx = torch.empty((B, Cc, H, W), dtype=ctx.meta[0], device=ctx.meta[1], memory_format=torch.channels_last)
        dS = dS.to(torch.float32).contiguous()
        with _prof("mdam_window_sums_bwd", 0, dx.numel() * dx.element_size()):
            call("mis_mdam_window_sums_bwd", dS.data_ptr(), _dt(dx), B, H, W, Cc, dx.data_ptr(), _stream())
        return dx


class ChannelScaleFn(torch.autograd.Function):
    """x * s[b, c] (s float32 [B, C]): `group_x * weights.sigmoid()` of attention.py:189."""

    @staticmethod
    def forward(ctx, x, s):
        _check_act(x)
        B, Cc, H, W = x.shape
        s = s.to(torch.float32).contiguous()
        out = torch.empty_like(x)
        with _prof("channel_scale", 0, 2 * x.numel() * x.element_size()):
            call("mis_channel_scale", x.data_ptr(), s.data_ptr(), _dt(x), B, H * W, Cc, out.data_ptr(), _stream())
        ctx.save_for_backward(x, s)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, s = ctx.saved_tensors
        B, Cc, H, W = x.shape
        dout = to_act(dout, x.dtype)
        dx = ds = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            with _prof("channel_scale", 0, 2 * x.numel() * x.element_size()):
                call("mis_channel_scale", dout.data_ptr(), s.data_ptr(), _dt(x), B, H * W, Cc, dx.data_ptr(), _stream())
        if ctx.needs_input_grad[1]:
            ds = torch.empty((B, Cc), dtype=torch.float32, device=x.device)
            ws = workspace(query("mis_mdam_workspace_bytes", B, Cc), x.device)
            with _prof("channel_dot", 0, 2 * x.numel() * x.element_size()):
                call("mis_channel_dot", x.data_ptr(), dout.data_ptr(), _dt(x), B, H * W, Cc, ds.data_ptr(),
                     ws.data_ptr(), _stream())
        return dx, ds
