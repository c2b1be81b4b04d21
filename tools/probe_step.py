#!/usr/bin/env python3
"""Which kernel produces the first non-finite value?  Runs training steps of the benchmark model with a
probe kernel (tools/micro/libpollute.so: counts non-finite values, no allocation, no synchronisation) behind
the outputs of the conv / pooling / upsampling wrappers, and reports the first probe that fired.
Start two of these at once to see them disturb each other:  python tools/probe_step.py [steps]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from medical_image_segmentation_amd import functional as F_  # noqa: E402
from medical_image_segmentation_amd.model.unet import UNet  # noqa: E402
from medical_image_segmentation_amd.utils.loss_fn import diceloss  # noqa: E402
from medical_image_segmentation_amd.utils.optim import clip_and_step  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
dev = torch.device("cuda:0")
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "micro", "libpollute.so"))
lib.probe_nonfinite.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
lib.probe_checksum.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
flags = torch.zeros(1 << 17, dtype=torch.int32, device=dev)
sums = torch.zeros(1 << 17, dtype=torch.int64, device=dev)
save_to = sys.argv[2] if len(sys.argv) > 2 else None          # write the checksums of every probed tensor here
labels = []
step_no = [0]


def probe(label, t):
    if t is None or not torch.is_tensor(t) or len(labels) >= flags.numel():
        return
    if t.dtype not in (torch.bfloat16, torch.float32):
        return
    lib.probe_nonfinite(t.data_ptr(), t.numel(), 1 if t.dtype == torch.bfloat16 else 0, flags.data_ptr(), len(labels),
                        F_._stream())
    if save_to and t.is_contiguous() or (save_to and t.is_contiguous(memory_format=torch.channels_last)):
        lib.probe_checksum(t.data_ptr(), t.numel() * t.element_size(), sums.data_ptr(), len(labels), F_._stream())
    labels.append((step_no[0], label, tuple(t.shape)))


def wrap(name, outs):
    orig = getattr(F_, name)

    def f(*a, **k):
        if outs:
            for i, t in enumerate(a[:outs]):
                if torch.is_tensor(t):
                    probe(f"{name} in{i}", t)
        r = orig(*a, **k)
        rs = r if isinstance(r, tuple) else (r,)
        for i, t in enumerate(rs):
            if torch.is_tensor(t):
                probe(f"{name} out{i}", t)
            elif isinstance(t, (tuple, list)):
                for j, u in enumerate(t):
                    if torch.is_tensor(u):
                        probe(f"{name} out{i}.{j}", u)
        return r
    setattr(F_, name, f)


for n in (() if os.environ.get("PROBE_ONLY_GRADS") else
          ("conv3x3_raw", "conv3x3_split_raw", "_dgrad_with_bn_sums", "_maxpool_fwd", "_maxpool_bwd", "conv3x3_wgrad_raw")):
    wrap(n, 2 if n in ("conv3x3_split_raw", "conv3x3_raw", "_dgrad_with_bn_sums") else None)

def probe_raw(label, ptr, nbytes):
    if len(labels) >= flags.numel() or not save_to:
        return
    lib.probe_checksum(ptr, nbytes, sums.data_ptr(), len(labels), F_._stream())
    labels.append((step_no[0], label, (nbytes,)))


_call = F_.call


def call_probed(name, *a):
    if name == "mis_bn_finalize" and not os.environ.get("PROBE_ONLY_GRADS"):
        probe_raw(f"bn_finalize rows in ({a[1]} x 2 x {a[2]})", a[0], a[1] * 2 * a[2] * 4)
        r = _call(name, *a)
        probe_raw(f"bn_finalize ss out (C={a[2]})", a[11], 4 * a[2] * 4)
        return r
    if name == "mis_upsample2x_cat_fwd" and os.environ.get("PROBE_UP_TWICE"):
        B_, h_, w_, Cup_, Csk_ = a[6], a[7], a[8], a[9], a[10]
        nbytes = B_ * 4 * h_ * w_ * (Cup_ + Csk_) * 2
        r = _call(name, *a)
        probe_raw("up first", a[11], nbytes)
        scratch = _scratch.setdefault(nbytes, torch.empty(nbytes, dtype=torch.uint8, device=dev))
        _call(name, *(a[:11] + (scratch.data_ptr(),) + a[12:]))
        probe_raw("up again", scratch.data_ptr(), nbytes)
        if os.environ.get("PROBE_UP_WHERE") and len(_where) < 3:
            import ctypes as _c
            n16 = nbytes // 2
            first = torch.frombuffer((_c.c_char * 0).from_address(0), dtype=torch.uint8) if False else None
            o1 = _as_tensor(a[11], n16).view(B_, 2 * h_, 2 * w_, Cup_ + Csk_)
            o2 = scratch.view(torch.int16)[:n16].view(B_, 2 * h_, 2 * w_, Cup_ + Csk_)
            neq = o1 != o2
            if bool(neq.any()):
                idx = torch.nonzero(neq)
                vals = [(hex(int(o1[tuple(i)]) & 0xFFFF), hex(int(o2[tuple(i)]) & 0xFFFF)) for i in idx[:24].tolist()]
                _where.append((step_no[0], (B_, h_, w_, Cup_), idx.shape[0], idx[:24].tolist(), vals, idx[-3:].tolist(),
                               sorted(set(idx[:, 3].tolist()))[:16], sorted(set(idx[:, 0].tolist())),
                               (int(idx[:, 1].min()), int(idx[:, 1].max())), (int(idx[:, 2].min()), int(idx[:, 2].max()))))
        return r
    if name == "mis_maxpool2x2_fwd" and os.environ.get("PROBE_UP_TWICE"):
        nbytes = a[4] * (a[5] // 2) * (a[6] // 2) * a[7] * 2
        r = _call(name, *a)
        probe_raw("pool first", a[8], nbytes)
        scratch = _scratch.setdefault(nbytes, torch.empty(nbytes, dtype=torch.uint8, device=dev))
        _call(name, *(a[:8] + (scratch.data_ptr(),) + a[9:]))
        probe_raw("pool again", scratch.data_ptr(), nbytes)
        return r
    return _call(name, *a)


_scratch = {}
_where = []
_live = {}


def _as_tensor(ptr, n16):
    """int16 view of a device buffer given by its address (the tensor that owns it is found among the live ones)"""
    for t in _live.values():
        if t.data_ptr() == ptr:
            return t.permute(0, 2, 3, 1).view(torch.int16).reshape(-1)[:n16]        # (physical order: channels last)
    raise KeyError(ptr)


_empty_act = F_._empty_act


def _empty_act_tracked(*a, **k):
    t = _empty_act(*a, **k)
    _live[t.data_ptr()] = t.detach()
    if len(_live) > 64:
        _live.pop(next(iter(_live)))
    return t


F_._empty_act = _empty_act_tracked
F_.call = call_probed

torch.manual_seed(1234)
model = UNet(1, 4, 0, base_channels=32, compute_dtype=torch.bfloat16).to(dev).train()
opt = torch.optim.AdamW(model.parameters(), lr=8e-4, betas=(0.95, 0.999), weight_decay=1e-4)
g = torch.Generator(device="cpu").manual_seed(100)
x = torch.randn(16, 1, 512, 512, generator=g).to(dev)
t = torch.randint(0, 4, (16, 512, 512), generator=g).to(dev)
loss_fn = diceloss(["a", "b", "c"])
params = list(model.parameters())
loss_log = torch.zeros(steps, device=dev)
for s in range(steps):
    step_no[0] = s
    opt.zero_grad()
    logits = model(x)
    if not os.environ.get("PROBE_ONLY_GRADS"):
        probe("logits", logits)
    ld = loss_fn(logits, t)
    ld["total_loss"].backward()
    if not os.environ.get("PROBE_FINAL_ONLY"):
        for n, p in model.named_parameters():
            probe("grad " + n, p.grad)
    clip_and_step(opt, params, max_norm=1.0)
    loss_log[s].copy_(ld["total_loss"].detach())
if os.environ.get("PROBE_FINAL_ONLY"):
    for n, p in model.named_parameters():
        probe("final " + n, p.detach())
torch.cuda.synchronize()
if os.environ.get("PROBE_UP_TWICE"):
    sm = sums[:len(labels)].cpu().tolist()
    for what in ("up", "pool"):
        pairs = [(i, labels[i]) for i in range(len(labels) - 1)
                 if labels[i][1] == what + " first" and labels[i + 1][1] == what + " again" and sm[i] != sm[i + 1]]
        tot = sum(1 for l in labels if l[1] == what + " first")
        print(f"pid {os.getpid()}: {what}: launches whose immediate repetition gave another result: {len(pairs)} of {tot}", pairs[:4])
for w_ in _where:
    print("WHERE", w_)
try:
    import ctypes as _c
    from medical_image_segmentation_amd import _lib as _L
    cnt = (_c.c_uint * 8)()
    _L.load().mis_debug_up_counters(cnt)
    print(f"pid {os.getpid()}: up_dbg counters [lds a, b, c, d re-read differs; registers differ; memory differs; -; lanes]:", list(cnt))
except AttributeError:
    pass
if save_to:
    import json
    json.dump({"labels": [list(map(str, l)) for l in labels], "sums": sums[:len(labels)].cpu().tolist(),
               "losses": [float(v).hex() for v in loss_log.cpu()]}, open(save_to, "w"))
f = flags.cpu().tolist()
first = next((i for i in range(len(labels)) if f[i]), None)
if first is None:
    print(f"pid {os.getpid()}: no non-finite value in {len(labels)} probes over {steps} steps; loss {float(ld['total_loss']):.6f}")
else:
    lo = max(0, first - 6)
    print(f"pid {os.getpid()}: first non-finite at probe {first}: {labels[first]} count {f[first]}")
    for i in range(lo, min(len(labels), first + 10)):
        print("    ", i, labels[i], f[i])
