"""clip_grad_norm_ + AdamW.step() in two kernel launches.

The reference's step is `torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)` followed by
`optimizer.step()` on a torch.optim.AdamW (utils/train_and_eval.py:115-118, train.py:221-226).
`clip_and_step` does the same arithmetic on the optimizer's own state tensors (`exp_avg`,
`exp_avg_sq`, `step`), so the optimizer object, its `state_dict()` and checkpoints stay
interchangeable with torch's.  Anything it does not recognise — another optimizer class,
amsgrad/maximize/capturable, sparse or non-float32 tensors, an enabled GradScaler — runs
torch's own `clip_grad_norm_` + `optimizer.step()` unchanged.
"""
import torch

from .. import functional as F_
from .._lib import call, query

CHUNK = 4096          # MIS_OPT_CHUNK in include/misamd.h


class _Plan:
    """Device table of (param, grad, exp_avg, exp_avg_sq, n) rows and the chunk prefix of one
    optimizer.  Gradients are re-allocated every step (`zero_grad()` drops them), at addresses that
    need not repeat, so the pointer rows are refreshed whenever they changed: written into one of
    a small ring of pinned staging buffers (allocated once; a slot is reused only after the copy
    that last read it has completed) and uploaded on the launch stream — no allocation, no host
    synchronisation.  The chunk prefix depends on the tensor sizes only and is built once."""
    RING = 4

    def __init__(self, sizes, group_chunks, prefix, dev):
        self.sizes = sizes
        self.group_chunks = group_chunks
        self.n_chunks = prefix[-1]
        self.rows = None
        n = len(sizes)
        self.descs = torch.empty((n, 5), dtype=torch.int64, device=dev)
        self.prefix = torch.tensor(prefix, dtype=torch.int32).pin_memory().to(dev, non_blocking=True)
        self.stage = [torch.empty((n, 5), dtype=torch.int64).pin_memory() for _ in range(self.RING)]
        self.done = [None] * self.RING
        self.slot = 0

    def upload(self, rows):
        global PLAN_UPLOADS
        PLAN_UPLOADS += 1
        k = self.slot
        self.slot = (k + 1) % self.RING
        if self.done[k] is not None:
            self.done[k].synchronize()
        self.stage[k].copy_(torch.tensor(rows, dtype=torch.int64))
        self.descs.copy_(self.stage[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.done[k] = ev
        self.rows = rows


_PLANS = {}           # id(optimizer) -> _Plan
PLAN_BUILDS = 0       # tables allocated / pointer-row uploads so far (diagnostics, tests)
PLAN_UPLOADS = 0


def _fusable(optimizer, params):
    if type(optimizer) is not torch.optim.AdamW:
        return False
    listed = set()
    for g in optimizer.param_groups:
        if g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable"):
            return False
        if isinstance(g["lr"], torch.Tensor):
            return False
        for p in g["params"]:
            if p.grad is None:
                continue
            if (not p.is_cuda or p.dtype != torch.float32 or p.grad.dtype != torch.float32 or p.grad.is_sparse
                    or not p.is_contiguous() or not p.grad.is_contiguous()):
                return False
            listed.add(id(p))
    # the clip norm must be taken over exactly the tensors being stepped
    clip = {id(p) for p in params if p.grad is not None}
    return clip == listed and len(listed) > 0


class _Fast:
    """What `clip_and_step` established about an (optimizer, params) pair on its first fused call,
    so that later calls only re-check what can change between steps."""

    def __init__(self):
        self.param_ids = None
        self.groups = None        # [(group dict, [params])]
        self.states = None        # [state dict per param], same order as the pointer rows
        self.static = None        # [(p ptr, exp_avg ptr, exp_avg_sq ptr, numel)]
        self.step_tensors = None
        self.step = None          # the value every step tensor holds


_FAST = {}


def _group_ok(g):
    return not (g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable")
                or isinstance(g["lr"], torch.Tensor))


def _fast_path(optimizer, params):
    """The cached description if nothing structural changed since it was built, else None."""
    f = _FAST.get(id(optimizer))
    if f is None or f.param_ids != [id(p) for p in params]:
        return None
    state = optimizer.state
    for (g, ps) in f.groups:
        if not _group_ok(g):
            return None
    for p, s, (sp, sm, sv, _) in zip((p for _, ps in f.groups for p in ps), f.states, f.static):
        gr = p.grad
        if (gr is None or gr.dtype != torch.float32 or not gr.is_contiguous() or state.get(p) is not s
                or p.data_ptr() != sp or s["exp_avg"].data_ptr() != sm or s["exp_avg_sq"].data_ptr() != sv):
            return None
    if float(f.step_tensors[0]) != f.step:
        return None
    return f


def clip_and_step(optimizer, params, max_norm=1.0):
    """clip_grad_norm_(params, max_norm) + optimizer.step().  Returns the total gradient norm (a
    0-dim tensor), like clip_grad_norm_."""
    params = [p for p in params]
    F_.flush_wgrad_reduces()       # (a no-op after a finished backward pass)
    fast = _fast_path(optimizer, params)
    if fast is not None:
        return _launch(optimizer, fast, max_norm)
    _FAST.pop(id(optimizer), None)
    return _clip_and_step_checked(optimizer, params, max_norm)


def _launch(optimizer, f, max_norm):
    dev = f.groups[0][1][0].device
    rows = [(sp, p.grad.data_ptr(), sm, sv, n)
            for p, (sp, sm, sv, n) in zip((p for _, ps in f.groups for p in ps), f.static)]
    plan = _PLANS[id(optimizer)]
    if plan.rows != rows:
        plan.upload(rows)
    st = F_._stream()
    ws = F_.workspace(plan.ws_bytes, dev)
    norm = torch.empty((), dtype=torch.float32, device=dev)
    n_t = len(rows)
    call("mis_grad_sqsum", plan.descs.data_ptr(), plan.prefix.data_ptr(), n_t, plan.n_chunks, ws.data_ptr(), st)
    step = int(f.step) + 1
    for (g, ps), (c0, c1) in zip(f.groups, plan.group_chunks):
        b1, b2 = g["betas"]
        call("mis_clip_adamw_step", plan.descs.data_ptr(), plan.prefix.data_ptr(), n_t, c0, c1, plan.n_chunks,
             ws.data_ptr(), float(max_norm), float(g["lr"]), float(b1), float(b2), float(g["eps"]),
             float(g["weight_decay"]), step, 1, norm.data_ptr() if c0 == 0 else None, st)
    torch._foreach_add_(f.step_tensors, 1)
    f.step = float(step)
    # the kernels wrote through raw pointers: tell autograd (and anything keyed on tensor versions,
    # e.g. cached operand images of the weights) that these tensors changed
    touched = []
    for _, ps in f.groups:
        touched += ps
        touched += [p.grad for p in ps]
    for s in f.states:
        touched += [s["exp_avg"], s["exp_avg_sq"]]
    torch._C._increment_version(touched)
    return norm


def _clip_and_step_checked(optimizer, params, max_norm):
    """Full validation of the (optimizer, params) pair; on success the first fused step and the
    cached description for the following ones."""
    if not _fusable(optimizer, params):
        norm = torch.nn.utils.clip_grad_norm_(params, max_norm=max_norm)
        optimizer.step()
        return norm
    groups, steps = [], []
    for g in optimizer.param_groups:
        ps = [p for p in g["params"] if p.grad is not None]
        if not ps:
            continue
        st = None
        for p in ps:
            s = optimizer.state[p]
            if len(s) == 0:                 # same lazy initialisation as torch.optim.AdamW
                s["step"] = torch.tensor(0.0, dtype=torch.float32)
                s["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                s["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            k = float(s["step"])
            if st is None:
                st = k
            elif k != st:                   # per-tensor step counts differ: leave it to torch
                norm = torch.nn.utils.clip_grad_norm_(params, max_norm=max_norm)
                optimizer.step()
                return norm
        groups.append((g, ps))
        steps.append(int(st) + 1)
    dev = groups[0][1][0].device
    rows = []
    for g, ps in groups:
        for p in ps:
            s = optimizer.state[p]
            rows.append((p.data_ptr(), p.grad.data_ptr(), s["exp_avg"].data_ptr(), s["exp_avg_sq"].data_ptr(), p.numel()))
    sizes = tuple(r[4] for r in rows)
    plan = _PLANS.get(id(optimizer))
    if plan is None or plan.sizes != sizes or plan.descs.device != dev:
        global PLAN_BUILDS
        PLAN_BUILDS += 1
        prefix, c, group_chunks = [], 0, []
        for g, ps in groups:
            c0 = c
            for p in ps:
                prefix.append(c)
                c += (p.numel() + CHUNK - 1) // CHUNK
            group_chunks.append((c0, c))
        prefix.append(c)
        plan = _PLANS[id(optimizer)] = _Plan(sizes, group_chunks, prefix, dev)
    plan.ws_bytes = query("mis_clip_adamw_workspace_bytes", plan.n_chunks)
    if len(set(steps)) != 1:              # groups at different step counts: torch handles that
        norm = torch.nn.utils.clip_grad_norm_(params, max_norm=max_norm)
        optimizer.step()
        return norm
    f = _Fast()
    f.param_ids = [id(p) for p in params]
    f.groups = groups
    f.states = [optimizer.state[p] for _, ps in groups for p in ps]
    f.static = [(r[0], r[2], r[3], r[4]) for r in rows]
    f.step_tensors = [s_["step"] for s_ in f.states]
    f.step = float(steps[0] - 1)
    _FAST[id(optimizer)] = f
    return _launch(optimizer, f, max_norm)


class CapturedStep:
    """clip_grad_norm_ + AdamW.step() in a form that can sit inside a HIP graph: the launches take
    everything that changes from step to step (learning rate, bias corrections) from a small
    device buffer, so one capture serves every replay.

    `launch()` is called once, inside the capture (the gradients exist by then; their addresses
    are those of every replay).  Before each replay `prepare()` forms the step's constants on the
    host — the same double arithmetic the eager path uses, so both give the same bits — and
    copies them in; after it `advance()` moves torch's own step counters, which live on the host.
    Pinned host buffers are private to this object: a captured copy re-reads them at every replay.
    """

    def __init__(self, optimizer, params, max_norm=1.0):
        params = [p for p in params]
        if _fast_path(optimizer, params) is None:
            raise RuntimeError("CapturedStep needs an optimizer that has taken at least one fused eager step "
                               "with these parameters (clip_and_step)")
        self.optimizer, self.params, self.max_norm = optimizer, params, float(max_norm)
        self.f = _FAST[id(optimizer)]
        self.plan = _PLANS[id(optimizer)]
        dev = self.f.groups[0][1][0].device
        n_t, n_g = len(self.f.static), len(self.f.groups)
        self.rows_host = torch.empty((n_t, 5), dtype=torch.int64).pin_memory()
        self.descs = torch.empty((n_t, 5), dtype=torch.int64, device=dev)
        # the host may run several steps ahead of the GPU: a small ring of pinned slots, each reused
        # only after the copy that read it has completed
        self.dyn_host = [torch.zeros((n_g, 4), dtype=torch.float32).pin_memory() for _ in range(8)]
        self.dyn_done = [None] * 8
        self.slot = 0
        self.dyn = torch.zeros((n_g, 4), dtype=torch.float32, device=dev)
        self.norm = None

    def prepare(self):
        """Host side of the next step: constants for step count + 1 and the current learning rates."""
        step = int(self.f.step) + 1
        i = self.slot
        self.slot = (i + 1) % len(self.dyn_host)
        if self.dyn_done[i] is not None:
            self.dyn_done[i].synchronize()
        host = self.dyn_host[i]
        for k, (g, _) in enumerate(self.f.groups):
            if not _group_ok(g):
                raise RuntimeError("optimizer options changed after the capture")
            b1, b2 = g["betas"]
            query("mis_adamw_step_constants", float(g["lr"]), float(b1), float(b2), float(g["weight_decay"]), step,
                  host[k].data_ptr())
        self.dyn.copy_(host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.dyn_done[i] = ev

    def launch(self):
        """The two launches (and the upload of the pointer rows) on the current, capturing stream."""
        f, plan = self.f, self.plan
        dev = self.descs.device
        rows = [(sp, p.grad.data_ptr(), sm, sv, n)
                for p, (sp, sm, sv, n) in zip((p for _, ps in f.groups for p in ps), f.static)]
        self.rows_host.copy_(torch.tensor(rows, dtype=torch.int64))
        self.descs.copy_(self.rows_host, non_blocking=True)
        st = F_._stream()
        ws = F_.workspace(plan.ws_bytes, dev)
        self.norm = torch.empty((), dtype=torch.float32, device=dev)
        n_t = len(rows)
        call("mis_grad_sqsum", self.descs.data_ptr(), plan.prefix.data_ptr(), n_t, plan.n_chunks, ws.data_ptr(), st)
        for k, ((g, ps), (c0, c1)) in enumerate(zip(f.groups, plan.group_chunks)):
            b1, b2 = g["betas"]
            call("mis_clip_adamw_step_dyn", self.descs.data_ptr(), plan.prefix.data_ptr(), n_t, c0, c1, plan.n_chunks,
                 ws.data_ptr(), self.max_norm, float(b1), float(b2), float(g["eps"]), self.dyn[k].data_ptr(), 1,
                 self.norm.data_ptr() if c0 == 0 else None, st)
        return self.norm

    def advance(self):
        """Host bookkeeping after a replay: torch's step counters, tensor versions."""
        f = self.f
        torch._foreach_add_(f.step_tensors, 1)
        f.step = float(int(f.step) + 1)
        touched = []
        for _, ps in f.groups:
            touched += ps
        torch._C._increment_version(touched)
