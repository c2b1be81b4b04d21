"""Data-parallel gradient averaging over RCCL (one process per GPU).

The reference is single-GPU (train.py:75-76 pins CUDA_VISIBLE_DEVICES=0); this is the new
exchange step the sharded batch needs: the 74 parameter gradients (2.9 M float32 = 11.6 MB)
are flattened into a few buckets in reverse layer order (out_conv, decoder4 ... encoder1, the
order backward produces them; the last-produced small tensors form a bucket of their own) and each
bucket is all-reduced (average) as soon as its last
gradient has been accumulated, on the process group's own stream, overlapping the rest of
the backward pass.  Per-peer messages are bucket/world bytes over the xGMI mesh, i.e.
latency-bound, hence few large buckets.

BatchNorm statistics stay rank-local (plain data-parallel semantics): the parity target is
"N reference replicas on N batch shards with averaged gradients".
"""
import os

import torch

from . import functional as F_
import torch.distributed as dist


class GradSync:
    def __init__(self, model, n_buckets=3, process_group=None, tail_numel=65536, always_reduce=False):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # always_reduce: run the collectives even in a group of one (exercises the whole path on one GPU)
        self.active = self.world > 1 or (always_reduce and dist.is_initialized())
        # RCCL averages in the collective; other backends (gloo) only sum
        self.avg_in_collective = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        params = [p for p in model.parameters() if p.requires_grad]
        order = list(reversed(params))                       # backward order
        total = sum(p.numel() for p in order)
        target = (total + n_buckets - 1) // n_buckets
        self.buckets = []
        cur, cur_n = [], 0
        for p in order:
            cur.append(p)
            cur_n += p.numel()
            if cur_n >= target:
                self.buckets.append(cur)
                cur, cur_n = [], 0
        if cur:
            self.buckets.append(cur)
        # The last bucket's exchange cannot overlap anything: it starts when the first layer's gradient
        # is done, i.e. when backward ends.  The parameters produced last (the full-resolution encoder
        # levels, a few thousand values whose backward takes a fifth of the step) go alone, so what is
        # exposed is one latency-bound message instead of the whole last third of the model.
        last = self.buckets[-1]
        tail, n_tail = [], 0
        for p in reversed(last):
            if n_tail + p.numel() > tail_numel:
                break
            tail.insert(0, p)
            n_tail += p.numel()
        if tail and len(tail) < len(last):
            self.buckets[-1] = last[:len(last) - len(tail)]
            self.buckets.append(tail)
        self.flat = [torch.zeros(sum(p.numel() for p in b), dtype=torch.float32, device=b[0].device)
                     for b in self.buckets]
        # one persistent view per parameter into its bucket, shaped like the parameter: the copy target of a
        # finished bucket when gradients were accumulated in place and, after the exchange, the parameter's .grad
        # (the same objects every step: creating 74 views per step cost the host 0.8 ms)
        self.views = [[v.view_as(p) for p, v in zip(b, flat.split([p.numel() for p in b]))]
                      for b, flat in zip(self.buckets, self.flat)]
        self.bucket_of = {}
        for bi, b in enumerate(self.buckets):
            for p in b:
                self.bucket_of[p] = bi
                p._mis_bucket = bi
        self.ready = [0] * len(self.buckets)
        self.handles = [None] * len(self.buckets)
        self.host = [None] * len(self.buckets)
        # (one-rank RCCL group on one MI355X, 8 hardware queues: 4.60 -> 4.49 ms per step, against 4.45 without any
        # exchange; a test may set this False to keep the bucket work on the stream of the pass)
        self.launch_on_second_stream = True
        self._held, self._side_used = [], False      # see _launch(): what the second stream still reads
        for p in params:
            # our hook enqueues the pending weight-gradient reductions before it reads a gradient
            p._mis_flushing_hooks = not getattr(p, "_post_accumulate_grad_hooks", None)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in params]
        model._mis_grad_sync = self
        self.model = model

    def broadcast_parameters(self, src=0):
        """Start every replica from rank `src`'s parameters and buffers: one collective per dtype
        over a flattened copy (not one per tensor).  On RCCL the flat buffer stays on the GPU; other
        backends (gloo: rehearsal and tests) move it through host memory — gloo's broadcast of
        device tensors does not complete on this platform, its all-reduce does."""
        if self.world == 1:
            return
        by_dtype = {}
        for t in list(self.model.parameters()) + list(self.model.buffers()):
            by_dtype.setdefault(t.dtype, []).append(t)
        with torch.no_grad():
            for tensors in by_dtype.values():
                flat = torch.cat([t.detach().reshape(-1) for t in tensors])
                if self.avg_in_collective:
                    dist.broadcast(flat, src=src, group=self.group)
                else:
                    host = flat.cpu()
                    dist.broadcast(host, src=src, group=self.group)
                    flat = host.to(flat.device)
                # written through detach() (shares the version counter; `.data` has a fresh one): the
                # in-place copy bumps t._version, which invalidates operand images packed from old values
                for t, v in zip(tensors, flat.split([t.numel() for t in tensors])):
                    t.detach().copy_(v.view_as(t))

    def begin_step(self):
        self.ready = [0] * len(self.buckets)
        self.handles = [None] * len(self.buckets)
        self._held, self._side_used = [], False

    def _launch(self, bi):
        # With the weight-gradient launches on the device's second stream, the bucket is filled and its collective
        # started from that stream too: it waits for the stream of the pass (the other gradients of the bucket were
        # written there), and the stream of the pass goes on with backward and meets the result in finish_step().
        flat = self.flat[bi]
        # (only when slab launches of this pass actually went to the second stream of THIS device: if every layer took
        # another path, hopping streams would buy a cross-stream wait and nothing else)
        pending_here = [k for k, v in F_._wgrad_pending.items()
                        if k[0] == flat.device.index and any(len(keep) > 4 for _f, keep in v)] if flat.is_cuda else []
        assert all(k[0] == flat.device.index for k in F_._wgrad_pending) or not flat.is_cuda, \
            "pending weight-gradient reductions of another device: one device per process is the supported layout"
        if (self.launch_on_second_stream and pending_here and F_.WGRAD_SIDE_STREAM and F_.DP_SIDE_STREAM
                and F_.PROFILE is None and not torch.cuda.is_current_stream_capturing()):
            side = F_._side_stream(flat.device.index)
            side.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(side):
                self._launch_on_current(bi, True)
            self._side_used = True
        else:
            self._launch_on_current(bi, False)

    def _launch_on_current(self, bi, on_side):
        b, flat = self.buckets[bi], self.flat[bi]
        # weight gradients of this pass so far: reductions enqueued
        self._held.extend(F_.flush_wgrad_reduces(on_second_stream=on_side))
        views = self.views[bi]
        grads = [p.grad for p in b]
        if not any(g is v for v, g in zip(views, grads)):
            # one batched copy launch for the whole bucket (measured against torch._foreach_copy_ into the views:
            # 4.682 vs 4.699 ms per step with four buckets)
            torch.cat([g.reshape(-1) for g in grads], out=flat)
        else:
            # gradients that were accumulated into last step's bucket views (a zero_grad that keeps the tensors)
            # are in place already
            todo = [(v, g) for v, g in zip(views, grads) if g is not v]
            if todo:
                torch._foreach_copy_([v for v, _ in todo], [g for _, g in todo])
        if self.active and self.avg_in_collective:
            self.handles[bi] = dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        elif self.active:
            # other backends (gloo: tests and rehearsals) sum a host copy — gloo's asynchronous collectives
            # on device tensors do not complete on this platform — and have no AVG
            host = flat.cpu()
            self.host[bi] = host
            self.handles[bi] = dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _on_grad(self, p):
        bi = p._mis_bucket
        self.ready[bi] += 1
        if self.ready[bi] == len(self.buckets[bi]):
            self._launch(bi)

    def finish_step(self):
        """Wait for the buckets; afterwards every p.grad holds the averaged gradient."""
        for bi, b in enumerate(self.buckets):
            if self.ready[bi] != len(b):          # a parameter received no gradient this step
                for p in b:
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)
                self._launch(bi)
        if self._side_used:
            # the buckets were filled on the second stream: the optimizer's stream meets it here, and only then
            # may the memory that stream read (slabs, saved activations, unreduced gradients) be handed out again
            dev = self.flat[0].device
            torch.cuda.current_stream(dev).wait_stream(F_._side_stream(dev.index))
        self._held, self._side_used = [], False
        for bi, b in enumerate(self.buckets):
            if self.handles[bi] is not None:
                self.handles[bi].wait()
            flat = self.flat[bi]
            if self.active and not self.avg_in_collective:
                flat.copy_(self.host[bi].div_(self.world))
            if self.active:
                # the averaged values stay where they are: each p.grad becomes a view of the bucket (no
                # copy back; the optimizer's pointer table then repeats from step to step)
                for p, v in zip(b, self.views[bi]):
                    p.grad = v

    def remove(self):
        for h in self._hooks:
            h.remove()
        if getattr(self.model, "_mis_grad_sync", None) is self:
            del self.model._mis_grad_sync


def all_reduce_dice_sums(sums, group=None):
    """Global-batch Dice needs the pooled sums of every rank before the ratio is formed
    (utils/loss_fn.py:79-80 pools over the whole batch): 3*C floats, summed in place."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    return sums
