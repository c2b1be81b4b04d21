"""One training step captured in a HIP graph and replayed.

The reference issues a step from Python every time (utils/train_and_eval.py:72-119: forward,
loss, backward, clip_grad_norm_, optimizer.step — ~230 kernel launches here).  On MI355X the
launches of a 16 x 512^2 batch are hidden behind the GPU work, but for small batches or images
the host's issue time (~3 ms per step) is what limits the step.  `GraphedTrainStep` records the
whole step once — every launch goes to the capturing stream through the same C-ABI calls — and
then replays it with one host call per step.

What keeps a replay equal to an eager step:
  * inputs are copied into fixed tensors; every intermediate lives in the graph's private pool;
  * the weight images the convs read are re-packed inside the graph (the capture forces that
    launch), since the optimizer inside the graph changes the weights every replay;
  * the optimizer's step-dependent constants come from device memory (`optim.CapturedStep`),
    formed on the host before each replay with the arithmetic of the eager path;
  * torch's optimizer state (`exp_avg`, `exp_avg_sq`, `step`) is updated as in eager mode, so
    checkpoints and a later return to eager steps are unaffected.
Single-process only: a gradient exchange inside the captured region is not supported.
"""
import torch

from .. import functional as F_
from .optim import CapturedStep, clip_and_step


class GraphedTrainStep:
    """step = GraphedTrainStep(model, optimizer, loss_of, images, masks); out = step(images, masks)

    `loss_of(model, images, masks) -> (loss, aux)`: `loss` is the scalar to differentiate, `aux`
    any tensors (or nested tuples/dicts of tensors) the caller wants back each step; they are
    returned as the graph's own output tensors, overwritten by the next replay.
    The constructor runs `warmup` real training steps on the given batch (lazy initialisation of
    optimizer state, weight images, workspaces happens outside the capture) and then captures;
    capturing itself does not execute anything.  `steps_taken` counts the warm-up steps.
    """

    def __init__(self, model, optimizer, loss_of, images, masks, max_norm=1.0, warmup=2):
        if torch.distributed.is_available() and torch.distributed.is_initialized() \
                and torch.distributed.get_world_size() > 1:
            raise RuntimeError("GraphedTrainStep is single-process: the gradient exchange is not captured")
        if warmup < 1:
            raise ValueError("at least one eager step is needed before the capture")
        self.model, self.optimizer = model, optimizer
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.images = images.clone()
        self.masks = masks.clone()
        side = torch.cuda.Stream(device=images.device)
        side.wait_stream(torch.cuda.current_stream(images.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                optimizer.zero_grad(set_to_none=True)
                loss, _ = loss_of(model, self.images, self.masks)
                loss.backward()
                clip_and_step(optimizer, self.params, max_norm)
        torch.cuda.current_stream(images.device).wait_stream(side)
        self.steps_taken = warmup
        self.captured = CapturedStep(optimizer, self.params, max_norm)
        optimizer.zero_grad(set_to_none=True)
        torch._C._increment_version(self.params)     # the capture must contain the weight re-packing
        torch.cuda.synchronize(images.device)
        self.graph = torch.cuda.CUDAGraph()
        # scratch buffers of the captured kernels: allocated in the graph's pool, owned by this object
        with torch.cuda.graph(self.graph), F_.private_workspaces() as self._workspaces:
            loss, aux = loss_of(model, self.images, self.masks)
            loss.backward()
            norm = self.captured.launch()
        self.out = (loss.detach(), aux, norm)

    def __call__(self, images, masks):
        """One training step on (images, masks) — same shapes and dtypes as at construction.
        Returns (loss, aux, total_grad_norm): tensors owned by the graph."""
        if images.shape != self.images.shape or masks.shape != self.masks.shape:
            raise ValueError(f"captured for images {tuple(self.images.shape)} / masks {tuple(self.masks.shape)}, "
                             f"got {tuple(images.shape)} / {tuple(masks.shape)}")
        if images is not self.images:
            self.images.copy_(images, non_blocking=True)
        if masks is not self.masks:
            self.masks.copy_(masks, non_blocking=True)
        self.captured.prepare()
        self.graph.replay()
        self.captured.advance()
        self.steps_taken += 1
        return self.out
