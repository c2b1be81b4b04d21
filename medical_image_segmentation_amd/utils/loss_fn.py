"""diceloss and IOULoss — drop-ins for the reference's utils/loss_fn.py:55-88 and 90-124, plus the per-pixel
cross-entropy map of utils/loss_fn.py:199-203, both from ONE fused pass over the logits.
The same pass also counts the confusion matrix; it is attached to the logits tensor so a
following Metrics.update(logits, targets) does not read the logits again."""
from collections import OrderedDict

import torch

from .. import functional as F_


def _attach_counts(logits, targets, cm, status):
    try:
        logits._mis_counts = (targets, cm, status)
    except Exception:      # some tensor subclasses refuse attributes; Metrics then recounts
        pass


class diceloss():
    """`global_batch` (not in the reference, which is single-GPU): in a data-parallel run, pool the
    Dice sums over the batches of ALL ranks before the ratio, as utils/loss_fn.py:79-80 pools over
    its whole batch (True = the default process group, or pass a group).  The default, False,
    is the per-rank loss of "N reference replicas with averaged gradients"."""

    def __init__(self, class_names, smooth=1e-5, global_batch=False):
        self.smooth = smooth
        self.class_names = class_names
        self.global_batch = global_batch

    def __call__(self, logits, targets):
        """logits [B, C, H, W] (any float dtype/layout on the GPU), targets [B, H, W] int.
        Returns {class_names[i]: loss_i ..., 'total_loss': sum_i loss_i / C} — 'total_loss' last."""
        losses, _, cm, status = F_.DiceLossFn.apply(logits, targets, float(self.smooth), False, F_.LOSS_DICE,
                                                    self.global_batch or None)
        _attach_counts(logits, targets, cm, status)
        loss_dict = OrderedDict((name, losses[i]) for i, name in enumerate(self.class_names))
        loss_dict["total_loss"] = losses[logits.shape[1] - 1]
        return loss_dict


class IOULoss():
    """Drop-in for the reference's utils/loss_fn.py:90-124: same pass, same three pooled sums,
    loss_i = 1 - I/(P + T - I + smooth)."""

    def __init__(self, class_names, smooth=1e-5, global_batch=False):
        self.smooth = smooth
        self.class_names = class_names
        self.global_batch = global_batch

    def __call__(self, logits, targets):
        losses, _, cm, status = F_.DiceLossFn.apply(logits, targets, float(self.smooth), False, F_.LOSS_IOU,
                                                    self.global_batch or None)
        _attach_counts(logits, targets, cm, status)
        loss_dict = OrderedDict((name, losses[i]) for i, name in enumerate(self.class_names))
        loss_dict["total_loss"] = losses[logits.shape[1] - 1]
        return loss_dict


class DiceCELoss():
    """Dice (as above) + weight * mean cross-entropy map (label_smoothing=1e-7,
    ignore_index=0, the expression at utils/loss_fn.py:199-203), fused in one kernel."""

    def __init__(self, class_names, smooth=1e-5, ce_weight=1.0):
        self.smooth = smooth
        self.class_names = class_names
        self.ce_weight = ce_weight

    def __call__(self, logits, targets):
        losses, ce, cm, status = F_.DiceLossFn.apply(logits, targets, float(self.smooth), True)
        _attach_counts(logits, targets, cm, status)
        loss_dict = OrderedDict((name, losses[i]) for i, name in enumerate(self.class_names))
        loss_dict["ce_loss"] = ce.mean()
        loss_dict["total_loss"] = losses[logits.shape[1] - 1] + self.ce_weight * loss_dict["ce_loss"]
        return loss_dict


def ce_map(logits, targets):
    """nn.CrossEntropyLoss(label_smoothing=1e-7, reduction='none', ignore_index=0)(logits, targets)."""
    _, ce, _, _ = F_.DiceLossFn.apply(logits, targets, 1e-5, True)
    return ce
