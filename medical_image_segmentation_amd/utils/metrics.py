"""Metrics — drop-in for the reference's utils/metrics.py:17-152.

The GPU counts a per-sample confusion matrix cm[b][t][p] in int64 (one pass, or reused from
the loss pass over the same logits); every TP/FN/FP/TN of the reference's
compute_confusion_matrix (metrics.py:31-34) is a sum of its entries.  The float post-math is
the reference's own, run on the host on the tiny count tensors with the same torch CPU ops
and dtypes (int64 -> float32 ratios -> Python floats -> float64), quirks included: the
recall/precision one-hots are 9 wide whatever len(class_names) is (metrics.py:45-46,59-60)."""
import numpy as np
import torch
import torch.nn as nn

from .. import functional as F_


class Metrics(nn.Module):
    def __init__(self, class_names: list, smooth=1e-5, global_batch=False):
        """`global_batch` (not in the reference, which is single-GPU): in a data-parallel run the
        per-sample count matrices of all ranks are gathered (B*C*C int64 per rank) before the
        reference's formulas, so every row — the pooled dice/mIoU rows (metrics.py:95-97,113-115)
        and the per-sample means — is the one a single process would compute on the global batch.
        True = the default process group, or pass a group."""
        super().__init__()
        self.class_names = class_names
        self.num_classes = len(class_names)
        self.smooth = smooth
        self.global_batch = global_batch

    # -- GPU part ---------------------------------------------------------------
    def confusion_counts(self, prediction, target):
        """int64 [B, C, C] on the device + the device status word."""
        cached = getattr(prediction, "_mis_counts", None)
        if cached is not None and cached[0] is target:
            return cached[1], cached[2]
        if isinstance(prediction, (list, tuple)):
            prediction = prediction[0]
        _, _, cm, status, _, _ = F_.loss_metrics_raw(prediction.detach(), target, False, False, True)
        return cm, status

    def gather_counts(self, cm, status):
        """Counts of every rank's shard, in rank order (the global batch); identity outside a
        data-parallel run or with global_batch off."""
        import torch.distributed as dist
        if not self.global_batch or not dist.is_initialized():
            return cm, status
        pg = None if self.global_batch is True else self.global_batch
        world = dist.get_world_size(pg)
        if world == 1:
            return cm, status
        packed = torch.cat([cm.reshape(-1), status.to(torch.int64).reshape(-1)])
        out = torch.empty(world * packed.numel(), dtype=torch.int64, device=cm.device)
        dist.all_gather_into_tensor(out, packed, group=pg)
        out = out.view(world, -1)
        return out[:, :-1].reshape((world * cm.shape[0],) + tuple(cm.shape[1:])), out[:, -1].max().reshape(1)

    # -- host part: the reference's formulas on the counts ------------------------
    @staticmethod
    def _terms(cm, K):
        B, C, _ = cm.shape
        if C > K:
            raise RuntimeError("Class values must be smaller than num_classes.")
        d = torch.diagonal(cm, dim1=1, dim2=2)
        tp = torch.zeros((B, K), dtype=torch.int64)
        fn = torch.zeros((B, K), dtype=torch.int64)
        fp = torch.zeros((B, K), dtype=torch.int64)
        tp[:, :C] = d
        fn[:, :C] = cm.sum(dim=2) - d
        fp[:, :C] = cm.sum(dim=1) - d
        tn = cm[0].sum() - tp - fn - fp
        return tp, fn, fp, tn

    def from_counts(self, cm):
        """cm: int64 [B, C, C] CPU tensor -> float64 ndarray (6, n+1), rows
        [recall, precision, dice, f1, mIoU, accuracy] (metrics.py:137-152)."""
        n, s = self.num_classes, self.smooth
        tp, fn, fp, _ = self._terms(cm, 9)
        recall = (tp / (tp + fn + s)).mean(dim=0)
        recalls = [recall[i + 1].item() for i in range(n)] + [recall[1:].mean().item()]
        precision = (tp / (tp + fp + s)).mean(dim=0)
        precisions = [precision[i + 1].item() for i in range(n)] + [precision[1:].mean().item()]
        f1s = [2 * (recalls[i] * precisions[i]) / (recalls[i] + precisions[i] + s) for i in range(n + 1)]
        tp, fn, fp, tn = self._terms(cm, n + 1)
        inter = tp.sum(dim=0).to(torch.float32)
        psum = (tp + fp).sum(dim=0).to(torch.float32)
        tsum = (tp + fn).sum(dim=0).to(torch.float32)
        dice = (2 * inter) / (psum + tsum + s)
        dices = [dice[i + 1].item() for i in range(n)] + [dice[1:].mean().item()]
        iou = inter / (psum + tsum - inter + s)
        mious = [iou[i + 1].item() for i in range(n)] + [iou[1:].mean().item()]
        acc = ((tp + tn) / (tp + tn + fn + fp + s)).mean(dim=0)
        accs = [acc[i + 1].item() for i in range(n)] + [acc[1:].mean().item()]
        metrics = np.stack([recalls, precisions, dices, f1s, mious, accs], axis=0)
        return np.nan_to_num(metrics)

    def update(self, prediction, target):
        cm, status = self.gather_counts(*self.confusion_counts(prediction, target))
        host = torch.cat([cm.reshape(-1), status.to(torch.int64)]).cpu()     # one D2H copy
        if int(host[-1]) != 0:
            raise RuntimeError("Class values must be smaller than num_classes.")
        return self.from_counts(host[:-1].reshape(cm.shape))
