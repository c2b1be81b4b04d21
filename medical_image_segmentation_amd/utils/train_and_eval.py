"""train_one_epoch / evaluate — drop-ins for the reference's utils/train_and_eval.py:44-124
and :242-296: same arguments, same batch format (data[0][0]=images, data[0][1]=masks,
data[1]=names), same return values (per-class loss list with the total last, and the
(6, n+1) metric matrix averaged over batches).

Differences that do not change results:
  * autocast runs in bfloat16 (no loss scaling needed); a GradScaler passed by the caller is
    honoured if it is enabled, exactly as the reference calls it (train_and_eval.py:111-119);
  * the confusion counts of every batch stay on the GPU and the reference's float post-math
    runs once per epoch on the host, so the step loop has no device->host synchronisation
    (the reference synchronises 7*(n+1) times per batch through .item());
  * per-batch losses are accumulated detached (the reference keeps graph-attached tensors);
  * a target value outside [0, n_classes) is reported when the epoch's counts are read back (the
    kernels set a device status word), not on the offending batch as F.one_hot does;
  * one logits tensor per model call: list / (main, aux) outputs raise NotImplementedError;
  * with a GradSync (data parallel) attached to the model, gradients are averaged across
    ranks before unscale/clip, overlapped with the backward pass;
  * gradient clipping and the AdamW update run as two fused launches on the optimizer's own
    state (utils/optim.py); other optimizers, or an enabled GradScaler, use torch's calls.
"""
import numpy as np
import torch
from torch.amp import autocast

try:
    from tqdm import tqdm
except ImportError:          # progress bar only
    def tqdm(it, **kw):
        return it


from .optim import clip_and_step


def _grad_sync(model):
    return getattr(model, "_mis_grad_sync", None)


def _finish_metrics(Metric, pending, n_rows):
    """pending: list of (cm, status) device tensors, one per batch -> summed metric matrix."""
    total = np.zeros((6, n_rows))
    if not pending:
        return total
    pending = [Metric.gather_counts(cm, st) for cm, st in pending]      # the global batch in a data-parallel run
    flat = torch.cat([torch.cat([cm.reshape(-1), st.to(torch.int64)]) for cm, st in pending]).cpu()
    off = 0
    for cm, _ in pending:
        n = cm.numel()
        if int(flat[off + n]) != 0:
            raise RuntimeError("Class values must be smaller than num_classes.")
        total += Metric.from_counts(flat[off:off + n].reshape(cm.shape))
        off += n + 1
    return total


def _single_output(pred):
    """The reference's loops also accept models that return a list (U2Net deep supervision) or a
    (main, aux) pair mixed 0.6/0.4 (train_and_eval.py:75-103, 266-286); the fused loss kernel takes
    one logits tensor, and none of the models built here returns anything else."""
    if isinstance(pred, (list, tuple)):
        raise NotImplementedError("models returning several outputs (deep supervision lists, (main, aux) pairs) "
                                  "are not supported by the fused loss path: pass a model that returns one "
                                  "[B, C, H, W] logits tensor")
    return pred


def _device_batches(loader, device):
    """Yields (images, masks) of each `data` of the loader (batch format data[0][0], data[0][1],
    train_and_eval.py:72-73) already on `device`.  On a GPU the copy of batch i+1 is issued on a
    side stream right after the work of batch i has been enqueued, so it overlaps that work (fully
    when the loader pins its memory); the compute stream waits on the copy's event only."""
    dev = torch.device(device)
    if dev.type != "cuda":
        for data in loader:
            yield data[0][0].to(dev), data[0][1].to(dev)
        return
    side = torch.cuda.Stream(dev)

    def stage(data):
        with torch.cuda.stream(side):
            im = data[0][0].to(dev, non_blocking=True)
            mk = data[0][1].to(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(side)
        return im, mk, ev

    it = iter(loader)
    try:
        nxt = stage(next(it))
    except StopIteration:
        return
    while nxt is not None:
        im, mk, ev = nxt
        cur = torch.cuda.current_stream(dev)
        cur.wait_event(ev)
        im.record_stream(cur)
        mk.record_stream(cur)
        yield im, mk
        try:
            nxt = stage(next(it))          # runs after the consumer enqueued this batch's work
        except StopIteration:
            nxt = None


def train_one_epoch(model, optimizer, epoch, train_dataloader, device, loss_fn, scaler, Metric, scheduler,
                    class_names, elnloss, l1_lambda, l2_lambda):
    model.train()
    n_batches = len(train_dataloader)
    loader = tqdm(train_dataloader, desc=f" Training on Epoch :{epoch + 1}", leave=False)
    epoch_losses = None
    pending = []
    sync = _grad_sync(model)
    use_scaler = scaler is not None and scaler.is_enabled()
    params = [p for p in model.parameters() if p.requires_grad]

    for images, masks in _device_batches(loader, device):
        optimizer.zero_grad()
        with autocast(device_type="cuda", dtype=torch.bfloat16):
            pred = _single_output(model(images))
            masks = masks.to(torch.int64)
            loss_dict = loss_fn(pred, masks)
            losses = list(loss_dict.values())
            total_loss = loss_dict["total_loss"]
            if elnloss:
                total_loss = total_loss + model.elastic_net(l1_lambda=l1_lambda, l2_lambda=l2_lambda)
            pending.append(Metric.confusion_counts(pred, masks))
        if sync is not None:
            sync.begin_step()
        if use_scaler:
            scaler.scale(total_loss).backward()
        else:
            total_loss.backward()
        if sync is not None:
            sync.finish_step()
        if use_scaler:
            scaler.unscale_(optimizer)
            torch.nn.utils.clip_grad_norm_(params, max_norm=1.0)
            scaler.step(optimizer)
            scaler.update()
        else:
            clip_and_step(optimizer, params, max_norm=1.0)     # clip_grad_norm_ + optimizer.step(), two launches
        scheduler.step()
        vals = torch.stack([l.detach().float() for l in losses])
        epoch_losses = vals if epoch_losses is None else epoch_losses + vals

    Metric_list = _finish_metrics(Metric, pending, len(class_names) + 1) / max(n_batches, 1)
    if epoch_losses is None:
        return [0.0] * (len(class_names) + 1), Metric_list
    return list(epoch_losses.unbind(0)), Metric_list


def evaluate(model, device, data_loader, loss_fn, Metric, class_names, test: bool = False):
    model.eval()
    n_batches = len(data_loader)
    epoch_losses = None
    pending = []
    with torch.no_grad():
        loader = tqdm(data_loader, desc="  Validating ", leave=False)
        for images, masks in _device_batches(loader, device):
            with autocast(device_type="cuda", dtype=torch.bfloat16):
                pred_mask = _single_output(model(images))
                masks = masks.to(torch.int64)
                masks = masks.squeeze(1)
                loss_dict = loss_fn(pred_mask, masks)
                losses = list(loss_dict.values())
                pending.append(Metric.confusion_counts(pred_mask, masks))
            vals = torch.stack([l.detach().float() for l in losses])
            epoch_losses = vals if epoch_losses is None else epoch_losses + vals
    Metric_list = _finish_metrics(Metric, pending, len(class_names) + 1) / max(n_batches, 1)
    if epoch_losses is None:
        return [0.0] * (len(class_names) + 1), Metric_list
    return list(epoch_losses.unbind(0)), Metric_list
