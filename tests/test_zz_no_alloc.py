"""The allocation-free form of the fused loss/metrics pass (mis_loss_metrics_fwd_static: scratch in
a fixed array inside the library's device image) against the workspace form, through the C-ABI:
same kernels, so every output must be bit-identical — on two calls in a row, the second on a
smaller input (the scratch keeps the first call's contents)."""
import numpy as np
import pytest
import torch

from oracle import datagen as dg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(static, logits, target, want_cm=True, want_ce=True):
    from medical_image_segmentation_amd import _lib
    from medical_image_segmentation_amd.functional import _stream
    B, C, H, W = logits.shape
    sb, sc, sh, sw = logits.stride()
    sums = torch.full((3, C), float("nan"), device=DEV)
    ce = torch.full((B, H, W), float("nan"), device=DEV) if want_ce else None
    cm = torch.full((B, C, C), -1, dtype=torch.int64, device=DEV) if want_cm else None
    status = torch.full((1,), -1, dtype=torch.int32, device=DEV)
    p = lambda t: t.data_ptr() if t is not None else None
    dt = _lib.BF16 if logits.dtype == torch.bfloat16 else _lib.F32
    args = [logits.data_ptr(), dt, sb, sc, sw, target.data_ptr(), B, C, H * W, p(sums), p(ce), p(cm), p(status)]
    if static:
        _lib.call("mis_loss_metrics_fwd_static", *args, _stream())
    else:
        nbytes = _lib.query("mis_loss_metrics_workspace_bytes", B, C, H * W)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
        _lib.call("mis_loss_metrics_fwd", *args, ws.data_ptr(), _stream())
    torch.cuda.synchronize()
    return sums, ce, cm, status


def _case(B, C, H, W, dtype, seed, layout):
    x = torch.from_numpy(dg.hash_normalish((B, C, H, W), seed)).to(DEV).to(dtype)
    if layout == "nhwc":
        x = x.contiguous(memory_format=torch.channels_last)
    t = torch.from_numpy(dg.blob_mask(B, H, W, C, seed + 1)).to(DEV)
    return x, t


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_static_scratch_form_equals_workspace_form(dtype, layout):
    if not torch.cuda.is_available():
        pytest.fail("GPU suite needs a HIP device")
    # large first (many partial rows), then smaller ones that leave most of the scratch stale;
    # the last without a count output (the counts then live in the scratch too)
    for (B, C, H, W, cm) in [(6, 4, 160, 192, True), (2, 4, 48, 40, True), (1, 9, 16, 24, True),
                             (3, 5, 64, 64, False), (1, 2, 8, 8, False)]:
        x, t = _case(B, C, H, W, dtype, 4000 + B * C, layout)
        got = _run(True, x, t, want_cm=cm)
        ref = _run(False, x, t, want_cm=cm)
        for name, a, b in zip(("sums", "ce_map", "cm", "status"), got, ref):
            if a is None:
                assert b is None
                continue
            assert torch.equal(a, b), f"{name} differs at {(B, C, H, W)}"
        assert int(got[3]) == 0
        # and against the CPU oracle (the workspace form's own parity test covers the rest)
        from oracle import unet_oracle as orc
        if cm:
            assert np.array_equal(got[2].cpu().numpy(), np.asarray(orc.confusion_counts(x.float().cpu(), t.cpu())))


def test_static_scratch_form_rejects_oversized_problem():
    from medical_image_segmentation_amd import _lib
    from medical_image_segmentation_amd.functional import _stream
    B, C, HW = 600, 16, 4096          # workspace of the ordinary form: > 1 MiB
    assert _lib.query("mis_loss_metrics_workspace_bytes", B, C, HW) > (1 << 20)
    one = torch.zeros(1, device=DEV)
    with pytest.raises(RuntimeError, match="scratch"):
        _lib.call("mis_loss_metrics_fwd_static", one.data_ptr(), _lib.F32, C * HW, HW, 1, one.data_ptr(), B, C, HW,
                  None, None, None, one.data_ptr(), _stream())
