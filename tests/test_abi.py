"""CPU suite: the C-ABI library loads here (no GPU needed to dlopen it) and exports every
symbol include/misamd.h declares; the ctypes prototype table covers the same set."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "misamd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mis_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from medical_image_segmentation_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() / make -C medical_image_segmentation_amd/csrc"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in misamd.h but not exported"
    assert set(_lib.PROTOTYPES) == set(syms)


def test_probe_calls_without_gpu():
    from medical_image_segmentation_amd import _lib
    lib = _lib.load()
    assert lib.mis_version() >= 100
    assert lib.mis_arch() == b"gfx950"
    # pure host-side size queries
    assert lib.mis_loss_metrics_workspace_bytes(16, 4, 512 * 512) > 0
    assert lib.mis_conv3x3_workspace_bytes(16, 512, 512, 8, 32) > 0
    assert lib.mis_loss_metrics_workspace_bytes(0, 4, 16) == 0


def test_argument_errors_are_reported_not_crashed():
    from medical_image_segmentation_amd import _lib
    lib = _lib.load()
    # null pointers / bad class count are rejected on the host before any launch
    rc = lib.mis_dice_from_sums(None, 4, 1e-5, 0, None, None)
    assert rc == 1 and b"dice_from_sums" in lib.mis_last_error()
    rc = lib.mis_maxpool2x2_fwd(ctypes.c_void_p(16), None, None, 0, 1, 3, 4, 8, ctypes.c_void_p(16), None)
    assert rc == 1 and b"even" in lib.mis_last_error()


def test_product_fails_loudly_without_gpu():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from medical_image_segmentation_amd.utils.metrics import Metrics
    with pytest.raises(RuntimeError):
        Metrics(["a", "b", "c"]).update(torch.zeros(1, 4, 16, 16), torch.zeros(1, 16, 16, dtype=torch.int64))
