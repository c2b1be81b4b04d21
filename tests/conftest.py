import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    class G:
        def __init__(self):
            self._c = {}

        def __call__(self, group):
            if group not in self._c:
                path = os.path.join(ROOT, "tests", "golden", group + ".npz")
                self._c[group] = dict(np.load(path, allow_pickle=False))
            return self._c[group]

    return G()


@pytest.fixture(autouse=True)
def _isolated_test_state():
    """Every test starts from the same torch generator state and leaves the package's
    module-level switches as it found them, so neither inputs nor code paths depend on which
    tests ran before (or on a test that failed between flipping a switch and restoring it)."""
    import torch
    from medical_image_segmentation_amd import functional as F_
    names = ("FUSE_BN_SUMS", "BN_SUMS_MIN_NUMEL", "DEFER_WGRAD_REDUCE", "FUSE_HEAD_DZ", "WGRAD_SIDE_STREAM", "FUSE_FAN_OUT", "FUSE_AMSFN_BWD", "FUSE_AMSFN_DZ4", "CENTER_TAP_CONVS", "BATCH_WEIGHT_PREP", "DEFER_COMPOSED", "FUSE_DGRAD_BN_SUMS", "PROFILE", "DROPOUT_RNG", "call", "query")
    saved = {n: getattr(F_, n) for n in names if hasattr(F_, n)}
    torch.manual_seed(0)
    try:
        yield
    finally:
        for n, v in saved.items():
            setattr(F_, n, v)


@pytest.fixture
def report(request):
    """report(label, value): prints a measured error (shown with -rA / on failure) and appends it
    to gpurun_out/parity_report.txt, so the margins behind the asserted bounds can be read."""
    def _report(label, value):
        line = f"{request.node.name}: {label} = {value:.3e}"
        print(line)
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "parity_report.txt"), "a") as f:
                f.write(line + "\n")
        except OSError:
            pass
    return _report
