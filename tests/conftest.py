import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when no device is visible, so `-m "not gpu"` and a bare
    # `pytest tests` both work in the GPU-less build container.
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(params=["f32", "bf16x3"])
def parity_mode(request):
    """The 1e-3 parity gates run twice: in the exact fp32-input MFMA mode (generic tile GEMMs, unfused attention) and in
    the split-operand bf16x3 mode, which runs the token-on-lane Linear, fused attention and weight-gradient kernels on hi / lo
    bf16 halves of the fp32 operands (GG_PREC_BF16X3).  gpu_util.engine_from_cfg builds its engines in the mode set here."""
    import gpu_util
    gpu_util.PARITY_PRECISION = request.param
    yield request.param
    gpu_util.PARITY_PRECISION = "f32"
