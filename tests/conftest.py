import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no GPU is visible, whatever -m says."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def cpu_oracle_backend(monkeypatch):
    """Run enerf_amd's host-side wrappers on CPU tensors against the C oracle (tests only; the product never does)."""
    import torch
    from oracle import backend as ob
    import enerf_amd.raymarching as rm
    import enerf_amd.gridencoder as ge
    import enerf_amd.shencoder as sh
    import enerf_amd.ffmlp as ff
    monkeypatch.setattr(rm, "_backend", ob.raymarching_backend)
    monkeypatch.setattr(rm, "_DEVICE", "cpu")
    monkeypatch.setattr(ge, "_backend", ob.gridencoder_backend)
    monkeypatch.setattr(sh, "_backend", ob.shencoder_backend)
    monkeypatch.setattr(ff, "_backend", ob.ffmlp_backend)
    monkeypatch.setattr(ff.FFMLP, "compute_dtype", torch.float32)
    return ob


@pytest.fixture(params=["split-bf16", "fp32"])
def mlp32_mode(request):
    """Arithmetic of the fused fp32 MLP kernels for tests that compare them with an fp32 statement on the oracle side:
    "fp32" (v_mfma_f32_32x32x2_f32, bit-comparable fmaf chains) keeps the round-off-level bars; "split-bf16" (the
    product's default: three bf16 MFMA products per fp32 product, ~2^-16 per product) is held to the path's 1e-4 on what
    is rendered, and on gradients to bars that allow for the hidden units it leaves on the other side of a ReLU."""
    from enerf_amd import _lib
    mode = 1 if request.param == "split-bf16" else 0
    prev = _lib.lib().enerf_mlp32_precision(mode)
    yield request.param
    _lib.lib().enerf_mlp32_precision(prev)
