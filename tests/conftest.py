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
