"""The C-ABI library loads on a machine without a GPU and exports every symbol include/enerf_hip.h declares
(no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "enerf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(enerf_[a-zA-Z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    from enerf_amd import build, _lib
    build.build(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/enerf_hip.h but not exported"
    # every compute entry point has a ctypes signature in enerf_amd/_lib.py
    bound = set(_lib.SIGNATURES) | {"enerf_last_error", "enerf_workspace_generation"}
    assert set(names) <= bound, sorted(set(names) - bound)
    assert _lib.lib().enerf_abi_version() == 1


def test_backends_expose_reference_function_names():
    from enerf_amd.backends import _raymarching, _gridencoder, _shencoder, _ffmlp
    # raymarching/src/bindings.cpp:5-20, gridencoder/src/bindings.cpp:5-8, shencoder/src/bindings.cpp:5-8,
    # ffmlp/src/bindings.cpp:5-11
    for n in ["packbits", "near_far_from_aabb", "polar_from_ray", "morton3D", "morton3D_invert", "march_rays_train",
              "composite_rays_train_forward", "composite_rays_train_backward", "march_rays", "composite_rays",
              "compact_rays"]:
        assert callable(getattr(_raymarching, n))
    for n in ["grid_encode_forward", "grid_encode_backward"]:
        assert callable(getattr(_gridencoder, n))
    for n in ["sh_encode_forward", "sh_encode_backward"]:
        assert callable(getattr(_shencoder, n))
    for n in ["ffmlp_forward", "ffmlp_inference", "ffmlp_backward", "allocate_splitk", "free_splitk"]:
        assert callable(getattr(_ffmlp, n))


def test_product_path_refuses_cpu_tensors():
    """No CPU fallback: the HIP backends reject non-CUDA tensors like the reference's CHECK_CUDA."""
    import torch
    from enerf_amd.backends import _gridencoder, _shencoder
    x = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        _shencoder.sh_encode_forward(x, torch.zeros(4, 16), 4, 3, 4, False, torch.zeros(1))
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        _gridencoder.grid_encode_forward(x, torch.zeros(8, 2), torch.zeros(2, dtype=torch.int32), torch.zeros(1, 4, 2),
                                         4, 3, 2, 1, 0.0, 16, False, torch.zeros(1), 0)


def test_dropin_modules_importable_as_top_level():
    import importlib
    import sys
    d = os.path.join(ROOT, "enerf_amd", "dropin")
    sys.path.insert(0, d)
    try:
        for n in ("_raymarching", "_gridencoder", "_shencoder", "_ffmlp"):
            sys.modules.pop(n, None)
            m = importlib.import_module(n)
            assert m.__file__.startswith(d)
    finally:
        sys.path.remove(d)
        for n in ("_raymarching", "_gridencoder", "_shencoder", "_ffmlp"):
            sys.modules.pop(n, None)
