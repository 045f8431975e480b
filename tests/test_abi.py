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
    assert _lib.lib().enerf_abi_version() == _lib.header_abi_version() >= 2


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


def test_extension_modules_import_as_top_level_and_match_the_reference_prototypes():
    """The four pybind11 modules (enerf_amd/ext, built by __graft_entry__.build) import under the names the reference's
    wrappers look for, and every function takes the reference's arguments -- kinds and order -- as recorded from the
    reference's own headers (tests/golden/ref_binding_signatures.json: raymarching.h:7-19, gridencoder.h:12-13,
    shencoder.h:9,12, ffmlp.h:8-14).  The ctypes face of the same entry points (enerf_amd/backends) is held to the same
    parameter names."""
    import importlib
    import inspect
    import json
    import sys
    from enerf_amd import ext
    from enerf_amd.ext import build as eb
    eb.build(verbose=False)
    sigs = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_binding_signatures.json")))
    d = ext.activate()
    kind_of = {"Tensor": "torch.Tensor", "int": "SupportsInt", "float": "SupportsFloat", "bool": "bool"}
    try:
        for mod_name, functions in sigs.items():
            sys.modules.pop(mod_name, None)
            m = importlib.import_module(mod_name)
            assert m.__file__.startswith(d)
            back = importlib.import_module("enerf_amd.backends." + mod_name)
            assert sorted(functions) == sorted(n for n in dir(m) if not n.startswith("_"))
            for fname, params in functions.items():
                doc = getattr(m, fname).__doc__.split("->")[0]
                got = [a.split(":")[1].strip() for a in doc[doc.index("(") + 1:doc.rindex(")")].split(",") if ":" in a]
                assert len(got) == len(params), (mod_name, fname, got, params)
                for g, (kind, _) in zip(got, params):
                    assert kind_of[kind] in g, (mod_name, fname, g, kind)
                names = [p.rstrip("_") for p in inspect.signature(getattr(back, fname)).parameters]
                ref_names = [n.rstrip("_") for _, n in params]
                assert names[:len(ref_names)] == ref_names, (mod_name, fname, names, ref_names)
    finally:
        if d in sys.path:
            sys.path.remove(d)
        for n in sigs:
            sys.modules.pop(n, None)


def test_reference_kernel_build_recipe_is_declared():
    """oracle/build_ref.py: which reference modules are built for gfx950 as the GPU tests' cross-check, which are not and
    why (no compute here: the recipe needs /root/reference to build and a GPU to run)."""
    from oracle import build_ref as br
    assert set(br.MODULES) == {"raymarching", "shencoder", "gridencoder"} and set(br.UNBUILDABLE) == {"ffmlp"}
    # the one thing the recipe changes beyond PyTorch's translator: two call NAMES, both in at::Half-only code
    assert {k: [(a, b) for a, b, _ in v] for k, v in br.RESPELL.items()} == {("gridencoder", "gridencoder.cu"): [
        ("atomicAdd(reinterpret_cast<__half*>(", "unsafeAtomicAdd(reinterpret_cast<__half*>("),
        ("atomicAdd((__half2*)", "unsafeAtomicAdd((__half2*)")]}
    assert all(name.startswith("_ref_") for _, name in br.MODULES.values())
    assert isinstance(br.built(), list)


def test_graft_entry_build_runs_end_to_end():
    """The driver's build check: `__graft_entry__.build()` must return (it raised at the end of round 5 on a stale ABI
    constant no test looked at).  Everything is built already when the suite runs, so this is an incremental no-op plus
    the entry point's own assertions; run in a subprocess, as the driver does, from the repo root."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=root,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "[build] ok" in r.stdout
