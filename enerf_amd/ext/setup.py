"""Builds the four CPython extension modules the reference's wrappers import -- `_raymarching`, `_gridencoder`,
`_shencoder`, `_ffmlp` (the names of raymarching/setup.py:48, gridencoder/setup.py:36, shencoder/setup.py:36,
ffmlp/setup.py:38) -- as pybind11 shims over libenerf_hip.so:

    python enerf_amd/ext/setup.py build_ext --inplace        (or: python -m enerf_amd.ext.build)

The shims are host C++ only (tensor checks, current stream, status -> exception); every kernel stays in the C-ABI
library, so nothing is hipified and no HIP code is compiled here.  With this directory on sys.path the reference's
untouched raymarching/raymarching.py, gridencoder/grid.py, shencoder/sphere_harmonics.py and ffmlp/ffmlp.py run on the
MI355X kernels.
"""
import os

from setuptools import setup
from torch.utils.cpp_extension import BuildExtension, CppExtension

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")

common = dict(
    include_dirs=[os.path.join(ROOT, "include"), HERE, os.path.join(ROCM, "include")],
    define_macros=[("__HIP_PLATFORM_AMD__", "1"), ("USE_ROCM", "1")],
    library_dirs=[os.path.join(ROOT, "enerf_amd", "lib"), os.path.join(ROCM, "lib")],
    libraries=["enerf_hip", "amdhip64", "c10_hip"],
    runtime_library_dirs=["$ORIGIN/../lib"],
    extra_compile_args=["-O2", "-g0", "-std=c++17", "-Wno-deprecated-declarations"],
    extra_link_args=["-s"],
)

if __name__ == "__main__":
    os.chdir(HERE)
    setup(
        name="enerf_hip_ext",
        ext_modules=[CppExtension(name, [src], **common) for name, src in (
            ("_raymarching", "raymarching_shim.cpp"), ("_gridencoder", "gridencoder_shim.cpp"),
            ("_shencoder", "shencoder_shim.cpp"), ("_ffmlp", "ffmlp_shim.cpp"))],
        cmdclass={"build_ext": BuildExtension.with_options(use_ninja=False)},
    )
