"""oracle/_ref: the reference's OWN native kernels, built for gfx950.  TEST INFRASTRUCTURE ONLY.

What this is.  `raymarching/src/raymarching.cu`, `shencoder/src/shencoder.cu` and `gridencoder/src/gridencoder.cu` (with
their `bindings.cpp` / headers / `pcg32.h`) are plain CUDA C++ without warp intrinsics, textures or third-party
libraries.  This image's PyTorch is a ROCm build, and a ROCm PyTorch carries the source translator its extension builder
applies to every `CUDAExtension` (`torch.utils.hipify`): it is the step the reference's own `setup.py` would run on this
machine.  The recipe below does that step by hand and nothing else:

  1. copy the module's `src/` directory from where it lies under /root/reference to a scratch directory under /tmp
     (never into this repository, never written back into /root/reference);
  2. run `torch.utils.hipify.hipify_python.hipify` on the scratch copy (header names and API spellings only: `cuda_fp16.h`
     -> `hip/hip_fp16.h`, `ATen/cuda/CUDAContext.h` -> `ATen/hip/HIPContext.h`, `cudaStream_t` -> `hipStream_t` ...; no
     kernel body is touched by us, no header, library or source file is written by us);
  3. `hipcc --offload-arch=gfx950` on the translated files with PyTorch's own include paths, linked against the libtorch
     the process already has, into `oracle/_ref/_ref_<module>*.so` -- a pybind11 module exposing exactly the reference's
     `bindings.cpp` functions (renamed `_ref_raymarching` ... through TORCH_EXTENSION_NAME so it cannot be mistaken for the
     product's `_raymarching`).  The scratch directory is deleted; only the `.so` files stay, git-ignored, and travel to
     the GPU box with the tree.

`gridencoder` needs ONE MORE API SPELLING than PyTorch's translator knows (RESPELL below; rounds 1-3 left the module
unbuilt over it).  CUDA overloads `atomicAdd` for `__half*` / `__half2*`; ROCm 7.2 gives the same two operations the name
`unsafeAtomicAdd` (hip/amd_detail/amd_hip_fp16.h:882,910) and declares no `atomicAdd` for them, so `gridencoder.cu`
stops at two call sites -- its `at::Half` helper ("never used, just for compatability", gridencoder.cu:24-26) and the
packed-half branch of the backward kernel (gridencoder.cu:301) -- that every instantiation must compile and only an
`at::Half` table ever executes.  Step 2b respells exactly those two calls in the scratch copy, the way step 2 respells
`cudaStream_t`: call name only, arguments untouched, the count of rewritten sites asserted.  What that does and does
not touch, so that a reader can weigh it:
  * `kernel_grid` (the forward -- this repository's roofline kernel), `kernel_input_backward`, the index / hash / scale
    arithmetic and the fp32 branch of `kernel_grid_backward` (`atomicAdd(float*, float)`, which HIP does declare) are
    compiled exactly as hipify leaves them;
  * nothing is added: no header, no function, no macro, no library.  The two operations are ROCm's own.
Tests that rest on this build say so (tests/test_gpu_ref_gridencoder.py, the grid_* arrays of
tests/golden/ref_kernels_gfx950.npz); the builder-written second statements of the grid (fp64 torch grid, linear-field
reproduction) stay in the suite beside them.

`ffmlp` is NOT built (UNBUILDABLE below: the empty CUTLASS submodule and `nvcuda::wmma`).  No stand-ins are written -- it
stays on the second statements DESIGN.md section 2 lists.

Who may use the result: tests (`tests/test_gpu_ref_kernels.py` compares the product's HIP kernels and the C oracle with
these kernels on the same GPU, same inputs), `oracle/mint_ref_gpu.py` (writes small input / output fixtures of the
reference's kernels under tests/golden/ so that the CPU suite pins the C oracle against them).  Nothing under enerf_amd/
imports it.  It needs /root/reference to BUILD (this container) and a GPU to RUN (the box).
"""
import glob
import os
import shutil
import subprocess
import sys
import sysconfig
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REFERENCE = "/root/reference"
MODULES = {
    # module directory under the reference -> (translation units, extension name)
    "raymarching": (["raymarching.cu", "bindings.cpp"], "_ref_raymarching"),
    "shencoder": (["shencoder.cu", "bindings.cpp"], "_ref_shencoder"),
    "gridencoder": (["gridencoder.cu", "bindings.cpp"], "_ref_gridencoder"),
}
# (Tried: the same module with -munsafe-fp-atomics -- no change, 2490 vs 2497 us for the training batch's backward: hipcc
#  already emits global_atomic_add_f32 for `atomicAdd(float*, float)` on gfx950; the kernel is slow because it issues one
#  memory-side atomic per corner and channel, 34 M per step at the ~14-21 G/s those sustain.)
# API spellings PyTorch's hipify does not carry (see the module docstring): translation unit -> ((CUDA spelling, ROCm
# spelling, number of sites that must be found), ...).  Call names only.
RESPELL = {
    ("gridencoder", "gridencoder.cu"): (
        ("atomicAdd(reinterpret_cast<__half*>(", "unsafeAtomicAdd(reinterpret_cast<__half*>(", 1),
        ("atomicAdd((__half2*)", "unsafeAtomicAdd((__half2*)", 1),
    ),
}


# Tried with the same recipe and NOT buildable without writing something the image lacks (so: not built, no stand-ins):
UNBUILDABLE = {
    "ffmlp": "needs the CUTLASS submodule (empty directory, commit not recorded) and nvcuda::wmma (mma.h)",
}


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE, "raymarching", "src"))


def built() -> list:
    return sorted(glob.glob(os.path.join(OUT, "_ref_*.so")))


def _flags():
    import torch
    from torch.utils import cpp_extension as ce
    inc = []
    for p in ce.include_paths("cuda") + [sysconfig.get_paths()["include"]]:
        inc += ["-isystem", p]
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    cxx11 = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    common = ["-O3", "-std=c++17", "-fPIC", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DHIPBLAS_V2",
              f"-D_GLIBCXX_USE_CXX11_ABI={cxx11}", "-DTORCH_API_INCLUDE_EXTENSION_H", "-w",
              # the reference's setup.py: -U__CUDA_NO_HALF_OPERATORS__ etc. (it wants the half operators); the HIP
              # spelling of the same request is simply not to define __HIP_NO_HALF_OPERATORS__
              "-fno-gpu-rdc", "--offload-arch=gfx950"]
    link = ["-shared", f"-L{libdir}", f"-Wl,-rpath,{libdir}", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip",
            "-ltorch", "-ltorch_python"]
    return inc, common, link


def build(verbose: bool = False) -> list:
    """Build every module whose .so is missing.  Returns the list of built libraries ([] when /root/reference is absent
    and nothing was built before -- the GPU box only uses prebuilt files)."""
    if not available():
        return built()
    os.makedirs(OUT, exist_ok=True)
    from torch.utils.hipify import hipify_python
    inc, common, link = _flags()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    for mod, (units, name) in MODULES.items():
        target = os.path.join(OUT, name + suffix)
        src = os.path.join(REFERENCE, mod, "src")
        newest = max(os.path.getmtime(os.path.join(src, f)) for f in os.listdir(src))
        if os.path.exists(target) and os.path.getmtime(target) >= newest:
            continue
        scratch = tempfile.mkdtemp(prefix=f"enerf_ref_{mod}_", dir="/tmp")
        try:
            work = os.path.join(scratch, "src")
            shutil.copytree(src, work)
            res = hipify_python.hipify(project_directory=work, output_directory=work, includes=[os.path.join(work, "*")],
                                       extensions=(".cu", ".cuh", ".h", ".cpp", ".hpp"), show_detailed=False,
                                       show_progress=False, is_pytorch_extension=True, hip_clang_launch=True)
            objs = []
            for u in units:
                hip = res[os.path.join(work, u)].hipified_path or os.path.join(work, u)
                for cuda_name, rocm_name, sites in RESPELL.get((mod, u), ()):             # step 2b
                    text = open(hip).read()
                    if text.count(cuda_name) != sites:
                        raise RuntimeError(f"{mod}/{u}: expected {sites} site(s) of '{cuda_name}', found "
                                           f"{text.count(cuda_name)} -- the reference changed; not guessing")
                    with open(hip, "w") as f:
                        f.write(text.replace(cuda_name, rocm_name))
                obj = os.path.join(scratch, os.path.basename(hip) + ".o")
                cmd = [hipcc, "-x", "hip", "-c", hip, "-o", obj, f"-DTORCH_EXTENSION_NAME={name}", f"-I{work}"] \
                    + inc + common
                if verbose:
                    print("[oracle/_ref]", " ".join(cmd), flush=True)
                subprocess.run(cmd, check=True)
                objs.append(obj)
            tmp = target + ".tmp"
            subprocess.run([hipcc, "--offload-arch=gfx950"] + objs + link + ["-o", tmp], check=True)
            os.replace(tmp, target)
            if verbose:
                print(f"[oracle/_ref] built {target}", flush=True)
        finally:
            shutil.rmtree(scratch, ignore_errors=True)
    return built()


def load(name: str):
    """Import oracle/_ref/_ref_<name>; raises ImportError when it was not built (tests skip on that)."""
    import importlib
    import torch  # noqa: F401  (libtorch must be loaded first)
    if OUT not in sys.path:
        sys.path.insert(0, OUT)
    return importlib.import_module("_ref_" + name)


if __name__ == "__main__":
    print("\n".join(build(verbose=True)))
