"""Build recipe for libenerf_hip.so: hipcc, gfx950 only, in-tree output (enerf_amd/lib/).

No hipify, no torch cpp_extension: the device sources are hand-written HIP and the library has a plain C ABI
(include/enerf_hip.h).  Every .hip file is compiled to an object (in parallel, rebuilt only when it or a header
changed) and the objects are linked into one shared library.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIBDIR = os.path.join(_HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libenerf_hip.so")
SOURCES = ["runtime.hip", "raymarching.hip", "gridencoder.hip", "shencoder.hip", "ffmlp.hip", "ffmlp_wgrad.hip",
           "mlp32.hip", "mlp32s.hip", "mlp32s_f16.hip", "nerf_mlp.hip", "nerf_mlp_bwd.hip", "optim.hip", "density_update.hip", "ffnerf.hip", "event_pairs.hip", "train_step.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics", "-Wall",
         "-Wno-unused-function"]
# Per-file extras.  ffmlp.hip (forward + dgrad) keeps its MFMA accumulators in arch VGPRs: every accumulator is
# post-processed by VALU code (activation, 16-bit conversion) right away, and the default AGPR form costs a
# v_accvgpr_read/write pair per element.  The weight-gradient kernels hold up to 192 accumulator registers and need
# the AGPR half of the register file, so they live in their own translation unit without the flag.
# (ENERF_MFMA_VGPR_FORM arms csrc/mfma_guard.h: that form lets a zero-initialised MFMA's result land on its operands)
_VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form", "-DENERF_MFMA_VGPR_FORM"]
EXTRA = {"ffmlp.hip": _VGPR_FORM, "mlp32s.hip": _VGPR_FORM, "mlp32s_f16.hip": _VGPR_FORM, "nerf_mlp.hip": _VGPR_FORM,
         "nerf_mlp_bwd.hip": _VGPR_FORM}
# development aid: extra -D flags for every file (e.g. ENERF_DEFINES="-DENERF_BIN_TIMING" python -m enerf_amd.build --force)
FLAGS += os.environ.get("ENERF_DEFINES", "").split()
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "ffmlp_common.h"), os.path.join(CSRC, "mlp32_common.h"), os.path.join(CSRC, "mlp32s_ops.h"), os.path.join(CSRC, "mfma_guard.h"),
           os.path.join(_HERE, "..", "include", "enerf_hip.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return "hipcc"


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


def _stale(obj, src):
    t = _mtime(obj)
    extra = [os.path.join(CSRC, "mlp32s.hip")] if src.endswith("mlp32s_f16.hip") else []     # (it #includes that file)
    if src.endswith("nerf_mlp_bwd.hip"):
        extra = [os.path.join(CSRC, "nerf_mlp.hip")]
    return t == 0.0 or _mtime(src) > t or any(_mtime(h) > t for h in HEADERS + extra) or _mtime(__file__) > t


def needs_build():
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objs = [os.path.join(OBJDIR, s.replace(".hip", ".o")) for s in srcs]
    return not os.path.exists(LIB) or any(_stale(o, os.path.join(CSRC, s)) for o, s in zip(objs, srcs)) \
        or any(_mtime(o) > _mtime(LIB) for o in objs)


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 and link enerf_amd/lib/libenerf_hip.so (cross-compiles without a GPU)."""
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJDIR, s.replace(".hip", ".o"))
        if force or _stale(obj, src):
            jobs.append([hipcc] + FLAGS + EXTRA.get(s, []) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print("[enerf_amd.build]", " ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(OBJDIR, s.replace(".hip", ".o")) for s in srcs]
    if force or jobs or not os.path.exists(LIB) or any(_mtime(o) > _mtime(LIB) for o in objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", LIB + ".tmp"])
        os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
