"""Build recipe for libenerf_hip.so: hipcc, gfx950 only, in-tree output (enerf_amd/lib/).

No hipify, no torch cpp_extension: the device sources are hand-written HIP and the library has a plain C ABI
(include/enerf_hip.h), so a single hipcc invocation is the whole build.
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIBDIR = os.path.join(_HERE, "lib")
LIB = os.path.join(LIBDIR, "libenerf_hip.so")
SOURCES = ["runtime.hip", "raymarching.hip", "gridencoder.hip", "shencoder.hip", "ffmlp.hip", "mlp32.hip", "optim.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(_HERE, "..", "include", "enerf_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True):
    """Compile every HIP source into enerf_amd/lib/libenerf_hip.so (cross-compiles without a GPU)."""
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [_hipcc()] + FLAGS + srcs + ["-o", LIB + ".tmp"]
    if verbose:
        print("[enerf_amd.build]", " ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
