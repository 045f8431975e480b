"""python -m enerf_amd.ext.build [--force]: (re)build the four extension modules in place when a source, the shared
header or libenerf_hip.so's C ABI header is newer than the built module."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
NAMES = {"_raymarching": "raymarching_shim.cpp", "_gridencoder": "gridencoder_shim.cpp",
         "_shencoder": "shencoder_shim.cpp", "_ffmlp": "ffmlp_shim.cpp"}


def built(name):
    hits = glob.glob(os.path.join(HERE, name + ".*.so"))
    return hits[0] if hits else None


def needs_build():
    deps = [os.path.join(HERE, "shim_common.h"), os.path.join(ROOT, "include", "enerf_hip.h"),
            os.path.join(HERE, "setup.py")]
    for name, src in NAMES.items():
        so = built(name)
        if so is None:
            return True
        t = os.path.getmtime(so)
        if any(os.path.getmtime(p) > t for p in deps + [os.path.join(HERE, src)]):
            return True
    return False


def build(force=False, verbose=True):
    if not (force or needs_build()):
        return [built(n) for n in NAMES]
    # --force: setuptools only compares the .cpp files' times, not the shared headers' that needs_build() watches
    cmd = [sys.executable, os.path.join(HERE, "setup.py"), "build_ext", "--inplace", "-j", "4", "--force"]
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(cmd, cwd=HERE, stdout=out, stderr=out)
    return [built(n) for n in NAMES]


if __name__ == "__main__":
    print("\n".join(build(force="--force" in sys.argv)))
