"""pybind11 extension modules over libenerf_hip.so, built in place by enerf_amd/ext/setup.py (see its docstring).
`import enerf_amd.ext` puts this directory on sys.path so that `_raymarching`, `_gridencoder`, `_shencoder`, `_ffmlp`
import as the top-level modules the reference's wrappers look for."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
MODULES = ("_raymarching", "_gridencoder", "_shencoder", "_ffmlp")


def activate():
    if HERE not in sys.path:
        sys.path.insert(0, HERE)
    return HERE
