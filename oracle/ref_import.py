"""Import recipe for the reference's *Python* (this container only; /root/reference never travels).

TEST INFRASTRUCTURE ONLY.  Used by oracle/make_golden.py to mint tests/golden/*.npz.  Must be run with
`python -B` (sys.dont_write_bytecode) so nothing is written into /root/reference, and the fake native backends must be
in sys.modules *before* the reference goes on sys.path, otherwise its backend.py JIT-builds (hipifies) the .cu files.
"""
import sys
import types
from unittest.mock import MagicMock

REFERENCE = "/root/reference"


def install():
    sys.dont_write_bytecode = True
    from . import backend as ob

    for m in ["trimesh", "cv2", "tensorboardX", "imageio", "mcubes", "torch_ema", "lpips", "skimage",
              "skimage.metrics", "h5py", "numba", "pyvista", "hdf5plugin", "configargparse", "dearpygui",
              "dearpygui.dearpygui", "tinycudann", "rich", "rich.console", "turtle", "tkinter"]:
        sys.modules.setdefault(m, MagicMock())
    # numba.jit is used as a bare decorator in utils/event_utils.py
    sys.modules["numba"].jit = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))

    # native backends: the reference's own wrappers (grid.py, sphere_harmonics.py, ffmlp.py) then run on CPU
    sys.modules["_gridencoder"] = ob.as_module("_gridencoder", ob.gridencoder_backend)
    sys.modules["_shencoder"] = ob.as_module("_shencoder", ob.shencoder_backend)
    sys.modules["_ffmlp"] = ob.as_module("_ffmlp", ob.ffmlp_backend)
    # raymarching/raymarching.py force-moves tensors with .cuda(): replace the whole package by a CPU facade
    rm = types.ModuleType("raymarching")

    def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
        import torch
        from . import oracle as O
        n, f = O.near_far_from_aabb(rays_o.detach().reshape(-1, 3).numpy(), rays_d.detach().reshape(-1, 3).numpy(),
                                    aabb.numpy(), float(min_near))
        return torch.from_numpy(n), torch.from_numpy(f)

    rm.near_far_from_aabb = near_far_from_aabb
    sys.modules["raymarching"] = rm
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
