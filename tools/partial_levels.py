"""The partial density update's grid encode (1.57 M cells drawn from the grid, sorted), one level at a time, beside the full
sweep's -- ns per point and level (development aid):  python tools/partial_levels.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from enerf_amd import _lib, fused_network  # noqa: E402
from enerf_amd.network import NeRFNetwork  # noqa: E402
from enerf_amd.trainer import TrainHarness  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = NeRFNetwork(encoding="hashgrid", bound=3, cuda_ray=True, out_dim_color=3).to(dev)
h = TrainHarness(model, occupancy="synthetic")
L = _lib.lib()
C, H = int(model.cascade), int(model.grid_size)
N = H ** 3 // 4
P = C * 2 * N
indices = torch.empty(P, dtype=torch.int32, device=dev)
xyzs = torch.empty(P, 3, dtype=torch.float32, device=dev)
_lib.check(L.enerf_density_grid_cells(model.density_grid.data_ptr(), C, H, float(model.bound), N, ctypes.c_uint64(12345),
                                      indices.data_ptr(), xyzs.data_ptr(), _lib.stream_handle()), "cells")
perm = torch.randperm(P, device=dev)
x_shuffled = xyzs[perm].contiguous()
n_full = C * H ** 3


def timed(fn, mask):
    L.enerf_debug_grid_level_mask(mask)
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    _lib.prof.reset()
    _lib.prof.enable(True, only=("grid_fwd",))
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    _lib.prof.enable(False)
    ms, n = _lib.prof.read("grid_fwd")
    return 1e3 * ms / n


with torch.no_grad():
    sweep = lambda: fused_network.density_sigma_sweep(model, C, H, 7)
    part = lambda: fused_network.density_sigma(model, xyzs)
    shuf = lambda: fused_network.density_sigma(model, x_shuffled)
    print(f"full sweep {n_full} points, partial {P} points (occupied fraction {float((model.density_grid > 0).float().mean()):.3f})")
    a, b, c = timed(sweep, 0xffffffff), timed(part, 0xffffffff), timed(shuf, 0xffffffff)
    print(f"all levels: sweep {a:.1f} us = {1e3 * a / n_full:.3f} ns/pt   partial (sorted) {b:.1f} us = {1e3 * b / P:.3f} ns/pt   "
          f"partial shuffled {c:.1f} us = {1e3 * c / P:.3f} ns/pt")
    for lv in range(16):
        a, b, c = timed(sweep, 1 << lv), timed(part, 1 << lv), timed(shuf, 1 << lv)
        print(f"level {lv:2d}: sweep {1e3 * a / n_full:.4f}   partial sorted {1e3 * b / P:.4f}   shuffled {1e3 * c / P:.4f}  ns/pt")
L.enerf_debug_grid_level_mask(0xffffffff)
