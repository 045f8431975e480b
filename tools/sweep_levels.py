"""Full density sweep's grid encode, one level at a time (development aid): where do the 1.5 ms go?
   python tools/sweep_levels.py  ->  us per level (all cascades in one launch), and the same for random points"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from enerf_amd import _lib, fused_network  # noqa: E402
from enerf_amd.network import NeRFNetwork  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = NeRFNetwork(encoding="hashgrid", bound=3, cuda_ray=True, out_dim_color=3).to(dev)
L = _lib.lib()
n_pts = model.cascade * model.grid_size ** 3
x_rand = (torch.rand(n_pts, 3, device=dev) * 2 - 1) * 3


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
    sweep = lambda: fused_network.density_sigma_sweep(model, model.cascade, model.grid_size, 7)
    rand = lambda: fused_network.density_sigma(model, x_rand)
    print(f"points {n_pts}; all levels: sweep {timed(sweep, 0xffffffff):.1f} us   random {timed(rand, 0xffffffff):.1f} us")
    for lv in range(16):
        print(f"level {lv:2d}: sweep {timed(sweep, 1 << lv):7.1f} us   random {timed(rand, 1 << lv):7.1f} us")
L.enerf_debug_grid_level_mask(0xffffffff)
