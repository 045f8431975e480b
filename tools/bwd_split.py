"""grid_encode_backward at the 4096-ray training batch, all levels or one (`--level k`), repeated: run under
rocprofv3 --kernel-trace --stats to split the time between k_grid_bwd_bin and k_grid_bwd_tile."""
import argparse, ctypes, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from enerf_amd import _lib, scene, raymarching
from enerf_amd.backends import _gridencoder as ge
from enerf_amd.gridencoder import GridEncoder

ap = argparse.ArgumentParser()
ap.add_argument("--level", type=int, default=-1)
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--rays", type=int, default=4096)
a = ap.parse_args()
dev = "cuda"; bound = 3; C = 3
bits = raymarching.packbits(scene.density_grid(bound, dev), 0.01)
g = torch.Generator(device=dev).manual_seed(1)
(ro, rd), _ = scene.training_batch(0, a.rays, dev, generator=g)
aabb = torch.tensor([-bound] * 3 + [bound] * 3, dtype=torch.float32, device=dev)
nears, fars = raymarching.near_far_from_aabb(ro, rd, aabb, 0.2)
counter = torch.zeros(2, dtype=torch.int32, device=dev)
xyzs, dirs, deltas, rays = raymarching.march_rays_train(ro, rd, bound, bits, C, 128, nears, fars, counter, -1, True, 128, True)
enc = GridEncoder(desired_resolution=2048 * bound).to(dev)
x01 = ((xyzs + bound) / (2 * bound)).contiguous()
B = x01.shape[0]
S = float(torch.log2(torch.tensor(enc.per_level_scale, dtype=torch.float64)))
gout = torch.randn(B, 32, device=dev); gemb = torch.zeros_like(enc.embeddings); dummy = torch.empty(1, device=dev)
if a.level >= 0:
    _lib.lib().enerf_debug_grid_level_mask(1 << a.level)
for _ in range(a.reps):
    ge.grid_encode_backward(gout, x01, enc.embeddings.data, enc.offsets, gemb, B, 3, 2, 16, S, 16, False, dummy, dummy, 0, layout=1)
torch.cuda.synchronize()
print("B", B)

lib = _lib.lib()
if hasattr(lib, "enerf_debug_bin_phases"):
    import numpy as np
    arr = (ctypes.c_ulonglong * 16)()
    lib.enerf_debug_bin_phases(arr, 1)
    for _ in range(a.reps):
        ge.grid_encode_backward(gout, x01, enc.embeddings.data, enc.offsets, gemb, B, 3, 2, 16, S, 16, False, dummy, dummy, 0, layout=1)
    torch.cuda.synchronize()
    lib.enerf_debug_bin_phases(arr, 0)
    v = np.array(list(arr), dtype=np.float64) / a.reps / 100.0      # wall_clock64: 100 MHz -> us, summed over workgroups
    names = ["A load+cell+contrib", "A aggregate+rows", "A rank atomics+sync", "A scan+reserve+sync", "A stage+sync", "A copy out",
             "", "", "B cursor+sync", "B zero+sync", "B accumulate+sync", "B flush"]
    for k, nm in enumerate(names):
        if nm:
            print(f"{nm:24s} {v[k]:10.1f} us summed over workgroups")
