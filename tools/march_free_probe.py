"""What does the (side-stream) march cost the step?  The bench's 8 ray sets recur and the synthetic bitfield is
restored after every update, so a batch's march output never changes: memoise it and time the step without any march
work (development aid; an upper bound on what a cheaper marcher could give)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from enerf_amd import fused_render  # noqa: E402
from enerf_amd.network import NeRFNetwork  # noqa: E402
from enerf_amd.trainer import TrainHarness  # noqa: E402

os.sched_setaffinity(0, set(range(8, 16)))
dev = torch.device("cuda", 0)
batches = bench.build_batches(8, 4096, dev, 0, 3)
orig = fused_render.march_stage
memo = {}


def memoised(model, rays_o, rays_d, counter, mean_count, *a, **k):
    key = (rays_o.data_ptr(), mean_count)
    if key not in memo:
        pre = orig(model, rays_o, rays_d, counter, mean_count, *a, **k)
        memo[key] = (pre, counter.clone())
        return pre
    pre, cnt = memo[key]
    counter.copy_(cnt)
    out = dict(pre)
    out["counter"] = counter
    return out


from enerf_amd import _lib  # noqa: E402
configs = (("side stream, 1 block per CU", orig, True, 0), ("march inline", orig, False, 0),
           ("no march work (memoised)", memoised, True, 0),
           ("side stream, 1024 blocks (all rays at once)", orig, True, 1024),
           ("side stream, 512 blocks", orig, True, 512), ("side stream, 128 blocks", orig, True, 128))
for tag, fn, prefetch, blocks in configs + configs[::-1]:
    fused_render.march_stage = fn
    _lib.lib().enerf_debug_march_bg_blocks(blocks)
    memo.clear()
    torch.manual_seed(0)
    model = NeRFNetwork(encoding="hashgrid", bound=3, cuda_ray=True, out_dim_color=3).to(dev)
    h = TrainHarness(model, occupancy="synthetic")
    h.prefetch = prefetch

    def step(i):
        nx = batches[(i + 1) % 8]
        h.step_rgb(*batches[i % 8], next_rays=(nx[0], nx[1]))
    for i in range(64):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(64, 64 + 160):
        step(i)
    torch.cuda.synchronize()
    print(f"{tag:28s} {(time.perf_counter() - t0) / 160 * 1e3:.3f} ms/step")
fused_render.march_stage = orig
_lib.lib().enerf_debug_march_bg_blocks(0)
