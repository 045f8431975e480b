"""update_extra_state's full sweep, query points generated inside the grid kernel (shipped) against written out first and read
as an input array: ms per update, hipEvent-timed, bound 3 (3 x 128^3 = 6.29 M points).   python tools/sweep_ab.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from enerf_amd import density_update  # noqa: E402
from enerf_amd.network import NeRFNetwork  # noqa: E402

torch.manual_seed(0)
model = NeRFNetwork(encoding="hashgrid", bound=3, cuda_ray=True, out_dim_color=3).cuda()
model.train()


def timed(n=6):
    ts = []
    for i in range(n + 2):
        model.iter_density = 0                       # (full sweeps)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        density_update.update(model)
        b.record()
        torch.cuda.synchronize()
        if i >= 2:
            ts.append(a.elapsed_time(b))
    return sum(ts) / len(ts), min(ts)


for flag in (True, False, True, False):
    density_update.SWEEP_IN_KERNEL = flag
    mean, best = timed()
    print(f"SWEEP_IN_KERNEL={flag}: {mean:.3f} ms per update (best {best:.3f})", flush=True)
