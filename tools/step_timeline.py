"""Host enqueue time vs drained time of the training step, with and without the early march (development aid)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from enerf_amd.network import NeRFNetwork  # noqa: E402
from enerf_amd.trainer import TrainHarness  # noqa: E402

os.sched_setaffinity(0, set(range(8)))
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = NeRFNetwork(encoding="hashgrid", bound=3, cuda_ray=True, out_dim_color=3).to(dev)
h = TrainHarness(model, occupancy="synthetic", world=1)
batches = bench.build_batches(8, 4096, dev, 0, 3)


def run(n, nxt):
    for i in range(n):
        b = batches[(i + 1) % 8]
        h.step_rgb(*batches[i % 8], next_rays=(b[0], b[1]) if nxt else None)


for pf in (False, True, False, True):
    run(48, pf)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(96, pf)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"prefetch={pf}: enqueue {1e3 * (t1 - t0) / 96:.3f} ms/step, drained {1e3 * (t2 - t0) / 96:.3f} ms/step")

# cost of the density-grid update itself (runs every 16th step): full sweep (first 16 updates) and partial update
for label, it in (("full", 0), ("partial", 100)):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    h.model.iter_density = it
    h.model.update_extra_state()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a.record()
    for _ in range(8):
        h.model.iter_density = it
        h.model.update_extra_state()
    b.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"update_extra_state[{label}]: host {1e3 * (t1 - t0) / 8:.3f} ms, device {a.elapsed_time(b) / 8:.3f} ms")
