"""Where does an update_extra_state step spend its time on the HOST, full sweep vs partial (iter_density >= 16)?
Per update step: wall time of the step call (enqueue), wall time until the device drained, device time (events)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from enerf_amd.network import NeRFNetwork          # noqa: E402
from enerf_amd.trainer import TrainHarness         # noqa: E402
sys.path.insert(0, ROOT)
import bench                                       # noqa: E402


def run(iter_density, profile=False):
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    m = NeRFNetwork(encoding="hashgrid", bound=3, cuda_ray=True, out_dim_color=3).to(dev)
    h = TrainHarness(m, occupancy="synthetic")
    m.iter_density = iter_density
    b = bench.build_batches(8, 4096, dev, 0, 3)
    rows = []
    for i in range(100):
        upd = h.global_step % 16 == 0
        nxt = b[(i + 1) % 8]
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        if profile and upd and i >= 32:
            import cProfile, pstats
            pr = cProfile.Profile()
            pr.enable()
            h.step_rgb(*b[i % 8], next_rays=(nxt[0], nxt[1]))
            pr.disable()
            pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
            profile = False
        else:
            h.step_rgb(*b[i % 8], next_rays=(nxt[0], nxt[1]))
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if i >= 32:
            rows.append((upd, (t1 - t0) * 1e3, (t2 - t0) * 1e3, e0.elapsed_time(e1)))
    for kind in (True, False):
        sel = [r for r in rows if r[0] == kind]
        n = len(sel)
        print(f"iter_density={iter_density} {'update' if kind else 'plain '} steps n={n}: enqueue {sum(r[1] for r in sel)/n:.3f} ms, "
              f"drained {sum(r[2] for r in sel)/n:.3f} ms, device {sum(r[3] for r in sel)/n:.3f} ms")


if __name__ == "__main__":
    run(0)
    run(16)
    run(16, profile=True)
