"""Where does the host time of a training step go?  cProfile over N steps (development aid)."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from enerf_amd.network import NeRFNetwork  # noqa: E402
from enerf_amd.trainer import TrainHarness  # noqa: E402

import argparse
_ap = argparse.ArgumentParser()
_ap.add_argument("--rays", type=int, default=4096, help="64: the device is never the limit, what is timed is the host")
_ap.add_argument("--python-step", action="store_true", help="the Python-driven step instead of enerf_train_step_mse")
_ap.add_argument("--no-profile", action="store_true")
_ap.add_argument("--stub", action="store_true",
                 help="after the timed runs: the same steps with enerf_train_step_mse replaced by a no-op (nothing reaches the "
                      "device but the stage hand-over's event packets): what the Python around the one call costs")
_a = _ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = NeRFNetwork(encoding="hashgrid", bound=3, cuda_ray=True, out_dim_color=3).to(dev)
h = TrainHarness(model, occupancy="synthetic", world=1)
h.native_step = not _a.python_step
batches = bench.build_batches(8, _a.rays, dev, 0, 3)
def step(i):
    nxt = batches[(i + 1) % 8]
    return h.step_rgb(*batches[i % 8], next_rays=(nxt[0], nxt[1]))


# pin like bench.py does (an unpinned launch thread wanders over the box's 256 cores)
if hasattr(os, "sched_setaffinity") and len(os.sched_getaffinity(0)) >= 16:
    cores = sorted(os.sched_getaffinity(0))
    os.sched_setaffinity(0, cores[8:16])
for i in range(40):
    step(i)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for i in range(41, 41 + 15):          # (no update_extra_state inside: steps 41..55)
    step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"rays {_a.rays} native_step {h.native_step}: enqueue {1e3 * (t1 - t0) / 15:.3f} ms/step, drained after "
      f"{1e3 * (t2 - t0) / 15:.3f} ms/step")
if h.native_step:
    import ctypes
    from enerf_amd import _lib
    _lib.lib().enerf_debug_step_timing(1, None)
    for i in range(81, 81 + 14):
        step(i)
    torch.cuda.synchronize()
    arr = (ctypes.c_double * 16)()
    _lib.lib().enerf_debug_step_timing(0, arr)
    names = ["grid_fwd", "mlp_fwd_sigma", "mlp_fwd_colour", "composite", "mlp_bwd_colour", "mlp_bwd_sigma(+reduce)",
             "wait_signal", "near_far", "march", "grid_bwd", "table_adam"]
    print("host us per call inside enerf_train_step_mse:", {n: round(arr[k], 1) for k, n in enumerate(names)},
          "sum", round(sum(arr), 1))
if _a.stub and h.native_step:
    from enerf_amd import _lib
    real = _lib.lib().enerf_train_step_mse
    _lib.lib().enerf_train_step_mse = lambda a: 0
    for i in range(97, 97 + 3):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(100, 100 + 12):      # (steps 100..111: no update_extra_state inside)
        step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"Python around a stubbed enerf_train_step_mse: {1e3 * (t1 - t0) / 12:.3f} ms/step")
    prs = cProfile.Profile()
    prs.enable()
    for i in range(116, 116 + 12):
        step(i)
    prs.disable()
    torch.cuda.synchronize()
    _lib.lib().enerf_train_step_mse = real
    pstats.Stats(prs).sort_stats("tottime").print_stats(22)
if _a.no_profile:
    sys.exit(0)
pr = cProfile.Profile()
pr.enable()
n_prof = 0
for base in range(65, 65 + 16 * 8, 16):          # 8 windows of 14 steps between density-grid updates
    pr.disable()
    step(base - 1)                                # (the update step itself stays outside the profile)
    pr.enable()
    for i in range(base, base + 14):
        step(i)
        n_prof += 1
pr.disable()
print("profiled steps:", n_prof)
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(40)
