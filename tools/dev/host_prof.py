"""cProfile of the host side of the training step (development aid): python tools/dev/host_prof.py [sharded]"""
import cProfile
import os
import pstats
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from enerf_amd.network import NeRFNetwork  # noqa: E402
from enerf_amd.trainer import TrainHarness  # noqa: E402

sharded = len(sys.argv) > 1 and sys.argv[1] == "sharded"
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29578", HSA_ENABLE_IPC_MODE_LEGACY="0")
os.sched_setaffinity(0, set(range(8, 16)))
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
if sharded:
    dist.init_process_group("nccl", rank=0, world_size=1)
batches = bench.build_batches(8, 4096, dev, 0, 3)
torch.manual_seed(0)
model = NeRFNetwork(encoding="hashgrid", bound=3, cuda_ray=True, out_dim_color=3).to(dev)
h = TrainHarness(model, occupancy="synthetic", world=2 if sharded else 1)
if sharded:
    h.comm_chunks, h.comm_dtype, h.comm_mode, h.fused_sharded, h.pretend_world = 4, None, "sharded", True, 0


def step(i):
    ro, rd, tg = batches[i % 8]
    nx = batches[(i + 1) % 8]
    h.step_rgb(ro, rd, tg, next_rays=(nx[0], nx[1]))


for i in range(48):
    step(i)
torch.cuda.synchronize()
# steady steps only: 15 between two updates
pr = cProfile.Profile()
n = 0
t_host = 0.0
for rep in range(20):
    while h.global_step % 16 != 1:
        step(h.global_step)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pr.enable()
    for k in range(12):
        step(h.global_step)
    pr.disable()
    t_host += time.perf_counter() - t0
    n += 12
    torch.cuda.synchronize()
print(f"{'sharded' if sharded else 'single'}: host {t_host / n * 1e6:.1f} us/step under cProfile over {n} steady steps")
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(30)
if sharded:
    dist.destroy_process_group()
