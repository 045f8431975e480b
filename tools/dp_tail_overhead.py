"""Fixed cost of the data-parallel tail, measured on ONE GPU: RCCL with a world of one rank (the collectives degenerate
to a local pass), step time with the tail against the plain single-process step: shows the host-side cost of the
tail (launches per piece) and that its device side works on the real backend.  Caveat: with one rank RCCL runs
`oneRankReduce<PreMulSum>` over the buffer for ReduceOp.AVG (~120 us for the 52 MB table) -- an artefact of the
one-rank world that the ring kernels replace at N > 1, so the difference printed here is NOT the tail's fixed cost.
python tools/dp_tail_overhead.py [--rays 4096] [--steps 96]
(the library-side RCCL tail of rounds 3 / 4 -- csrc/dp_tail.hip -- lost every row of this table and was retired in round 5)"""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from enerf_amd.network import NeRFNetwork  # noqa: E402
from enerf_amd.trainer import TrainHarness  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--bound", type=int, default=3)
    ap.add_argument("--only", type=int, default=None, help="index of the one configuration to run (for profiling)")
    ap.add_argument("--host-slots", action="store_true", help="print enerf_debug_step_timing's per-call host times")
    a = ap.parse_args()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.sched_setaffinity(0, set(range(8, 16)))          # as bench.py pins rank 0
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    dev = torch.device("cuda", 0)
    batches = bench.build_batches(8, a.rays, dev, 0, a.bound)
    # ("sharded, own slice fused": the default sharded tail -- this rank's slice of the table keeps its record lists for
    #  the optimizer pass, TrainHarness._finish_sharded_fused; "dense": every tile made dense first, _finish_sharded)
    # (last column: pretend_world -- this one rank owns 1 / N of the table and pays an N-rank world's dense route for the
    #  rest: what a rank of N = 8 does between its backward and its optimizer pass, minus the wire)
    configs = (("single process", 1, 4, None, None, "allreduce", True, 0),
               ("torch tail, 1 piece", 2, 1, None, False, "allreduce", True, 0),
               ("torch tail, 4 pieces", 2, 4, None, False, "allreduce", True, 0),
               ("torch tail, sharded, own slice fused", 2, 4, None, False, "sharded", True, 0),
               ("torch tail, sharded, dense", 2, 4, None, False, "sharded", False, 0),
               ("sharded, own slice fused, owning 1/2", 2, 4, None, False, "sharded", True, 2),
               ("sharded, own slice fused, owning 1/8", 2, 4, None, False, "sharded", True, 8),
               ("torch tail, 4 pieces, bf16 wire", 2, 4, torch.bfloat16, False, "allreduce", True, 0))
    for tag, dp, chunks, dtype, native, mode, fused, pretend in (configs if a.only is None else configs[a.only:a.only + 1]):
        torch.manual_seed(0)
        model = NeRFNetwork(encoding="hashgrid", bound=a.bound, cuda_ray=True, out_dim_color=3).to(dev)
        h = TrainHarness(model, occupancy="synthetic", world=dp)
        h.comm_chunks, h.comm_dtype, h.comm_mode, h.fused_sharded, h.pretend_world = chunks, dtype, mode, fused, pretend

        def step(i):
            ro, rd, tg = batches[i % 8]
            nx = batches[(i + 1) % 8]
            h.step_rgb(ro, rd, tg, next_rays=(nx[0], nx[1]))
        for i in range(32):
            step(i)
        torch.cuda.synchronize()
        if a.host_slots:
            from enerf_amd import _lib
            _lib.lib().enerf_debug_step_timing(1, None)
        t0 = time.perf_counter()
        for i in range(32, 32 + a.steps):
            step(i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        if a.host_slots:
            import ctypes
            arr = (ctypes.c_double * 16)()
            _lib.lib().enerf_debug_step_timing(0, arr)
            print("   host us per library call inside the one-call step:", [round(v, 1) for v in arr])
        print(f"{tag:42s} {(time.perf_counter() - t0) / a.steps * 1e3:.3f} ms/step   (host enqueue "
              f"{(t1 - t0) / a.steps * 1e3:.3f})")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
