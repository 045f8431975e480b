"""The reference's training step on the reference's OWN native code, on an MI355X, beside this library's -- BASELINE configs[1]
(bound 3, L16 F2 T2^19, 4096 rays per step, nn.Linear nets), bench.py's batches and occupancy.  TEST INFRASTRUCTURE (it lives
under tests/ because it runs oracle/_ref).

  reference kernels : `run_cuda` op by op through the reference-shaped autograd Functions with `_backend` = oracle/_ref's
                      _ref_raymarching / _ref_gridencoder / _ref_shencoder (raymarching.cu, gridencoder.cu, shencoder.cu built
                      for gfx950 by oracle/build_ref.py), the nn.Linear nets as the plain Linear / ReLU loop on torch's GEMMs
                      (`fused_mlp.ENABLED = False`: what nerf/network.py runs on), F.mse_loss + autograd, the Python
                      update_extra_state (nerf/renderer.py:472-560), torch.optim.Adam.  Everything native on this route is
                      the reference's or PyTorch's; the Python above it is this repository's restatement of
                      nerf/renderer.py (same calls in the same order).
  drop-in           : that route with `_backend` = this library's pybind modules and its nets on this library's MLP kernels
                      behind autograd (bench.py's `dropin_route_rgb`).
  product           : what bench.py's headline times.

    gpurun -- 'python tests/refcheck/ref_route_speed.py > gpurun_out/ref_route_speed.txt'
"""
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import build_ref as br  # noqa: E402
import enerf_amd.raymarching as rmod  # noqa: E402
import enerf_amd.gridencoder as gmod  # noqa: E402
import enerf_amd.shencoder as smod  # noqa: E402
from enerf_amd import ext as ext_pkg, fused_mlp as fm_, fused_network as fn_, fused_render as fr_, density_update as du_  # noqa: E402
from enerf_amd.network import NeRFNetwork  # noqa: E402
from enerf_amd.trainer import TrainHarness  # noqa: E402

RAYS, BOUND, WARM, STEPS = 4096, 3, 20, 80


def run(kind, batches, dev):
    own = (rmod._backend, gmod._backend, smod._backend)
    try:
        if kind != "product":
            if kind.startswith("reference kernels"):
                mods = [br.load(n) for n in ("raymarching", "gridencoder", "shencoder")]
                fm_.ENABLED = False
            else:
                ext_pkg.activate()
                mods = [importlib.import_module(n) for n in ("_raymarching", "_gridencoder", "_shencoder")]
            rmod._backend, gmod._backend, smod._backend = mods
            gmod._layout_support = {}
            fr_.ENABLED = fn_.ENABLED = du_.ENABLED = False
        torch.manual_seed(0)
        m = NeRFNetwork(encoding="hashgrid", bound=BOUND, cuda_ray=True, out_dim_color=3).to(dev)
        if kind == "product":
            h = TrainHarness(m, occupancy="synthetic", world=1)
        else:
            h = TrainHarness(m, occupancy="synthetic", world=1, optimizer=torch.optim.Adam)
            h.native_step = h.manual_mse = h.fuse_table_adam = h.prefetch = False

        def step(i):
            nxt = batches[(i + 1) % len(batches)]
            return h.step_rgb(*batches[i % len(batches)], next_rays=(nxt[0], nxt[1]) if kind == "product" else None)
        for i in range(WARM):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(WARM, WARM + STEPS):
            loss = step(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / STEPS
        return {"route": kind, "ms_per_step": dt * 1e3, "rays_per_sec": RAYS / dt, "steps": STEPS, "warmup": WARM,
                "includes_update_extra_state_steps": STEPS // 16, "final_loss": float(loss),
                "samples_per_step": int(m.mean_count)}
    finally:
        rmod._backend, gmod._backend, smod._backend = own
        gmod._layout_support = {}
        fr_.ENABLED = fn_.ENABLED = du_.ENABLED = fm_.ENABLED = True


def main():
    dev = torch.device("cuda", 0)
    os.sched_setaffinity(0, set(range(8, 16)))
    batches = bench.build_batches(8, RAYS, dev, 0, BOUND)
    only = os.environ.get("REF_ROUTE_ONLY")                   # one route only (for a rocprofv3 --stats run of it)
    rows = [run(k, batches, dev) for k in ("reference kernels", "drop-in", "product") if only in (None, k)]
    ref = rows[0]["ms_per_step"]
    for r in rows:
        r["speedup_vs_reference_kernels"] = ref / r["ms_per_step"]
        print(json.dumps(r))


if __name__ == "__main__":
    main()
