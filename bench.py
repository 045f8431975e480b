#!/usr/bin/env python
"""bench.py -- the hot path's headline measurement (BASELINE.json: train rays/sec + render Msamples/sec).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one full training iteration of BASELINE config 2 (shakeCarpet1-shaped: bound 3, hash grid L16 F2 T2^19,
HIP march_rays_train, nn.Linear MLPs, fp32, 4096 rays per GPU) on the synthetic scene of enerf_amd/scene.py:
update_extra_state (every 16 steps) + render + MSE + backward + [RCCL gradient all-reduce] + Adam.  Inputs (ray
batches, targets) are generated on the device before the timed region.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline      dominant HBM-bound kernel (grid_encode_forward): algorithmic bytes (1164 B/point x points per launch)
                / mean launch duration measured with hipEvents on the launch stream inside the timed region
  cpu_baseline  the reference's pure-PyTorch route (NeRFRenderer.run, 512 stratified samples per ray, nn.Linear nets,
                fwd + bwd + Adam) re-stated in enerf_amd and run on the host cores with the C oracle as the native
                backend, on a bounded sample (256 rays/step); rank 0, N=1 only.
  render        full 640x480 frame through the inference loop (fp32 nets, and the FFMLP bf16 nets: BASELINE configs[4])
  other_steps   one GPU: the event step of configs[2], the network_ff step, amp='bf16', the literal fp16 = True regime,
                the step after the 16 full density sweeps, the exact-fp32 MFMA arithmetic (enerf_mlp32_precision(0)) and
                `dropin_route_rgb`: the reference-shaped run_cuda op by op through the four pybind modules with autograd
                and torch.optim.Adam -- what an unmodified nerf/renderer.py:281-342 gets from this library
  gpu_reference_route  read from profiles/, NOT measured here: the same step on the reference's own kernels built for gfx950
                (tests/refcheck/ref_route_speed.py; bench.py may not touch oracle/)
  strong        BASELINE configs[3]: 65 536 rays per step over all ranks, timed like the main region
  comm_tuning   N > 1: measured ms/step per cut of the table-gradient all-reduce and per placement of the next batch's
                march (TrainHarness.tune_comm, untimed, before the warm-up) with the choices made, and the step time the
                opt-in bf16 wire format would give (reported only)
Defaults: N = 1, 20 warm-up + 200 timed steps (0.13 s), then the render / graph / CPU legs: about a minute in all.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
GRID_FWD_BYTES_PER_POINT = 1164   # 12 in + 16 levels x 8 corners x 8 B + 128 out
MFMA_F32_PEAK_TF = 157.3          # v_mfma_f32_32x32x2_f32, dense (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TF = 2500.0        # dense bf16 MFMA
MLP_LINEAR_FLOP_FWD = 18688       # nn.Linear nets, sigma 6144 + colour 12544 FLOP/sample forward
MLP_LINEAR_FLOP_STEP = 56064      # forward + dgrad + wgrad = 3 x forward
FFMLP_FLOP_FWD = 36864            # FFMLP nets (padded dims), sigma 14336 + colour 22528
PMC_FILE = "profiles/r06_pmc_hbm_bench.json"
REF_ROUTE_FILE = "profiles/r04_ref_route_speed.txt"


def reference_route_on_file():
    """NOT measured by this run (bench.py may not touch oracle/): the configs[1] training step on the reference's OWN native
    code on an MI355X -- its raymarching.cu / gridencoder.cu / shencoder.cu built for gfx950 (oracle/build_ref.py), nn.Linear
    on torch's GEMMs, autograd, torch.optim.Adam -- as tests/refcheck/ref_route_speed.py last wrote it to profiles/.  The GPU
    baseline beside `cpu_baseline`; None when the file is absent."""
    try:
        rows = [json.loads(line) for line in open(os.path.join(ROOT, REF_ROUTE_FILE)) if line.startswith("{")]
        ref = next(r for r in rows if r["route"] == "reference kernels")
        return {"ms_per_step": ref["ms_per_step"], "rays_per_sec": ref["rays_per_sec"], "steps": ref["steps"],
                "source": REF_ROUTE_FILE + " (tests/refcheck/ref_route_speed.py on an MI355X; not this run)",
                "same_script_product_ms_per_step": next(r for r in rows if r["route"] == "product")["ms_per_step"]}
    except Exception:
        return None
TABLE_OPT_BYTES_PER_PARAM = 24     # Adam over the table: p, m, v read + written, 4 B each


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # 125 ms timed: one scheduling hiccup no longer moves the figure
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rays", type=int, default=4096, help="rays per GPU per step")
    ap.add_argument("--global-rays", type=int, default=0,
                    help="strong scaling: this many rays per step over ALL ranks (BASELINE configs[3]: 65536 over 8 GPUs "
                         "= 8192 per rank); overrides --rays, the line says \"scaling\": \"strong\"")
    ap.add_argument("--bound", type=int, default=3)
    ap.add_argument("--mode", choices=["rgb", "events"], default="rgb")
    ap.add_argument("--render-frames", type=int, default=2, help="full 640x480 inference frames timed after training")
    ap.add_argument("--render-rounds", action="store_true",
                    help="render leg: the reference's round schedule (8x wider rounds) instead of the whole-frame pass")
    ap.add_argument("--net", choices=["linear", "ff"], default="linear",
                    help="linear = nerf/network.py (BASELINE configs[1-3]); ff = nerf/network_ff.py FFMLP bf16 (configs[4])")
    ap.add_argument("--amp-bf16", dest="fp16", action="store_true",
                    help="TrainHarness(amp='bf16') -- not the reference's fp16 regime: the closed-form step with the "
                         "networks on bf16 operands (fp32 accumulation, fp32 table, no loss scaling); dtype bf16-amp")
    ap.add_argument("--fp16", dest="fp16_true", action="store_true",
                    help="TrainHarness(fp16=True): the shipped configs' fp16 = True regime on the closed-form step -- fp16 "
                         "operands (v_mfma_f32_32x32x16_f16), the GradScaler protocol on the device (dtype f16-amp)")
    ap.add_argument("--fp16-autocast", dest="fp16_autocast", action="store_true",
                    help="TrainHarness(fp16='autocast'): the same regime op by op: autocast(float16) + GradScaler, half hash "
                         "table (588 B/point) (dtype f16-autocast)")
    ap.add_argument("--graphs", action="store_true",
                    help="replay render+loss+backward of the rgb step as a HIP graph (opt-in: the per-kernel hipEvent "
                         "timing behind `roofline` only sees the launches that stay eager)")
    ap.add_argument("--no-pin", action="store_true",
                    help="do not pin this rank to a block of 8 host cores (unpinned, the launch thread migrates over the "
                         "box's 256 cores and the host-bound step time jitters by 10-15 %)")
    ap.add_argument("--graph-leg-steps", type=int, default=0, help="(retired: graph replay lost to the one-call eager "
                    "step, 0.50 vs 0.41 ms; --graphs still runs the main region that way)")
    ap.add_argument("--comm-bf16", action="store_true",
                    help="N > 1: all-reduce the hash-table gradient in bf16 (opt-in; fp32 is the default and the headline)")
    ap.add_argument("--prefetch-at", choices=("forward", "mlp_backward"), default=None,
                    help="tuning: where the next batch's march is released on the side stream (default: the harness's)")
    ap.add_argument("--no-live-timing", action="store_true",
                    help="tuning: no hipEvent timing inside the timed region (the line then carries no `roofline`)")
    ap.add_argument("--no-early-budget", action="store_true",
                    help="A/B switch: update_extra_state waits for its own read-back before the step is queued (the route "
                         "before TrainHarness.early_budget)")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="do not march the next batch early (side stream under the backward / gradient all-reduce)")
    ap.add_argument("--time-every", type=int, default=4,
                    help="live hipEvent timing of the roofline kernel: one launch in N of the timed region (a timed launch "
                         "costs ~10-15 us of queue time: completion signals + a marker launch; 1 = every launch)")
    ap.add_argument("--prof-all", action="store_true", help="hipEvent-time every kernel family, not just grid_encode")
    ap.add_argument("--other-legs", type=int, default=48,
                    help="steps of each extra leg (event step of configs[2], network_ff step, fp16=True step); 0 = skip")
    ap.add_argument("--only-legs", default="", help="comma-separated tags of other_steps to run (default: all)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="(internal) print the cpu_baseline object and exit")
    ap.add_argument("--cpu-rays", type=int, default=256)
    ap.add_argument("--cpu-threads", type=int, default=16, help="torch / OpenMP threads of the cpu_baseline leg (0 = all cores)")
    ap.add_argument("--probe-steps", type=int, default=15,
                    help="extra steps after the timed region with every kernel family hipEvent-timed (MFMA roofline); 0 = skip")
    ap.add_argument("--cpu-budget-s", type=float, default=12.0, help="CPU work to spend on the cpu_baseline sample")
    # development aids: exercise the N > 1 code path on a single-GPU box (gloo all-reduce, every rank on one device)
    ap.add_argument("--verify-samples", action="store_true",
                    help="also count the timed steps' samples with one launch per render and compare (development aid)")
    ap.add_argument("--march-bg-blocks", type=int, default=0, help="tuning: workgroups of the side-stream march (0 = auto)")
    ap.add_argument("--no-march-clip", action="store_true",
                    help="tuning: do not test rays against the occupied cells' bounding box before marching")
    ap.add_argument("--no-comm-tune", action="store_true", help="N > 1: keep comm_chunks = 4 instead of timing 1/2/4/8")
    ap.add_argument("--backend", default=None, help="torch.distributed backend override (default: nccl = RCCL)")
    ap.add_argument("--strong-rays", type=int, default=65536,
                    help="second object `strong` on the line: BASELINE configs[3], this many rays per step over ALL ranks "
                         "(65536 / N per rank), timed after the main region on a model of its own; 0 = skip")
    ap.add_argument("--strong-steps", type=int, default=32, help="timed steps of the `strong` leg (after 20 warm-up steps)")
    ap.add_argument("--launch-check", action="store_true",
                    help="only the launch plumbing: rendezvous, barrier, MAX all-reduce, rank 0 prints one JSON line; no "
                         "compute, no GPU needed (tests/test_bench_launch.py drives `--gpus 2 --backend gloo` through it)")
    ap.add_argument("--force-device", type=int, default=None, help="put every rank on this device index")
    ap.add_argument("--mlp-exact", action="store_true",
                    help="development: the main region on the exact-fp32 MFMA kernels (enerf_mlp32_precision(0))")
    return ap.parse_args()


def build_batches(n_batches, n_rays, device, rank, bound):
    from enerf_amd import scene
    g = torch.Generator(device=device)
    g.manual_seed(1234 + rank)
    batches = []
    for b in range(n_batches):
        (ro, rd), inds = scene.training_batch(b, n_rays, device, generator=g, rank=rank)
        # synthetic target colour: a smooth function of the ray (there is no dataset; "data": "synthetic")
        target = (0.5 + 0.5 * torch.sin(rd * 4.0 + ro)).clamp(0, 1).contiguous()
        batches.append((ro, rd, target))
    return batches


def pmc_traffic(points_per_launch):
    """HBM-side bytes per grid_encode_forward launch from the committed rocprofv3 PMC passes (separate --pmc FETCH_SIZE /
    --pmc WRITE_SIZE runs of this very command).  The counters are averaged over EVERY k_grid_fwd dispatch of the profiled
    process (warm-up, timed region, probe and graph legs), so they are divided by the points per launch of that same
    population -- `grid_fwd_points_per_launch_all_dispatches`, which the profiled run's own JSON line carries
    (roofline.lifetime) -- and the resulting bytes per point are rescaled by this run's points per launch.  FETCH_SIZE is in
    KB and, per MI355X_MICROARCH.md (HBM), counts 64 B per 128-byte request on gfx950, so it is doubled; WRITE_SIZE is
    taken as is (uncalibrated).  None if absent."""
    try:
        path = os.path.join(ROOT, PMC_FILE)
        if not os.path.exists(path):
            path = os.path.join(ROOT, "profiles", "r05_pmc_hbm_bench.json")
        with open(path) as f:
            j = json.load(f)
        d = j["k_grid_fwd<float, 3, 2>"]
        meta = j["_meta"]
        ref_pts = float(meta.get("grid_fwd_points_per_launch_all_dispatches") or meta["grid_fwd_points_per_launch"])
        per_point = (2.0 * d["FETCH_SIZE_avg"] + d["WRITE_SIZE_avg"]) * 1024.0 / ref_pts
        return per_point * points_per_launch, os.path.relpath(path, ROOT)
    except Exception:
        return None, None


def table_backward_object(points, bin_ms, adam_ms, n_params, fused, where):
    """One object for "table backward + optimizer" (grid_encode_backward through the Adam update of the table):
    algorithmic bytes = 1164 B/point x points (the backward's scatter counted once) + 24 B x table
    parameters (p, m, v read and written), time = binning pass + (fused) k_grid_tile_adam, which does the scatter's
    summing AND the optimizer -- the scatter bytes belong to that pair, not to the binning pass alone."""
    out = {"points_per_launch": points, "binning_pass_ms": bin_ms, "timed_in": where,
           # what the binning pass itself moves: 12 B/point in, 128 B/point of dL/dfeat in, 16 levels x 8 corners x
           # 10-byte records out
           "binning_pass_GBs": points * (12 + 128 + 16 * 8 * 10) / (bin_ms * 1e-3) / 1e9,
           "binning_pass_bytes_per_point": 12 + 128 + 16 * 8 * 10}
    if fused and adam_ms is not None:
        bytes_alg = points * GRID_FWD_BYTES_PER_POINT + TABLE_OPT_BYTES_PER_PARAM * n_params
        t = (bin_ms + adam_ms) * 1e-3
        out.update({"tile_adam_ms": adam_ms, "algorithmic_bytes": bytes_alg, "achieved": bytes_alg / t / 1e9,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_alg / t / 1e9 / HBM_PEAK_GBS,
                    "kernels": "k_grid_bwd_bin + k_grid_tile_adam (record lists -> LDS tile sums -> Adam; the table "
                               "gradient is never materialised)"})
    elif not fused:
        out["kernels"] = "k_grid_bwd_bin + k_grid_bwd_tile (the dense gradient exists: all-reduce, then k_adam_multi)"
    return out


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args):
    """Reference pure-PyTorch route on the host: run() + nn.Linear + oracle-backed encoders, fwd+bwd+Adam."""
    import enerf_amd.raymarching as rm
    import enerf_amd.gridencoder as ge
    import enerf_amd.shencoder as sh
    from oracle import backend as ob
    from enerf_amd.network import NeRFNetwork
    from enerf_amd import scene
    saved = (rm._backend, rm._DEVICE, ge._backend, sh._backend)
    rm._backend, rm._DEVICE, ge._backend, sh._backend = (ob.raymarching_backend, "cpu", ob.gridencoder_backend,
                                                         ob.shencoder_backend)
    try:
        # 16 hash-grid levels = 16 OpenMP tasks; more torch threads than that only adds fork/join cost on the
        # small (131k x 64) GEMMs: tools/cpu_threads_sweep.py on the GPU box (profiles/r02_cpu_threads_sweep.json)
        # measured 16 threads fastest.  `cores` reports the threads actually used, `cores_total` what the box has.
        from oracle import oracle as O
        cores_total = os.cpu_count() or 1
        cores = min(cores_total, args.cpu_threads) if args.cpu_threads > 0 else cores_total
        prev_threads = torch.get_num_threads()
        torch.set_num_threads(cores)
        O.set_threads(cores)
        torch.manual_seed(0)
        model = NeRFNetwork(encoding="hashgrid", bound=args.bound, cuda_ray=False, out_dim_color=3)
        opt = torch.optim.Adam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
        model.train()
        n, T = args.cpu_rays, 512
        g = torch.Generator().manual_seed(7)
        steps, t_total = 0, 0.0
        for it in range(1 + 512):
            (ro, rd), _ = scene.training_batch(it, n, "cpu", generator=g)
            target = (0.5 + 0.5 * torch.sin(rd * 4.0 + ro)).clamp(0, 1)
            t0 = time.perf_counter()
            opt.zero_grad(set_to_none=True)
            out = model.render(ro, rd, staged=False, bg_color=None, perturb=True, num_steps=T, upsample_steps=0,
                               out_dim_color=3)
            loss = torch.nn.functional.mse_loss(out["image"], target)
            loss.backward()
            opt.step()
            dt = time.perf_counter() - t0
            if it == 0:
                continue        # warm-up
            steps += 1
            t_total += dt
            if t_total > args.cpu_budget_s:
                break
        rays_s = steps * n / t_total
        return {"value": rays_s, "unit": "rays/s", "cores": cores, "cores_total": cores_total, "cpu_model": _cpu_model(),
                "kind": "port",
                "network_evals_per_sec": rays_s * T,
                "sample": f"{steps} steps x {n} rays x {T} stratified samples/ray (NeRFRenderer.run + nn.Linear, "
                          f"fwd+bwd+Adam, hash grid via the C oracle with one OpenMP task per level), "
                          f"{t_total:.1f} s of CPU work, torch threads={cores}"}
    finally:
        rm._backend, rm._DEVICE, ge._backend, sh._backend = saved
        torch.set_num_threads(prev_threads)


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks ourselves, one process per
    GPU, through the same module the driver uses (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py <same flags>`), on a free port.  The ranks inherit stdout, so rank
    0's ONE JSON line is this process's output; the exit code is the launcher's.  The reference's own hook for this is
    `torch.distributed.get_rank/world_size` read by its Trainer (nerf/utils.py:299-300,351-354): launch is left to the
    user there as well."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC only on this driver stack
    env.setdefault("OMP_NUM_THREADS", "8")                   # (torchrun would set 1 and say so on stderr)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def launch_check(args):
    """The launch contract without the hot path (it has no CPU form): what a rank does around the timed region --
    rendezvous from the launcher's environment, barrier, MAX over ranks of a local time, rank 0 prints the line."""
    from enerf_amd import parallel
    import torch.distributed as dist
    rank, world, local_rank = parallel.init_from_env(backend=args.backend)
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "max_over_ranks": float(t.item()),
                          "backend": dist.get_backend() if world > 1 else None,
                          "global_rays": args.global_rays or world * args.rays,
                          "strong_rays_per_rank": args.strong_rays // world if args.strong_rays else 0}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    os.environ.setdefault("ENERF_DIST_TIMEOUT_S", "300")    # (this program's phases are seconds long: a missing rank ends it)
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args)))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    if args.launch_check:
        return launch_check(args)
    from enerf_amd import parallel, _lib
    from enerf_amd.backends import _gridencoder as gb, _raymarching as rb
    import torch.distributed as dist

    rank, world, local_rank = parallel.init_from_env(backend=args.backend)
    if args.force_device is not None:
        local_rank = args.force_device
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
    if args.global_rays:
        if args.global_rays % world:
            raise SystemExit(f"--global-rays {args.global_rays} does not divide over {world} ranks")
        args.rays = args.global_rays // world
    # one process per GPU, pinned to its own block of host cores (launch thread + autograd thread + HIP runtime threads)
    full_affinity = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    if full_affinity is not None and not args.no_pin and len(full_affinity) >= 16:
        cores = sorted(full_affinity)
        k = 8
        start = (k * (1 + local_rank)) % (len(cores) - k + 1)
        os.sched_setaffinity(0, cores[start:start + k])
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    if args.net == "ff":
        from enerf_amd.network_ff import NeRFNetwork
    else:
        from enerf_amd.network import NeRFNetwork
    from enerf_amd.trainer import TrainHarness
    from enerf_amd.events import EventOptions

    if args.mlp_exact:
        _lib.lib().enerf_mlp32_precision(0)
    if args.march_bg_blocks:
        _lib.lib().enerf_debug_march_bg_blocks(args.march_bg_blocks)
    if args.no_march_clip:
        _lib.lib().enerf_debug_march_clip(0)
    torch.manual_seed(0)
    model = NeRFNetwork(encoding="hashgrid", bound=args.bound, cuda_ray=True, out_dim_color=3).to(device)
    if args.render_rounds:
        from enerf_amd import frame
        frame.FRAME_ENABLED = False
        model.infer_batch_mult = 8
    harness = TrainHarness(model, occupancy="synthetic", world=world, use_graphs=args.graphs,
                           fp16="autocast" if args.fp16_autocast else args.fp16_true,
                           amp="bf16" if args.fp16 and not (args.fp16_autocast or args.fp16_true) else None)
    if args.fp16 or args.fp16_autocast or args.fp16_true:
        args.probe_steps = 0
        args.graph_leg_steps = 0
    harness.prefetch = not args.no_prefetch
    harness.early_budget = not args.no_early_budget
    if args.prefetch_at:
        harness.prefetch_at = args.prefetch_at
    if args.comm_bf16:
        harness.comm_dtype = torch.bfloat16
    n_table_params = int(model.encoder.embeddings.numel())
    parallel.broadcast_state(model)
    batches = build_batches(8, args.rays, device, rank, args.bound)
    ev_opt = EventOptions(C_thres=0.2, use_luma=True, linlog=True, event_only=True)

    def one_step(i):
        ro, rd, target = batches[i % len(batches)]
        if args.mode == "rgb":
            nxt = batches[(i + 1) % len(batches)]
            return harness.step_rgb(ro, rd, target, next_rays=(nxt[0], nxt[1]))
        return harness.step_events(event_data(i), ev_opt, next_data=event_data(i + 1))

    # --mode events: the step starts at the event stream, not at ready-made rays -- per step ONE launch draws 4096
    # event pairs from a pixel-grouped table of 2 M synthetic events, sums their polarities, interpolates the camera
    # pose at both event times from a 64-pose track (Slerp + cubic, as the reference's provider does with scipy on the
    # host) and emits the two ray sets (enerf_amd/event_sampler.event_pair_rays, csrc/event_pairs.hip)
    ev_state = {}

    def event_stream():
        from enerf_amd import scene
        from enerf_amd.event_sampler import build_event_tables
        from enerf_amd.pose_interp import PoseTrack
        g = torch.Generator(device=device).manual_seed(4321 + rank)
        n = 2_000_000
        span_ns = 2.0e8
        ev = torch.stack([torch.randint(0, scene.W, (n,), device=device, generator=g).float(),
                          torch.randint(0, scene.H, (n,), device=device, generator=g).float(),
                          torch.rand(n, device=device, generator=g) * span_ns,
                          torch.randint(0, 2, (n,), device=device, generator=g).float() * 2 - 1], dim=1)
        K = 64
        times = torch.linspace(-1.0, span_ns + 1.0, K, dtype=torch.float64)
        c2w = torch.stack([scene.pose(3.0 + 0.8 * k / (K - 1)) for k in range(K)])        # 9 degrees along the circle
        track = PoseTrack(times.numpy(), c2w[:, :3, :3].numpy(), c2w[:, :3, 3].numpy(), device=device)
        return build_event_tables(ev), track, g

    def event_data(i):
        from enerf_amd import scene
        from enerf_amd.event_sampler import event_pair_rays
        if "tables" not in ev_state:
            ev_state["tables"], ev_state["track"], ev_state["gen"] = event_stream()
            ev_state["images"] = torch.zeros(1, args.rays, 3, device=device)
        cache = ev_state.setdefault("cache", {})
        if i not in cache:
            for k in [k for k in cache if k < i - 1]:
                del cache[k]
            d = event_pair_rays(ev_state["tables"], ev_state["track"], scene.INTRINSICS, args.rays, 0,
                                generator=ev_state["gen"])
            cache[i] = {"images": ev_state["images"], "rays_evs_o1": d["rays_evs_o1"], "rays_evs_d1": d["rays_evs_d1"],
                        "rays_evs_o2": d["rays_evs_o2"], "rays_evs_d2": d["rays_evs_d2"], "pols": d["pols"]}
        return cache[i]


    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    comm_tuning = None
    if world > 1 and args.mode == "rgb" and not args.no_comm_tune:
        comm_tuning = harness.tune_comm(one_step)       # untimed: chooses how the gradient all-reduce is cut
        if comm_tuning:
            comm_tuning = {"ms_per_step": comm_tuning, "chosen_chunks": harness.comm_chunks,
                           "sharded_tail_ms_per_step": harness.tuned["sharded_ms_per_step"],
                           "chosen_tail": harness.comm_mode,
                           "prefetch_at_ms_per_step": harness.tuned["prefetch_at_ms_per_step"],
                           "chosen_prefetch_at": harness.prefetch_at,
                           }
            if harness.comm_dtype is None:      # reported only: what the opt-in 16-bit wire format would give here
                comm_tuning["bf16_wire_ms_per_step"] = harness.probe_comm_dtype(one_step, torch.bfloat16)
        # the tuning steps must not change what the timed region holds: back to step 0 of the density-grid schedule
        # (full sweeps for the first 16 updates, renderer.py:484), as in the N = 1 run
        harness.global_step = 0
        model.iter_density = 0
    # warm-up runs with the same timing hooks as the timed region, so their events exist before the clock starts
    # the timed region carries the event timing of the roofline kernel only (two kernel-attached events per launch);
    # the table's backward is timed in the probe steps after the region when there are any (its event pair costs
    # ~10 us per step: measured 0.424 -> 0.416 ms), inside the region otherwise
    probe_ok = args.probe_steps > 0 and args.mode == "rgb" and not args.graphs
    timed_families = None if args.prof_all else (("grid_fwd",) if probe_ok else ("grid_fwd", "grid_bwd"))
    if args.no_live_timing:
        timed_families = ()
    _lib.prof.sample_every(max(1, args.time_every))
    _lib.prof.enable(True, only=timed_families)
    for i in range(args.warmup):
        one_step(i)
    sync()

    samples_acc = torch.zeros((), dtype=torch.int64, device=device)
    gb.STATS.update(fwd_points=0, fwd_calls=0, bwd_points=0, bwd_calls=0, budget_rows=0)
    _lib.prof.reset()
    # live hipEvent timing of the roofline kernel (and its backward) over the timed region; the other kernel families
    # are in profiles/ (rocprofv3) -- timing all of them costs ~25 event records per step, 5 % of a 1 ms step.
    # One launch in --time-every is timed (deterministically: launch k of the region when k % N == 0): a timed launch
    # holds the queue for ~15 us (tools/step_timeline.py: marker kernel 4.4 + 6.8 us before it + 4.8 us after the kernel),
    # 5 % of a 0.3 ms step when every launch carries it; the figures below are over the timed launches only
    # (enerf_prof_read_units: their points).
    _lib.prof.sample_every(max(1, args.time_every))
    _lib.prof.enable(True, only=timed_families)
    sync()

    # ray samples of the timed steps: the march's scan pass keeps a running total on the device
    # (enerf_march_train_samples) -- no bookkeeping launch inside the timed region.  A march issued ahead of its step
    # (side-stream prefetch) belongs to the step that renders it: its count is moved across the region's two borders.
    def pending_marches():
        stash = getattr(model, "_premarched", None) or {}
        return sum(int(model.step_counter[p["slot"], 0]) for p in stash.values() if p.get("slot") is not None)

    def marched_total(reset):
        import ctypes
        v = ctypes.c_uint64(0)
        _lib.check(_lib.lib().enerf_march_train_samples(ctypes.byref(v), int(reset), _lib.stream_handle()),
                   "march_train_samples")
        return int(v.value)

    torch.cuda.synchronize()
    ahead_at_start = pending_marches()
    marched_total(reset=True)
    sync()
    t0 = time.perf_counter()
    # (graph mode counts per step: a capture inside the timed region runs warm-up marches that are not steps)
    per_step = torch.zeros((), dtype=torch.int64, device=device) if args.verify_samples or args.graphs else None
    # events at the borders between runs of like steps (no synchronisation): "cold" = before the first sample budget
    # existed (the reference's first 16 steps), "steady", and the steps that start with update_extra_state.  Not one
    # event per step: an event record is a barrier packet, ~8 us of idle queue per step at this step length.
    def step_class(i):
        gs = harness.global_step
        if gs % harness.update_interval == 0:
            return "with_update_extra_state"
        return "cold" if model.mean_count <= 0 else "steady"

    runs = []                                   # [class, steps, start event, end event]
    for i in range(args.warmup, args.warmup + args.steps):
        cls = step_class(i)
        if not runs or runs[-1][0] != cls:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            if runs:
                runs[-1][3] = ev
            runs.append([cls, 0, ev, None])
        runs[-1][1] += 1
        one_step(i)
        if per_step is not None:      # the per-step bookkeeping the running total replaces (one tiny launch per render)
            slot = getattr(model, "rendered_counter_slot", None)
            slot = (model.local_step - 1) % 16 if slot is None else slot
            per_step.add_(model.step_counter[slot, 0])
            if args.mode == "events":
                per_step.add_(model.step_counter[(slot - 1) % 16, 0])
    runs[-1][3] = torch.cuda.Event(enable_timing=True)
    runs[-1][3].record()
    t_enqueued = time.perf_counter()
    sync()
    t1 = time.perf_counter()
    if args.graphs:
        samples_acc += per_step
    else:
        samples_acc += marched_total(reset=False) + ahead_at_start - pending_marches()
        if per_step is not None:
            assert int(per_step.item()) == int(samples_acc.item()), (int(per_step.item()), int(samples_acc.item()))
    _lib.prof.enable(False)
    split = {}
    for name in ("cold", "steady", "with_update_extra_state"):
        sel = [r for r in runs if r[0] == name]
        n = sum(r[1] for r in sel)
        split[name] = {"ms_per_step": sum(r[2].elapsed_time(r[3]) for r in sel) / n if n else None, "steps": n}
    # The training launches carry a sample BUDGET (rows) of which the grid kernels encode / bin only what the marcher filled
    # (a device-side count: csrc/common.h grid_valid_rows): real points = rows x this rank's samples / its budget rows
    local_samples = int(samples_acc.item())
    fill = None
    if gb.STATS.get("budget_rows"):
        f = local_samples / gb.STATS["budget_rows"]
        fill = f if 0.0 < f <= 1.0 else None
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        dist.all_reduce(samples_acc, op=dist.ReduceOp.SUM)
    elapsed = float(elapsed.item())
    total_samples = int(samples_acc.item())
    renders_per_step = 2 if args.mode == "events" else 1
    total_rays = world * args.rays * renders_per_step * args.steps

    kernels = {}
    for name in ("grid_fwd", "grid_bwd", "march_train", "composite_fwd", "composite_bwd", "sh_fwd"):
        ms, n = _lib.prof.read(name)
        if n:
            units, seen = _lib.prof.read_units(name)
            kernels[name] = {"avg_ms": ms / n, "launches": int(n), "launches_in_region": int(seen),
                             "timed_one_in": max(1, args.time_every)}
            if units:
                kernels[name]["units_per_timed_launch"] = units / n
    _lib.prof.sample_every(1)
    roofline = None
    if "grid_fwd" in kernels and gb.STATS["fwd_calls"]:
        # points per launch of the launches that were TIMED (the region mixes 133 k-point training batches with
        # 6.3 M-point density sweeps: time and points must come from the same launches)
        rows = kernels["grid_fwd"].get("units_per_timed_launch", gb.STATS["fwd_points"] / gb.STATS["fwd_calls"])
        # rows -> real points for training-batch launches (a density sweep's 6.3 M rows are all real; the region's timed
        # launches are one kind or the other when one launch in --time-every is timed)
        train_launches = fill is not None and rows < 1.0e6
        pts = rows * fill if train_launches else rows
        # (--fp16: the timed launches mix half-table training batches, 588 B/point, with fp32 density sweeps; the
        # fp32 figure is kept so that the fraction is a lower bound)
        achieved = pts * GRID_FWD_BYTES_PER_POINT / (kernels["grid_fwd"]["avg_ms"] * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(pts)
        roofline = {"bound": "hbm", "kernel": "grid_encode_forward", "achieved": achieved, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "traffic_source": None if traffic is None else
                    f"{traffic_src}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (not this run), "
                    f"bytes per point x this run's points per launch",
                    "points_per_launch": pts, "rows_per_launch": rows,
                    "points_are": ("real samples: the launch's budget rows x (device-side sample total / budget rows of "
                                   "the region); the kernel skips the unfilled rows" if train_launches else "rows launched"),
                    "avg_launch_ms": kernels["grid_fwd"]["avg_ms"],
                    "timed_launches": kernels["grid_fwd"]["launches"],
                    "launches_in_region": kernels["grid_fwd"]["launches_in_region"],
                    "points_per_launch_all_region": gb.STATS["fwd_points"] / gb.STATS["fwd_calls"],
                    # every grid_encode_forward launch of this process so far (what a whole-process profile averages over)
                    "lifetime": {"launches": gb.LIFETIME["fwd_calls"],
                                 "points_per_launch": gb.LIFETIME["fwd_points"] / max(gb.LIFETIME["fwd_calls"], 1)}}
        if "grid_bwd" in kernels and gb.STATS["bwd_calls"]:
            ptsb = kernels["grid_bwd"].get("units_per_timed_launch", gb.STATS["bwd_points"] / gb.STATS["bwd_calls"])
            roofline["table_backward"] = table_backward_object(ptsb, kernels["grid_bwd"]["avg_ms"], None, n_table_params,
                                                               harness.fuse_table_adam and world == 1,
                                                               "the timed region")

    # ---- MFMA probe (not part of `value`): a few more steps with the MLP kernel families hipEvent-timed as well, no
    # density-grid update in between (its sigma-only sweep is a different launch shape).  Timing every family costs
    # ~5 % of a step, which is why the timed region above only times the grid kernels.
    roofline_mfma = None
    if probe_ok:
        keep_interval = harness.update_interval
        harness.update_interval = 10 ** 9
        try:
            base = args.warmup + args.steps
            one_step(base)
            sync()
            _lib.prof.reset()
            _lib.prof.enable(True, only=("ffmlp_fwd", "ffmlp_bwd", "mlp_reduce", "grid_bwd", "table_adam"))
            bwd_before = (gb.STATS["bwd_points"], gb.STATS["bwd_calls"])
            before = marched_total(reset=False)
            for i in range(base + 1, base + 1 + args.probe_steps):
                one_step(i)
            sync()
            _lib.prof.enable(False)
            probe_samples = (marched_total(reset=False) - before) / args.probe_steps      # (one march is always ahead)
            fwd_ms, nf = _lib.prof.read("ffmlp_fwd")
            bwd_ms, nb = _lib.prof.read("ffmlp_bwd")
            red_ms, _ = _lib.prof.read("mlp_reduce")
            mode = 2 if (args.net == "ff" or args.fp16 or args.fp16_true) else _lib.lib().enerf_mlp32_precision(-1)
            if mode != 0:                                 # (split kernels: timed by their own stamps, the reduce launch apart;
                bwd_ms += red_ms                          # the fp32 MFMA route's interval already spans its reduce)
            gb_ms, gb_n = _lib.prof.read("grid_bwd")
            ta_ms, ta_n = _lib.prof.read("table_adam")
            gb_calls = gb.STATS["bwd_calls"] - bwd_before[1]
            if roofline is not None and gb_n and gb_calls:
                ptsb = (gb.STATS["bwd_points"] - bwd_before[0]) / gb_calls
                if fill is not None and 0 < probe_samples < ptsb:
                    ptsb = probe_samples              # (real samples of the probe steps, not their budget rows)
                kernels["grid_bwd"] = {"avg_ms": gb_ms / gb_n, "launches": int(gb_n), "timed_in": "probe steps"}
                if ta_n:
                    kernels["table_adam"] = {"avg_ms": ta_ms / ta_n, "launches": int(ta_n), "timed_in": "probe steps"}
                roofline["table_backward"] = table_backward_object(
                    ptsb, gb_ms / gb_n, ta_ms / ta_n if ta_n else None, n_table_params,
                    harness.fuse_table_adam and world == 1, f"the {args.probe_steps} probe steps after the timed region")
            if nf and nb:
                per_step_ms = (fwd_ms + bwd_ms) / args.probe_steps
                flop_step = MLP_LINEAR_FLOP_STEP if args.net == "linear" else 3 * FFMLP_FLOP_FWD
                tf = probe_samples * flop_step / (per_step_ms * 1e-3) / 1e12
                issued = 3.0 if mode == 1 else 1.0          # MFMA products issued per algorithmic fp32 product
                # headline = the pipe the kernels actually run on: split-bf16 issues three bf16 MFMA products per fp32
                # product (flips of the weight-gradient tiles not counted), priced against the dense bf16 peak; the
                # algorithmic fp32 FLOP against the fp32 MFMA peak (the reference's arithmetic type) is the secondary view
                on_bf16 = mode in (1, 2)
                peak = MFMA_BF16_PEAK_TF if on_bf16 else MFMA_F32_PEAK_TF
                roofline_mfma = {"bound": "mfma", "kernel": "mlp32 sigma + colour nets, forward + fused dgrad/wgrad ("
                                 + ("v_mfma_f32_32x32x2_f32: fp32 fmaf chains" if mode == 0 else
                                    "v_mfma_f32_32x32x16_bf16 on split operands: hi*hi + hi*lo + lo*hi per fp32 product, "
                                    "fp32 accumulation" if mode == 1 else "v_mfma_f32_32x32x16_bf16, bf16 operands") + ")",
                                 "achieved": tf * issued, "peak": peak, "unit": "TFLOP/s", "frac": tf * issued / peak,
                                 "peak_note": ("flops ISSUED on the bf16 matrix pipe (3 products per algorithmic fp32 product) "
                                               "against the dense bf16 MFMA peak" if mode == 1 else
                                               "bf16 nets: FLOP of the nets against the dense bf16 MFMA peak" if mode == 2 else
                                               "fp32 FLOP of the nn.Linear nets against the dense fp32 MFMA peak"),
                                 "products_per_fp32_product": issued,
                                 "algorithmic_fp32": None if mode != 1 else {
                                     "achieved": tf, "peak": MFMA_F32_PEAK_TF, "frac": tf / MFMA_F32_PEAK_TF,
                                     "note": "algorithmic fp32 FLOP of the nn.Linear nets against the dense fp32 MFMA peak "
                                             "(what the reference's arithmetic type could reach at best)"},
                                 "mlp32_precision": mode, "flop_per_sample": flop_step,
                                 "samples_per_step": probe_samples, "kernel_ms_per_step": per_step_ms,
                                 "forward_ms_per_step": fwd_ms / args.probe_steps,
                                 "backward_ms_per_step": bwd_ms / args.probe_steps,
                                 "launches": int(nf + nb), "steps": args.probe_steps,
                                 "timing": "kernel-attached events (each MLP kernel's own dispatch-to-end, as rocprofv3 reports it) + the "
                                           "weight-gradient reduce launch between two event packets"}
        finally:
            harness.update_interval = keep_interval

    # ---- other steps (not part of `value`; one GPU): the event step of BASELINE configs[2] (two renders per step, rays
    # drawn from the event stream) and the training step of nerf/network_ff.py (FFMLP nets), each on a model of its own,
    # timed like the main region (warm-up, then steps between two synchronisations), so that they are measured by
    # whoever runs this file and not only quoted in DESIGN.md
    other_steps = None
    if world == 1 and args.other_legs > 0 and args.mode == "rgb" and args.net == "linear" and not args.fp16 \
            and not args.fp16_autocast and not args.fp16_true and not args.graphs:
        other_steps = {}
        # "after_step_256_rgb": the headline's step once the density grid has had its 16 full sweeps (renderer.py:484:
        # update_extra_state then samples 128^3 / 4 points per cascade instead of sweeping all of them) -- the regime a
        # training run of thousands of steps spends its time in; the headline's K steps all fall in the full-sweep phase
        for tag, net_kind, mode, bound, fp16, iter_density in (("events_configs2", "linear", "events", 2, False, 0),
                                                               ("network_ff_rgb", "ff", "rgb", args.bound, False, 0),
                                                               ("amp_bf16_rgb", "linear", "rgb", args.bound, "bf16", 0),
                                                               ("fp16_true_rgb", "linear", "rgb", args.bound, True, 0),
                                                               ("fp16_autocast_rgb", "linear", "rgb", args.bound, "autocast", 0),
                                                               ("after_step_256_rgb", "linear", "rgb", args.bound, False, 16),
                                                               ("mlp32_fp32_exact_rgb", "linear", "rgb", args.bound, False, 0),
                                                               ("dropin_route_rgb", "linear", "rgb", args.bound, False, 0),
                                                               ("dropin_route_fusedadam_rgb", "linear", "rgb", args.bound, False, 0)):
            restore = []
            if args.only_legs and tag not in args.only_legs.split(","):
                continue
            try:
                if tag == "mlp32_fp32_exact_rgb":
                    # the headline's step with the nn.Linear nets on v_mfma_f32_32x32x2_f32 (bit-comparable fmaf chains)
                    # instead of the default split-bf16 products
                    prev_prec = _lib.lib().enerf_mlp32_precision(0)
                    restore.append(lambda p_=prev_prec: _lib.lib().enerf_mlp32_precision(p_))
                if tag.startswith("dropin_route"):
                    # the boundary as the reference uses it: `_backend` of the reference-shaped wrappers = the pybind11
                    # modules _raymarching / _gridencoder / _shencoder (enerf_amd/ext), every fused route off, autograd,
                    # torch.optim.Adam -- what tests/test_gpu_ext.py checks against the oracle, timed
                    import importlib
                    import enerf_amd.raymarching as rmod, enerf_amd.gridencoder as gmod, enerf_amd.shencoder as smod
                    from enerf_amd import ext as ext_pkg, fused_network as fn_, fused_render as fr_, density_update as du_
                    ext_pkg.activate()
                    mods = [importlib.import_module(n) for n in ("_raymarching", "_gridencoder", "_shencoder")]
                    for obj, name, val in ((rmod, "_backend", mods[0]), (gmod, "_backend", mods[1]),
                                           (smod, "_backend", mods[2]), (fr_, "ENABLED", False), (fn_, "ENABLED", False),
                                           (du_, "ENABLED", False), (gmod, "_layout_support", {})):
                        restore.append(lambda o_=obj, n_=name, v_=getattr(obj, name): setattr(o_, n_, v_))
                        setattr(obj, name, val)
                if net_kind == "ff":
                    from enerf_amd.network_ff import NeRFNetwork as LegNet
                else:
                    from enerf_amd.network import NeRFNetwork as LegNet
                torch.manual_seed(0)
                m2 = LegNet(encoding="hashgrid", bound=bound, cuda_ray=True, out_dim_color=3).to(device)
                if tag.startswith("dropin_route"):
                    # ("_fusedadam": the same route with the one-line optimizer swap of INTEGRATION.md -- enerf_amd.optim.FusedAdam
                    #  in place of torch.optim.Adam: one launch over all parameters instead of torch's multi-tensor passes)
                    h2 = TrainHarness(m2, occupancy="synthetic", world=1,
                                      optimizer=None if tag.endswith("fusedadam_rgb") else torch.optim.Adam)
                    h2.native_step = h2.manual_mse = h2.fuse_table_adam = h2.prefetch = False
                else:
                    h2 = TrainHarness(m2, occupancy="synthetic", world=1, fp16=fp16 if fp16 in (True, "autocast") else False,
                                      amp="bf16" if fp16 == "bf16" else None)
                m2.iter_density = iter_density
                b2 = batches if bound == args.bound else build_batches(8, args.rays, device, rank, bound)

                def leg_step(i):
                    if mode == "rgb":
                        nxt = b2[(i + 1) % len(b2)]
                        return h2.step_rgb(*b2[i % len(b2)], next_rays=(nxt[0], nxt[1]))
                    return h2.step_events(event_data(i), ev_opt, next_data=event_data(i + 1))
                for i in range(20):
                    leg_step(i)
                sync()
                tl0 = time.perf_counter()
                for i in range(20, 20 + args.other_legs):
                    leg_step(i)
                sync()
                dt = (time.perf_counter() - tl0) / args.other_legs
                renders = 2 if mode == "events" else 1
                other_steps[tag] = {"ms_per_step": dt * 1e3, "rays_per_sec": args.rays * renders / dt,
                                    "steps": args.other_legs, "warmup": 20, "bound": bound, "net": net_kind, "mode": mode,
                                    "renders_per_step": renders, "fp16": fp16,
                                    "includes_update_extra_state_steps": args.other_legs // 16,
                                    "update_extra_state": "partial (iter_density >= 16)" if iter_density >= 16 else "full sweep"}
                del m2, h2
            except Exception as e:          # a leg that breaks must not take the headline down with it
                import traceback
                traceback.print_exc(file=sys.stderr)
                other_steps[tag] = {"error": repr(e)[:300], "where": traceback.format_exc()[-700:]}
            finally:
                for undo in reversed(restore):
                    undo()

    # ---- `strong`: BASELINE configs[3] -- 65 536 rays per step over ALL ranks (8192 per rank at N = 8), the same step on a
    # model of its own with the main region's communication settings, timed like the main region (barrier + synchronise
    # on both sides, MAX over ranks).  One launch of this file at N = 1, 2, 4, 8 therefore yields the weak curve (`value`)
    # AND the strong curve of configs[3] (`strong.value`).
    strong = None
    if args.strong_rays > 0 and not args.global_rays and args.mode == "rgb" and args.net == "linear" and not args.fp16 \
            and not args.fp16_autocast and not args.fp16_true and not args.graphs and args.strong_rays % world == 0:
        try:
            rays_s = args.strong_rays // world
            torch.manual_seed(0)
            m3 = NeRFNetwork(encoding="hashgrid", bound=args.bound, cuda_ray=True, out_dim_color=3).to(device)
            h3 = TrainHarness(m3, occupancy="synthetic", world=world)
            for k in ("prefetch", "early_budget", "prefetch_at", "comm_dtype", "comm_chunks", "comm_mode"):
                setattr(h3, k, getattr(harness, k))
            parallel.broadcast_state(m3)
            b3 = build_batches(4, rays_s, device, rank, args.bound)

            def strong_step(i):
                nxt = b3[(i + 1) % len(b3)]
                return h3.step_rgb(*b3[i % len(b3)], next_rays=(nxt[0], nxt[1]))
            for i in range(20):
                strong_step(i)
            sync()
            before = marched_total(reset=False) - sum(
                int(m3.step_counter[p_["slot"], 0]) for p_ in (getattr(m3, "_premarched", None) or {}).values()
                if p_.get("slot") is not None)
            sync()
            ts0 = time.perf_counter()
            for i in range(20, 20 + args.strong_steps):
                strong_step(i)
            ts_enq = time.perf_counter()
            sync()
            ts = time.perf_counter() - ts0
            after = marched_total(reset=False) - sum(
                int(m3.step_counter[p_["slot"], 0]) for p_ in (getattr(m3, "_premarched", None) or {}).values()
                if p_.get("slot") is not None)
            st = torch.tensor([ts, float(after - before)], dtype=torch.float64, device=device)
            if world > 1:
                t_max = st[:1].clone()
                dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
                dist.all_reduce(st[1:], op=dist.ReduceOp.SUM)
                st[0] = t_max[0]
            ts, smp = float(st[0]), float(st[1])
            strong = {"workload": f"BASELINE configs[3]: {args.strong_rays} rays/step ray-sharded over {world} rank(s) = "
                                  f"{rays_s} rays/GPU, bound={args.bound}, nn.Linear nets, RCCL gradient exchange",
                      "scaling": "strong", "global_rays": args.strong_rays, "rays_per_gpu": rays_s, "n_gpus": world,
                      "value": args.strong_rays * args.strong_steps / ts, "unit": "rays/s",
                      "ms_per_step": ts / args.strong_steps * 1e3, "steps": args.strong_steps, "warmup": 20,
                      "train_ray_samples_per_sec": smp / ts,
                      "host_enqueue_ms_per_step": (ts_enq - ts0) / args.strong_steps * 1e3,
                      "includes_update_extra_state_steps": sum(1 for i in range(20, 20 + args.strong_steps) if i % 16 == 0),
                      "tail": (h3.comm_mode + " torch.distributed") if world > 1 else None}
            del m3, h3, b3
        except Exception as e:                  # the headline must survive a leg that breaks
            strong = {"error": repr(e)[:300]}

    # ---- render leg (not part of `value`): full 640x480 frame, pixels sharded over the ranks + all_gather of the tiles
    # (SURVEY.md 8e).  msamples_per_sec is the whole job's: samples marched on all ranks / slowest rank's time.
    render = None
    if args.render_frames > 0:
        from enerf_amd import scene

        def time_frames(net):
            inds = torch.arange(scene.H * scene.W, device=device)
            ro, rd = scene.pixel_rays(scene.pose(3), inds, device)
            net.eval()
            with torch.no_grad():
                parallel.render_sharded(net, ro, rd, bg_color=None, perturb=False)      # warm-up
                sync()
                rb.STATS.update(infer_samples=0, infer_calls=0)
                _lib.prof.reset()
                _lib.prof.enable(True, only=("ffmlp_fwd",))
                tr0 = time.perf_counter()
                for _ in range(args.render_frames):
                    out = parallel.render_sharded(net, ro, rd, bg_color=None, perturb=False)
                sync()
                tr = time.perf_counter() - tr0
                _lib.prof.enable(False)
            mlp_ms, mlp_n = _lib.prof.read("ffmlp_fwd")
            stat = torch.tensor([tr, float(rb.STATS["infer_samples"])], dtype=torch.float64, device=device)
            if world > 1:
                t_max = stat[:1].clone()
                dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
                dist.all_reduce(stat[1:], op=dist.ReduceOp.SUM)
                stat[0] = t_max[0]
            tr, samples = float(stat[0]), float(stat[1])
            assert out["image"].shape == (1, scene.H * scene.W, 3)
            return {"msamples_per_sec": samples / tr / 1e6, "ms_per_frame": tr / args.render_frames * 1e3,
                    "samples_per_frame": samples / args.render_frames,
                    "launch_rounds_per_frame_per_rank": rb.STATS["infer_calls"] / args.render_frames}, \
                (mlp_ms / args.render_frames, rb.STATS["infer_samples"] / args.render_frames)

        r, _ = time_frames(model)
        render = dict(r, frames=args.render_frames, net=args.net, rays_per_frame=scene.H * scene.W,
                      schedule="whole frame: march all samples once, one compositing pass (enerf_amd/frame.py)",
                      sharding=f"pixels over {world} rank(s) + all_gather of the tiles")
        model.train()
        if args.net != "ff":
            # BASELINE configs[4]: the same frame through the fully-fused (bf16 MFMA) networks of nerf/network_ff.py
            from enerf_amd.network_ff import NeRFNetwork as FFNet
            torch.manual_seed(0)
            ffm = FFNet(encoding="hashgrid", bound=args.bound, cuda_ray=True).to(device).eval()
            scene.install_occupancy(ffm)
            ffm.infer_batch_mult = 8
            r, (mlp_ms, mlp_samples) = time_frames(ffm)
            render["ffmlp_nets"] = dict(r, dtype="bf16")
            if mlp_ms > 0:
                tf = mlp_samples * FFMLP_FLOP_FWD / (mlp_ms * 1e-3) / 1e12
                render["ffmlp_nets"]["roofline_mfma"] = {
                    "bound": "mfma", "kernel": "k_ffnerf_infer (both FFMLP nets of network_ff in one launch, bf16 MFMA; "
                    "this rank)", "achieved": tf, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                    "frac": tf / MFMA_BF16_PEAK_TF, "flop_per_sample": FFMLP_FLOP_FWD,
                    "samples_per_frame": mlp_samples, "kernel_ms_per_frame": mlp_ms}
            del ffm

    # ---- the reference's own fully-fused MLP entry point (a16, ffmlp_inference) at the frame's sample count: colour net
    # 32 -> 64 x 3 -> 16, bf16 storage, fp32 MFMA accumulation.  north_star's "MFMA utilisation for ffmlp" is about this
    # kernel; the frame leg above uses the larger fusion (k_ffnerf_infer: both nets + SH + activations, VALU-bound).
    ffmlp_kernel = None
    if args.render_frames > 0 and rank == 0:
        from enerf_amd.backends import _ffmlp as ffb
        Bk, k = 2 * 1024 * 1024, 3       # 192 MB of operands: cache-resident between launches (8 M samples: 0.33, HBM-fed)
        W = ((torch.rand(64 * (32 + 64 * (k - 1) + 16), device=device) - 0.5) * 0.5).to(torch.bfloat16)
        x = (torch.rand(Bk, 32, device=device) - 0.5).to(torch.bfloat16)
        y = torch.empty(Bk, 16, device=device, dtype=torch.bfloat16)
        ib = torch.empty(Bk, 64, device=device, dtype=torch.bfloat16)
        for _ in range(3):
            ffb.ffmlp_inference(x, W, Bk, 32, 16, 64, k, 0, 6, ib, y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ffb.ffmlp_inference(x, W, Bk, 32, 16, 64, k, 0, 6, ib, y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        flop = 2.0 * Bk * (32 * 64 + (k - 1) * 64 * 64 + 64 * 16)
        tf = flop / (ms * 1e-3) / 1e12
        ffmlp_kernel = {"bound": "mfma", "kernel": "ffmlp_inference, colour net 32->64x3->16, bf16 (k_ffmlp_fwd)",
                        "achieved": tf, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": tf / MFMA_BF16_PEAK_TF,
                        "samples": Bk, "flop_per_sample": flop / Bk, "avg_launch_ms": ms, "launches": 20}
        del W, x, y, ib

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # in a fresh, unpinned process: this one's thread pools were created under the 8-core pin
        import subprocess
        env = dict(os.environ)
        env.pop("HIP_VISIBLE_DEVICES", None)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-rays",
                            str(args.cpu_rays), "--cpu-budget-s", str(args.cpu_budget_s), "--bound", str(args.bound),
                            "--cpu-threads", str(args.cpu_threads)],
                           capture_output=True, text=True, env=env)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        cpu = json.loads(lines[-1]) if lines else {"error": (r.stderr or r.stdout)[-400:]}

    # who took part: RCCL's own view of the world and the device every rank ran on, so that a multi-GPU line is
    # self-evidently N ranks on N distinct GPUs
    props = torch.cuda.get_device_properties(device)
    me = {"rank": rank, "local_rank": local_rank, "device_index": torch.cuda.current_device(), "device_name": props.name,
          "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", "")) or None,
          "pid": os.getpid()}
    if world > 1:
        members = [None] * world
        dist.all_gather_object(members, me)
        world_obj = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks": members,
                     "distinct_devices": len({(m["device_index"], m["pci_bus_id"], m["uuid"]) for m in members})}
    else:
        world_obj = {"backend": None, "world_size": 1, "ranks": [me], "distinct_devices": 1}
    if rank == 0:
        out = {
            "metric": "train_rays_per_sec",
            "value": total_rays / elapsed,
            "unit": "rays/s",
            "n_gpus": world,
            "world": world_obj,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if args.global_rays else "weak",
            "vs_baseline": None,
            "dtype": "f16-autocast" if args.fp16_autocast else ("f16-amp" if args.fp16_true else (
                "bf16-amp" if args.fp16 else ("bf16" if args.net == "ff" else "f32"))),
            "data": "synthetic",
            "arithmetic": None if (args.net != "linear" or args.fp16 or args.fp16_autocast or args.fp16_true) else (
                "fp32 storage and accumulation; products of the nn.Linear nets as three bf16 MFMA terms per fp32 product "
                "(split-bf16: ~5e-6 relative forward error, profiles/r03_mlp32_accuracy.txt; both nets in one launch per "
                "direction, csrc/nerf_mlp.hip); the same step on the exact fp32 MFMA is value_exact_fp32 / "
                "other_steps.mlp32_fp32_exact_rgb" if _lib.lib().enerf_mlp32_precision(-1) == 1 else
                "fp32 MFMA (v_mfma_f32_32x32x2_f32)"),
            "config": {"workload": (f"BASELINE configs[3]: {args.global_rays} rays/step ray-sharded over {world} rank(s), "
                                    if args.global_rays else "BASELINE configs[1]: ") +
                                   f"shakeCarpet1-shaped train step, bound={args.bound}, hashgrid "
                                   f"L16 F2 T2^19 + HIP march_rays_train, {'nn.Linear MLPs fp32' if args.net == 'linear' else 'FFMLP bf16'}, {args.rays} rays/GPU, "
                                   f"mode={args.mode}",
                       "rays_per_gpu": args.rays, "global_rays": world * args.rays,
                       "parallelism": f"ray-sharded dp{world}" if world > 1 else "single"},
            "train_ray_samples_per_sec": total_samples / elapsed,
            "samples_per_step_per_gpu": total_samples / args.steps / world,
            # the strictly-fp32 figure beside `value` (the same step on v_mfma_f32_32x32x2_f32, other_steps.mlp32_fp32_exact_rgb:
            # 48 steps of its own after 20 warm-up steps -- three density-grid updates in the window, not one in twenty)
            "value_exact_fp32": (other_steps or {}).get("mlp32_fp32_exact_rgb", {}).get("rays_per_sec"),
            "render": render,
            "step_split": split,
            "other_steps": other_steps,
            "strong": strong,
            # every grid_encode_forward launch of this process (all legs): the population a whole-process counter profile
            # averages over (tools/profile_round.sh divides its per-dispatch FETCH / WRITE averages by this)
            "grid_fwd_lifetime": {"launches": gb.LIFETIME["fwd_calls"],
                                  "points_per_launch": gb.LIFETIME["fwd_points"] / max(gb.LIFETIME["fwd_calls"], 1)},
            "host_enqueue_ms_per_step": (t_enqueued - t0) / args.steps * 1e3,
            "roofline_mfma": roofline_mfma,
            "roofline_mfma_ffmlp": ffmlp_kernel,
            "comm_tuning": comm_tuning,
            "kernels": kernels,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "gpu_reference_route": reference_route_on_file(),
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
