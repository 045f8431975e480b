"""Does the MI355X-first training path train to the same quality as the reference-shaped route?

A = what bench.py times: closed-form MSE step, one fused render node, fused fp32 MFMA networks, device-side
    update_extra_state, side-stream march, fused Adam.
B = the reference's structure over the same HIP entry points: run_cuda op by op through the four autograd Functions
    (raymarching.py / grid.py / sphere_harmonics.py), nn.Linear networks, F.mse_loss + autograd, the Python
    update_extra_state (nerf/renderer.py:472-560), torch.optim.Adam.

Same model seed, same batches, the occupancy grid learned by update_extra_state itself.  Reports held-out PSNR
(16 K pixels never trained on) for several seeds of each; writes a JSON summary.
python tools/psnr_ab.py [steps] [seeds] [out.json] [first_seed]

tests/refcheck/psnr_vs_reference_kernels.py runs the same experiment with route B's ray marching, compositing (forward and
backward) and SH encoding on the REFERENCE's own kernels (its raymarching.cu / shencoder.cu built for gfx950, oracle/_ref),
bound to the wrappers exactly as the reference's raymarching.py / sphere_harmonics.py bind them; only the hash grid and the
MLPs of route B stay on this library (gridencoder.cu / ffmlp cannot be built here)."""
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from enerf_amd import density_update, fused_mlp, fused_network, fused_render  # noqa: E402
from enerf_amd.network import NeRFNetwork  # noqa: E402
from enerf_amd.trainer import TrainHarness  # noqa: E402
from test_gpu_training import _batches  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
out_path = sys.argv[3] if len(sys.argv) > 3 else None
first_seed = int(sys.argv[4]) if len(sys.argv) > 4 else 0
# ROUTE_B_BACKENDS = (raymarching module, shencoder module[, gridencoder module]): set by
# tests/refcheck/psnr_vs_reference_kernels.py, which runs this file with the reference's own kernels bound to route B's
# wrappers (this tool itself never touches oracle/).  With all three, route B is the reference's native code end to end:
# its nets are nn.Linear (torch's GEMMs, as in the reference's nerf/network.py), its optimizer torch.optim.Adam.
_ref = globals().get("ROUTE_B_BACKENDS")
ref_kernels = _ref is not None
_own = None
if ref_kernels:
    import enerf_amd.raymarching as _rmod
    import enerf_amd.shencoder as _smod
    import enerf_amd.gridencoder as _gmod
    _own = (_rmod._backend, _smod._backend, _gmod._backend)
    _ref = tuple(_ref) + (_gmod._backend,) * (3 - len(_ref))
SHUFFLE = os.environ.get("ENERF_PSNR_SHUFFLE", "0") == "1"
# ENERF_PSNR_NO_DROPS=1: the sample budget of both routes is three times the mean count of the window, so that no ray is
# ever dropped for lack of room (which rays a budget drops is the one thing the two marchers do differently: the reference
# whatever its atomics' order gives, this library the last rays of the batch)
NO_DROPS = os.environ.get("ENERF_PSNR_NO_DROPS", "0") == "1"
data = _batches(32, 4096, 2, seed=5)
held = _batches(1, 16384, 2, seed=77)[0]


def _route(fused):
    fused_render.ENABLED = fused_network.ENABLED = density_update.ENABLED = fused
    # route B entirely on the reference's native code (three backends given): its nets as the plain Linear / ReLU loop on
    # torch's GEMMs too; otherwise route B's nn.Linear nets run on this library's MLP kernels behind autograd
    fused_mlp.ENABLED = fused or not (_own is not None and _ref[2] is not _own[2])
    if _own is not None:
        _rmod._backend, _smod._backend, _gmod._backend = _own if fused else _ref


def run(route, seed):
    fused = route == "A"
    _route(fused)
    torch.manual_seed(seed)
    model = NeRFNetwork(encoding="hashgrid", bound=2, cuda_ray=True, out_dim_color=3).cuda()
    h = TrainHarness(model, lr=1e-2, occupancy="learned")
    h.manual_mse = h.prefetch = fused
    if not fused:
        h.opt = torch.optim.Adam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
        h._params = [p for g in h.opt.param_groups for p in g["params"]]
        h._opt_step = h.opt.step
    if NO_DROPS:
        inner = h.maybe_update_extra_state

        def generous(*a, **k):
            before = model.iter_density
            inner(*a, **k)
            if model.iter_density != before and model.mean_count > 0:
                model.mean_count = int(model.mean_count) * 3
        h.maybe_update_extra_state = generous
    torch.cuda.synchronize()
    t0 = time.time()
    psnrs = []
    gperm = torch.Generator(device="cuda").manual_seed(10_000 + seed)

    def batch(i):
        """ENERF_PSNR_SHUFFLE=1: the rays of a batch in a fresh order every time it is used (the same order for both
        routes).  With a sample budget the marcher drops the rays that do not fit: the reference in the order its atomics
        land, this library the LAST rays of the batch -- with 32 fixed batches cycled in a fixed order those are always
        the same pixels, which a data loader that draws fresh rays every step never produces."""
        ro, rd, tg = data[i % len(data)]
        if not SHUFFLE:
            return ro, rd, tg
        if i not in cache:
            for k in [k for k in cache if k < i - 1]:
                del cache[k]
            p = torch.randperm(ro.shape[-2], device=ro.device, generator=gperm)
            cache[i] = (ro[..., p, :].contiguous(), rd[..., p, :].contiguous(), tg[..., p, :].contiguous())
        return cache[i]
    cache = {}
    eval_s = 0.0
    for i in range(steps):
        cur = batch(i)
        nxt = batch(i + 1)
        loss = h.step_rgb(*cur, next_rays=(nxt[0], nxt[1]) if fused else None)
        if i + 1 > steps - steps // 5 and (steps - 1 - i) % (steps // 25) == 0:      # 5 evaluations over the last fifth
            # the held-out render is the measuring instrument, not the route under test: both routes' models are rendered
            # by the same (product) renderer
            torch.cuda.synchronize()
            te = time.time()
            _route(True)
            model.eval()
            with torch.no_grad():
                img = model.render(held[0], held[1], staged=False, bg_color=None, perturb=False)["image"].reshape(-1, 3)
            model.train()
            _route(fused)
            torch.cuda.synchronize()
            eval_s += time.time() - te
            psnrs.append(-10 * math.log10(float(((img - held[2]) ** 2).mean())))
    torch.cuda.synchronize()
    ms = 1e3 * (time.time() - t0 - eval_s) / steps
    psnr = sum(psnrs) / len(psnrs)
    occ = float((model.density_grid > min(model.mean_density, model.density_thresh)).float().mean())
    return {"route": route, "seed": seed, "psnr_db": psnr, "final_loss": float(loss), "occupied_frac": occ,
            "ms_per_step": ms, "eval_seconds": eval_s, "evaluations": len(psnrs), "samples_per_step": int(model.mean_count)}


rows = []
for seed in range(first_seed, first_seed + seeds):
    for route in ("A", "B"):
        r = run(route, seed)
        rows.append(r)
        print(route, seed, round(r["psnr_db"], 3), flush=True)
fused_render.ENABLED = fused_network.ENABLED = density_update.ENABLED = fused_mlp.ENABLED = True
if _own is not None:
    _rmod._backend, _smod._backend, _gmod._backend = _own
mean = {k: sum(r["psnr_db"] for r in rows if r["route"] == k) / seeds for k in ("A", "B")}
spread = {k: max(r["psnr_db"] for r in rows if r["route"] == k) - min(r["psnr_db"] for r in rows if r["route"] == k)
          for k in ("A", "B")}
diffs = [a["psnr_db"] - b["psnr_db"] for a, b in zip(rows[0::2], rows[1::2])]           # paired by seed
dmean = sum(diffs) / len(diffs)
dstd = (sum((d - dmean) ** 2 for d in diffs) / max(len(diffs) - 1, 1)) ** 0.5
summary = {"steps": steps, "seeds": seeds, "rays_reshuffled_per_use": SHUFFLE, "no_budget_drops": NO_DROPS, "route_B_kernels": ("reference raymarching.cu + shencoder.cu" + (" + gridencoder.cu, nets on torch's GEMMs" if _ref[2] is not _own[2] else "") + " (oracle/_ref)") if ref_kernels else "this library", "mean_psnr_db": mean, "seed_spread_db": spread,
           "A_minus_B_db": dmean, "paired_std_db": dstd, "standard_error_db": dstd / len(diffs) ** 0.5,
           "runs": rows}
print(json.dumps({k: v for k, v in summary.items() if k != "runs"}))
if out_path:
    with open(out_path, "w") as f:
        json.dump(summary, f, indent=1)
