"""Event-only training (nerf/utils.py:482-573: two renders per step, one loss on their (lin-)log difference): does the
default arithmetic of the fast path train to the same quality as exact fp32 and as the reference-shaped route?

The event loss differentiates a DIFFERENCE of two renders of nearly the same rays, so a weight gradient is what is left of two
contributions ~40x its size that cancel; in split-bf16 arithmetic its error is 5.5e-3 of the largest entry against 4.4e-6 on
the fp32 MFMA kernels (tests/test_gpu_cuda_ray_vs_reference_fixture.py).  tools/psnr_ab.py answers the quality question for
the RGB-MSE step only; this file answers it for the loss E-NeRF actually trains with.

Arms (same model seed, same event batches, the occupancy grid learned by update_extra_state itself):
  A = what bench.py's events leg times: the one-call event step, both nets one launch per direction, products as three bf16
      MFMA terms (the default), fused Adam, device-side update_extra_state, side-stream march
  X = the same route with model.mlp_precision = 0 (enerf_mlp32_precision(0): fp32 MFMA products, bit-comparable arithmetic)
  B = the reference's structure: run_cuda op by op through the autograd Functions, nn.Linear nets, event_loss + autograd, the
      Python update_extra_state, torch.optim.Adam.  With ROUTE_B_BACKENDS set (tests/refcheck/psnr_events_vs_reference_kernels.py)
      its marching / compositing / SH / hash-grid kernels are the REFERENCE's own (oracle/_ref) and its nets torch's GEMMs.

Synthetic events: the teacher is the analytic scene of tools/psnr_ab.py (colour of the point where the ray meets the
0.6-sphere, white elsewhere); a step's batch is `rays` random pixels seen from two poses `DELTA_DEG` apart along the camera
circle, and `pols` the accumulated polarity the teacher's two images imply, (linlog(luma2*255) - linlog(luma1*255)) / C_thres
(accumulate_evs = 1: a real-valued event count, so that the teacher itself has zero loss).

Measured on 16 K held-out pixel pairs (two poses never trained on), both by the product's renderer whatever the arm:
  * `event_db`  = -10 log10 of the held-out event loss mean((delta - pols*C)^2): the quantity being trained, on unseen pairs
  * `psnr_db`   = PSNR of the predicted lin-log luma against the teacher's after the best affine fit a*x + b (event-only
                  supervision fixes intensity only up to that gauge), in units of linlog(255)
and the training-loss curve (mean over windows of steps/50).
    python tools/psnr_ab_events.py [steps] [seeds] [out.json] [first_seed] [arms, e.g. AXB]"""
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from enerf_amd import density_update, fused_mlp, fused_network, fused_render, scene  # noqa: E402
from enerf_amd.events import EventOptions, lin_log, rgb_to_luma  # noqa: E402
from enerf_amd.network import NeRFNetwork  # noqa: E402
from enerf_amd.trainer import TrainHarness  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
out_path = sys.argv[3] if len(sys.argv) > 3 else None
first_seed = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ARMS = sys.argv[5] if len(sys.argv) > 5 else "AXB"
RAYS = int(os.environ.get("ENERF_AB_RAYS", "4096"))
DELTA_DEG = float(os.environ.get("ENERF_AB_DELTA_DEG", "1.0"))
C_THRES = 0.2
DEV = "cuda"

_ref = globals().get("ROUTE_B_BACKENDS")
ref_kernels = _ref is not None
_own = None
if ref_kernels:
    import enerf_amd.raymarching as _rmod
    import enerf_amd.shencoder as _smod
    import enerf_amd.gridencoder as _gmod
    _own = (_rmod._backend, _smod._backend, _gmod._backend)
    _ref = tuple(_ref) + (_gmod._backend,) * (3 - len(_ref))


def teacher(ro, rd):
    b_ = (ro * rd).sum(-1)
    disc = b_ ** 2 - ((ro * ro).sum(-1) - 0.36)
    hit = disc > 0
    t = -b_ - torch.sqrt(disc.clamp(min=0))
    p = ro + rd * t.unsqueeze(-1)
    return torch.where(hit.unsqueeze(-1), scene.analytic_color(p).clamp(0, 1), torch.ones_like(p)), hit


def linlog_luma(img):
    return lin_log(rgb_to_luma(img, esim=True) * 255, linlog_thres=20)


def pair(k, n_rays, gen):
    """One event batch: `n_rays` pixels seen from pose k and from the pose DELTA_DEG further along the circle."""
    inds = torch.randint(0, scene.H * scene.W, (n_rays,), device=DEV, generator=gen)
    o1, d1 = scene.pixel_rays(scene.pose(k), inds, DEV)
    o2, d2 = scene.pixel_rays(scene.pose(k + DELTA_DEG / (360.0 / 32)), inds, DEV)
    t1, h1 = teacher(o1, d1)
    t2, h2 = teacher(o2, d2)
    pols = ((linlog_luma(t2) - linlog_luma(t1)) / C_THRES).reshape(1, n_rays).contiguous()
    return {"images": torch.zeros(1, n_rays, 3, device=DEV), "rays_evs_o1": o1, "rays_evs_d1": d1, "rays_evs_o2": o2,
            "rays_evs_d2": d2, "pols": pols, "_t1": t1, "_hit": h1 & h2}


_g = torch.Generator(device=DEV).manual_seed(5)
data = [pair((b * 7) % 32, RAYS, _g) for b in range(32)]
_g = torch.Generator(device=DEV).manual_seed(77)
held = pair(10.37, 16384, _g)                      # (a pose pair between the training poses)


def _route(fused):
    fused_render.ENABLED = fused_network.ENABLED = density_update.ENABLED = fused
    fused_mlp.ENABLED = fused or not (_own is not None and _ref[2] is not _own[2])
    if _own is not None:
        _rmod._backend, _smod._backend, _gmod._backend = _own if fused else _ref


def evaluate(model):
    """Held-out metrics, by the product's renderer in its default arithmetic whatever trained the model."""
    prec = model.__dict__.pop("mlp_precision", None)
    _route(True)
    model.eval()
    with torch.no_grad():
        i1 = model.render(held["rays_evs_o1"], held["rays_evs_d1"], staged=False, bg_color=None, perturb=False)["image"]
        i2 = model.render(held["rays_evs_o2"], held["rays_evs_d2"], staged=False, bg_color=None, perturb=False)["image"]
    model.train()
    if prec is not None:
        model.mlp_precision = prec
    p1, p2 = linlog_luma(i1.reshape(1, -1, 3)).reshape(-1), linlog_luma(i2.reshape(1, -1, 3)).reshape(-1)
    ev = float((((p2 - p1) - held["pols"].reshape(-1) * C_THRES) ** 2).mean())
    t = linlog_luma(held["_t1"]).reshape(-1)
    x, y = p1.double(), t.double()
    vx = ((x - x.mean()) ** 2).mean()
    a = (((x - x.mean()) * (y - y.mean())).mean() / vx) if float(vx) > 0 else torch.zeros((), dtype=torch.float64)
    b = y.mean() - a * x.mean()
    mse = float(((a * x + b - y) ** 2).mean()) / math.log(255.0) ** 2
    return -10 * math.log10(max(ev, 1e-30)), -10 * math.log10(max(mse, 1e-30))


def run(arm, seed):
    fused = arm in "AX"
    _route(fused)
    torch.manual_seed(seed)
    model = NeRFNetwork(encoding="hashgrid", bound=2, cuda_ray=True, out_dim_color=3).cuda()
    if arm == "X":
        model.mlp_precision = 0
    h = TrainHarness(model, lr=1e-2, occupancy="learned")
    h.manual_mse = h.prefetch = fused
    if not fused:
        h.opt = torch.optim.Adam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
        h._params = [p for g in h.opt.param_groups for p in g["params"]]
        h._opt_step = h.opt.step
    opt = EventOptions(use_luma=True, linlog=True, C_thres=C_THRES, event_only=True)
    torch.cuda.synchronize()
    t0 = time.time()
    eval_s = 0.0
    evs, psnrs, curve = [], [], []
    win = max(1, steps // 50)
    acc = torch.zeros((), device=DEV)
    every = max(1, steps // 25)
    for i in range(steps):
        cur = data[i % len(data)]
        nxt = data[(i + 1) % len(data)]
        loss = h.step_events(cur, opt, next_data=nxt if fused else None)
        acc += loss.detach().float()
        if (i + 1) % win == 0:
            curve.append(acc / win)
            acc = torch.zeros((), device=DEV)
        if i + 1 > steps - steps // 5 and (steps - 1 - i) % every == 0:          # 5 evaluations over the last fifth
            torch.cuda.synchronize()
            te = time.time()
            e, p = evaluate(model)
            _route(fused)
            torch.cuda.synchronize()
            eval_s += time.time() - te
            evs.append(e)
            psnrs.append(p)
    torch.cuda.synchronize()
    ms = 1e3 * (time.time() - t0 - eval_s) / steps
    return {"arm": arm, "seed": seed, "event_db": sum(evs) / len(evs), "psnr_db": sum(psnrs) / len(psnrs),
            "final_loss": float(loss), "ms_per_step": ms, "evaluations": len(evs),
            "samples_per_step": int(model.mean_count), "loss_curve": [float(c) for c in torch.stack(curve).cpu()]}


def _paired(rows, a, b, key):
    da = {r["seed"]: r[key] for r in rows if r["arm"] == a}
    db = {r["seed"]: r[key] for r in rows if r["arm"] == b}
    d = [da[s] - db[s] for s in sorted(da) if s in db]
    if not d:
        return None
    m = sum(d) / len(d)
    sd = (sum((x - m) ** 2 for x in d) / max(len(d) - 1, 1)) ** 0.5
    return {"pairs": len(d), "mean_diff_db": m, "paired_std_db": sd, "standard_error_db": sd / len(d) ** 0.5}


rows = []
for seed in range(first_seed, first_seed + seeds):
    for arm in ARMS:
        r = run(arm, seed)
        rows.append(r)
        print(arm, seed, f"event_db {r['event_db']:.3f} psnr_db {r['psnr_db']:.3f} loss {r['final_loss']:.3e} "
              f"{r['ms_per_step']:.3f} ms/step", flush=True)
fused_render.ENABLED = fused_network.ENABLED = density_update.ENABLED = fused_mlp.ENABLED = True
if _own is not None:
    _rmod._backend, _smod._backend, _gmod._backend = _own

summary = {"steps": steps, "seeds": seeds, "first_seed": first_seed, "rays": RAYS, "delta_deg": DELTA_DEG, "C_thres": C_THRES,
           "arms": {"A": "one-call event step, split-bf16 products (default)", "X": "same route, fp32 MFMA products",
                    "B": ("reference raymarching.cu + shencoder.cu" + (" + gridencoder.cu, nets on torch's GEMMs"
                                                                       if _ref[2] is not _own[2] else "") + " (oracle/_ref), "
                          if ref_kernels else "this library's entry points, ") +
                         "reference-shaped route: op-by-op autograd, event_loss, Python update_extra_state, torch Adam"},
           "mean": {arm: {k: sum(r[k] for r in rows if r["arm"] == arm) / max(1, sum(1 for r in rows if r["arm"] == arm))
                          for k in ("event_db", "psnr_db", "final_loss", "ms_per_step")} for arm in ARMS},
           "paired": {f"{a}-{b}": {k: _paired(rows, a, b, k) for k in ("event_db", "psnr_db")}
                      for a, b in (("A", "X"), ("A", "B"), ("X", "B")) if a in ARMS and b in ARMS},
           "mean_loss_curve": {arm: [sum(c) / len(c) for c in zip(*[r["loss_curve"] for r in rows if r["arm"] == arm])]
                               for arm in ARMS},
           "curve_window_steps": max(1, steps // 50)}
print(json.dumps({k: v for k, v in summary.items() if k != "mean_loss_curve"}))
summary["runs"] = rows
if out_path:
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(summary, f, indent=1)
