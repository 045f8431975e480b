"""Merge the round's event-loss A/B runs (tools/ab_round6.sh -> gpurun_out/r06/ab/*.json) into one summary:
    python tools/ab_summary.py gpurun_out/r06/ab profiles/r06_event_ab
writes <out>.json (per-arm means, paired differences by seed with standard errors, mean loss curves, every run's scalars)
and <out>.txt (the same as a table, the loss curves overlaid column by column).  Arm R = arm B of
tests/refcheck/psnr_events_vs_reference_kernels.py (the reference's own kernels + torch GEMMs + torch Adam)."""
import glob
import json
import math
import os
import sys

src, out = sys.argv[1], sys.argv[2]
runs, meta = [], {}
for path in sorted(glob.glob(os.path.join(src, "events_*.json"))):
    d = json.load(open(path))
    ref = "oracle/_ref" in d["arms"].get("B", "")
    for r in d["runs"]:
        r = dict(r)
        if r["arm"] == "B" and ref:
            r["arm"] = "R"
        r["file"] = os.path.basename(path)
        runs.append(r)
    meta.update({k: d[k] for k in ("steps", "rays", "delta_deg", "C_thres", "curve_window_steps")})
    if ref:
        meta["arm_R"] = d["arms"]["B"]
    else:
        meta.setdefault("arms", {}).update(d["arms"])
arms = [a for a in "AXBR" if any(r["arm"] == a for r in runs)]


def stats(vals):
    n = len(vals)
    m = sum(vals) / n
    sd = (sum((v - m) ** 2 for v in vals) / max(n - 1, 1)) ** 0.5
    return {"n": n, "mean": m, "std": sd, "standard_error": sd / math.sqrt(n)}


def paired(a, b, key):
    da = {r["seed"]: r[key] for r in runs if r["arm"] == a}
    db = {r["seed"]: r[key] for r in runs if r["arm"] == b}
    d = [da[s] - db[s] for s in sorted(da) if s in db]
    return None if not d else dict(stats(d), pairs=len(d))


summary = dict(meta)
summary["per_arm"] = {a: {k: stats([r[k] for r in runs if r["arm"] == a]) for k in ("event_db", "psnr_db", "final_loss", "ms_per_step",
                                                                                   "samples_per_step")} for a in arms}
summary["paired_by_seed"] = {f"{a}-{b}": {k: paired(a, b, k) for k in ("event_db", "psnr_db")}
                             for a, b in (("A", "X"), ("A", "B"), ("X", "B"), ("A", "R"), ("X", "R"), ("B", "R"))
                             if a in arms and b in arms}
curves = {}
for a in arms:
    cs = [r["loss_curve"] for r in runs if r["arm"] == a]
    curves[a] = [sum(c) / len(c) for c in zip(*cs)]
summary["mean_loss_curve"] = curves
# how far apart the arms' mean curves are, relative, over the second half of training
half = len(curves[arms[0]]) // 2
summary["mean_curve_ratio_second_half"] = {f"{a}/{b}": stats([curves[a][i] / curves[b][i] for i in range(half, len(curves[a]))])["mean"]
                                           for a, b in (("A", "X"), ("A", "B"), ("A", "R")) if a in arms and b in arms}
summary["runs"] = [{k: v for k, v in r.items() if k != "loss_curve"} for r in runs]
os.makedirs(os.path.dirname(os.path.abspath(out)) or ".", exist_ok=True)
json.dump(summary, open(out + ".json", "w"), indent=1)

with open(out + ".txt", "w") as f:
    w = f.write
    w(f"event-only training A/B, {meta['steps']} steps x {meta['rays']} event pairs/step, poses {meta['delta_deg']} deg apart, "
      f"C_thres {meta['C_thres']}; held-out metrics on 16 K pixel pairs (tools/psnr_ab_events.py)\n")
    for a in arms:
        w(f"  {a} = {meta.get('arm_R') if a == 'R' else meta['arms'][a]}\n")
    w("\nper arm (mean +- std over seeds)\n")
    for a in arms:
        p = summary["per_arm"][a]
        w(f"  {a}: n={p['event_db']['n']:3d}  event_db {p['event_db']['mean']:7.3f} +- {p['event_db']['std']:.3f}   psnr_db "
          f"{p['psnr_db']['mean']:7.3f} +- {p['psnr_db']['std']:.3f}   final loss {p['final_loss']['mean']:.3e}   "
          f"{p['ms_per_step']['mean']:.2f} ms/step   {p['samples_per_step']['mean'] / 1e6:.2f} M samples/step\n")
    w("\npaired by seed: mean difference +- standard error [dB] (pairs)\n")
    for name, d in summary["paired_by_seed"].items():
        e, p = d["event_db"], d["psnr_db"]
        w(f"  {name}: event_db {e['mean']:+.3f} +- {e['standard_error']:.3f}   psnr_db {p['mean']:+.3f} +- {p['standard_error']:.3f}"
          f"   ({e['pairs']} pairs; paired std {e['std']:.2f} / {p['std']:.2f})\n")
    w(f"\nmean training-loss curves (window = {meta['curve_window_steps']} steps), one column per arm\n")
    w("  step   " + "".join(f"{a:>12s}" for a in arms) + "\n")
    for i in range(len(curves[arms[0]])):
        w(f"  {(i + 1) * meta['curve_window_steps']:5d}  " + "".join(f"{curves[a][i]:12.4e}" for a in arms) + "\n")
print(open(out + ".txt").read())
