"""Timeline of the steady-state training step from a rocprofv3 kernel trace (development aid): for every kernel of a
step, in start order, its duration and the idle time of ITS stream before it -- where the step's time goes between the
kernels.  usage: python tools/step_timeline.py kernel_trace.csv [steps-from-the-end=40]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 40


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|enerf_mlp32::|void ", "", n)
    return re.sub(r"\(.*", "", n)[:44]


# a step starts at the training-size k_grid_fwd launch (the sweeps are 10x longer: drop steps that contain one)
starts = [i for i, r in enumerate(rows) if "k_grid_fwd<" in r["Kernel_Name"]]
steps = []
for a, b in zip(starts[:-1], starts[1:]):
    seg = rows[a:b]
    if any("k_packbits" in r["Kernel_Name"] or "k_ema" in r["Kernel_Name"] for r in seg):
        continue
    if rows[a]["e"] - rows[a]["s"] > 200000:
        continue
    steps.append((a, b))
steps = steps[-nsteps:]
qkey = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)
acc = collections.OrderedDict()
span = []
for a, b in steps:
    seg = rows[a:b]
    last_end = {}
    span.append(rows[b]["s"] - rows[a]["s"])
    seen = collections.Counter()
    for r in seg:
        q = r.get(qkey, "0")
        name = short(r["Kernel_Name"])
        seen[name] += 1
        key = (name, seen[name], q)
        gap = r["s"] - last_end[q] if q in last_end else 0
        last_end[q] = r["e"]
        d = acc.setdefault(key, [0, 0, 0, 0])
        d[0] += 1
        d[1] += r["e"] - r["s"]
        d[2] += gap
        d[3] += r["s"] - rows[a]["s"]
print(f"{len(steps)} steady steps, start-to-start {sum(span) / len(span) / 1e3:.1f} us")
print(f"{'kernel':46s} {'queue':>6s} {'n':>4s} {'start':>8s} {'dur us':>8s} {'gap before':>10s}")
tot_d = collections.Counter()
tot_g = collections.Counter()
for (name, k, q), (n, d, g, s0) in sorted(acc.items(), key=lambda kv: kv[1][3] / kv[1][0]):
    if n < len(steps) // 2:
        continue
    print(f"{name:46s} {q:>6s} {n:4d} {s0 / n / 1e3:8.1f} {d / n / 1e3:8.1f} {g / n / 1e3:10.1f}")
    tot_d[q] += d / len(steps) / 1e3
    tot_g[q] += g / len(steps) / 1e3
for q in tot_d:
    print(f"queue {q}: kernels {tot_d[q]:.1f} us/step, gaps {tot_g[q]:.1f} us/step")
