#!/bin/bash
# Steady-state per-kernel time of the bench's training step: difference of two rocprofv3 --stats runs (32 and 96 steps),
# so warm-up (first 16 steps march without a sample budget) cancels.   bash tools/kstats.sh [extra bench flags]
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
for N in 16 80; do
  rm -rf /tmp/ks_$N
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$N -o s -- python $R/bench.py --no-cpu-baseline --render-frames 0 --graph-leg-steps 0 --probe-steps 0 --other-legs 0 --strong-rays 0 --steps $N --warmup 16 "$@" > $R/gpurun_out/kstats_$N.log 2>&1
  find /tmp/ks_$N -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/kstats_$N.csv \;
done
python - <<PY
import csv, re
def load(p):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(p))}
a, b = load("$R/gpurun_out/kstats_16.csv"), load("$R/gpurun_out/kstats_80.csv")
rows = []
for k, (cb, tb) in b.items():
    ca, ta = a.get(k, (0, 0.0))
    rows.append((k, (cb - ca) / 64, (tb - ta) / 64e3))
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
with open("$R/gpurun_out/kstats_steady.csv", "w") as f:
    f.write("kernel,calls_per_step,us_per_step\n")
    for k, c, t in rows:
        f.write(f"\"{k}\",{c:.3f},{t:.2f}\n")
for k, c, t in rows[:30]:
    k = re.sub(r"\(anonymous namespace\)::", "", k)
    print(f"{k[:72]:72s} calls/step {c:5.2f}  us/step {t:7.1f}")
print(f"sum of kernel time: {tot:.1f} us/step")
PY
grep '^{"metric"' $R/gpurun_out/kstats_80.log | cut -c1-200
