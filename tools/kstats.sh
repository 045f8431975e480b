#!/bin/bash
# per-kernel time of the default bench (training steps only): bash tools/kstats.sh [extra bench flags]
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/ks
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o s -- python $R/bench.py --no-cpu-baseline --render-frames 0 --graph-leg-steps 0 --steps 64 --warmup 16 "$@" > $R/gpurun_out/kstats.log 2>&1
find /tmp/ks -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/kstats.csv \;
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/kstats.csv")))
steps=80
tot=0
for r in rows[:28]:
    t=float(r["TotalDurationNs"])/1e3/steps; tot+=t
    print(f'{r["Name"][:70]:70s} calls/step {int(r["Calls"])/steps:6.2f}  us/step {t:7.1f}  avg {float(r["AverageNs"])/1e3:7.1f}')
print("sum us/step of all kernels", sum(float(r["TotalDurationNs"]) for r in rows)/1e3/steps)
PY
grep '^{"metric"' $R/gpurun_out/kstats.log | cut -c1-160
