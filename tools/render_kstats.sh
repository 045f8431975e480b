#!/bin/bash
# per-kernel time of one full-frame render: bash tools/render_kstats.sh [--net ff] [--bound 2]
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/rk
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rk -o s -- python $R/tools/bench_render.py --frames 9 "$@" > $R/gpurun_out/render_kstats.log 2>&1
find /tmp/rk -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/render_kstats.csv \;
python - <<PY
import csv, re
rows=list(csv.DictReader(open("$R/gpurun_out/render_kstats.csv")))
tot=0
for r in rows[:22]:
    n=re.sub(r"\(anonymous namespace\)::","",r["Name"])
    print(f'{n[:90]:90s} calls/frame {int(r["Calls"])/10:6.1f}  us/frame {float(r["TotalDurationNs"])/1e4:8.1f}')
print("sum us/frame", sum(float(r["TotalDurationNs"]) for r in rows)/1e4)
PY
tail -1 $R/gpurun_out/render_kstats.log
