#!/bin/bash
# per-kernel stats of the default bench (rocprofv3 --kernel-trace --stats); args are passed to bench.py
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --graph-leg-steps 0 --render-frames 0 --probe-steps 0 "$@" > /tmp/ks.log 2>&1
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:24]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):6d} avg_us {float(r["AverageNs"])/1e3:9.2f} total_ms {float(r["TotalDurationNs"])/1e6:9.3f}')
PY
