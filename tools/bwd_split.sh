#!/bin/bash
# usage: bash tools/bwd_split.sh  -> per-kernel average durations of the binned grid backward (all levels, level 10, level 2)
cd /tmp && export TMPDIR=/tmp
for lv in -1 10 2; do
  rm -rf /tmp/bs_$lv
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bs_$lv -- python $GRAFT_REPO_ROOT/tools/bwd_split.py --level $lv > /dev/null 2>&1
  f=$(find /tmp/bs_$lv -name "*kernel_stats.csv" | head -1)
  echo "== level $lv"
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_grid_bwd" in r["Name"]:
        print(r["Name"][:60], "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 2))
PY
done
