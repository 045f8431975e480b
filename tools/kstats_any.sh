#!/bin/bash
# per-kernel average duration of any python command: bash tools/kstats_any.sh <script.py> [args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/ka
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ka -o s -- python "$@" > $R/gpurun_out/kstats_any.log 2>&1
python - <<'PY'
import csv, glob, re
rows = list(csv.DictReader(open(glob.glob("/tmp/ka/**/*kernel_stats.csv", recursive=True)[0])))
for r in rows[:18]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    print("%-100s calls %6s  avg us %8.1f" % (n[:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
