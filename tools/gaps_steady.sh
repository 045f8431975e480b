#!/bin/bash
# Gaps between consecutive kernels of the main HIP queue in steady training steps (host-bound or device-bound?)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/gs
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/gs -o s -- python $R/bench.py --no-cpu-baseline --render-frames 0 --graph-leg-steps 0 --probe-steps 0 --steps 60 --warmup 20 "$@" > /tmp/gs.log 2>&1
T=$(find /tmp/gs -name "*kernel_trace.csv" | head -1)
python - "$T" <<'PY'
import csv, sys, re
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steady steps: take the window between the last two density sweeps' k_ema kernels... simpler: last 14 steps
names = [re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:28] for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith("void k_grid_tile_adam")]
lo, hi = idx[-14], idx[-2]
sel = rows[lo:hi + 1]
q_main = sel[0]["Queue_Id"]
main = [r for r in sel if r["Queue_Id"] == q_main]
gap_after = defaultdict(list)
for a, b in zip(main, main[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    gap_after[re.sub(r"\(anonymous namespace\)::", "", a["Kernel_Name"])[:40]].append(g)
tot = 0
for k, v in gap_after.items():
    v.sort()
    print(f"gap after {k:40s} n={len(v):3d} median {v[len(v)//2]:6.2f} us  max {v[-1]:6.2f}")
    tot += sum(v) / 12
span = (int(main[-1]["End_Timestamp"]) - int(main[0]["End_Timestamp"])) / 1e3 / 12
print(f"steps spanned 12: {span:.1f} us per step; sum of main-queue gaps {tot:.1f} us per step")
PY
grep '^{"metric"' /tmp/gs.log | cut -c1-120
