#!/bin/bash
# Timeline (all queues) of one COLD training step (no sample budget yet: the reference's first 16 steps).
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/ct
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ct -o s -- python $R/bench.py --no-cpu-baseline --render-frames 0 --graph-leg-steps 0 --probe-steps 0 --no-live-timing --steps 12 --warmup 2 "$@" > /tmp/ct.log 2>&1
T=$(find /tmp/ct -name "*kernel_trace.csv" | head -1)
python - "$T" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nm = lambda r: re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:44]
adam = [i for i, r in enumerate(rows) if nm(r).startswith("void k_grid_tile_adam")]
lo, hi = adam[7], adam[8]
t0 = int(rows[lo]["End_Timestamp"])
last_end = {}
print(f"cold step: {(int(rows[hi]['End_Timestamp']) - t0) / 1e3:.1f} us from the previous optimizer launch's end")
for r in rows[lo + 1:hi + 1]:
    q = r["Queue_Id"]
    s, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    idle = (s - last_end.get(q, t0)) / 1e3
    last_end[q] = en
    print(f"q{q:>3} +{(s - t0) / 1e3:9.1f} us  {(en - s) / 1e3:8.1f} us  idle before {idle:7.1f}  {nm(r)}")
PY
grep '^{"metric"' /tmp/ct.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['step_split'])"
