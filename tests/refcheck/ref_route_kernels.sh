#!/bin/bash
# Per-kernel time of the reference-native training route (tests/refcheck/ref_route_speed.py, "reference kernels" only) under
# rocprofv3 --kernel-trace --stats:  bash tests/refcheck/ref_route_kernels.sh > gpurun_out/r04_ref_route_kernels.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/rr
REF_ROUTE_ONLY="reference kernels" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rr -o r -- python $R/tests/refcheck/ref_route_speed.py > /tmp/rr.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/rr/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("the reference-native route (its raymarching.cu / gridencoder.cu / shencoder.cu for gfx950, nn.Linear on torch's GEMMs, autograd,")
print("torch Adam, Python update_extra_state) under rocprofv3 --kernel-trace --stats: 100 steps (20 + 80), 7 with update_extra_state")
print(f"sum of kernel time: {tot / 100 / 1e3:.1f} us/step")
for r in rows[:26]:
    print(f"{r['Name'][:96]:96s} calls/step {int(r['Calls']) / 100:6.2f}  us/step {int(r['TotalDurationNs']) / 100 / 1e3:9.1f}")
PY
grep '^{' /tmp/rr.log | cut -c1-160
