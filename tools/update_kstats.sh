#!/bin/bash
# per-kernel time of update_extra_state (full sweep and partial): bash tools/update_kstats.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
cat > /tmp/upd.py <<PY
import sys, torch
sys.path.insert(0, "$R")
from enerf_amd.network import NeRFNetwork
from enerf_amd import scene
m = NeRFNetwork(encoding="hashgrid", bound=3, cuda_ray=True, out_dim_color=3).cuda()
scene.install_occupancy(m)
m.train()
mode = sys.argv[1]
for i in range(10):
    m.iter_density = 0 if mode == "full" else 100
    m.update_extra_state()
torch.cuda.synchronize()
PY
for MODE in full partial; do
rm -rf /tmp/ks
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o s -- python /tmp/upd.py $MODE > $R/gpurun_out/upd_$MODE.log 2>&1
find /tmp/ks -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/upd_$MODE.csv \;
python - <<PY
import csv
rows=[r for r in csv.DictReader(open("$R/gpurun_out/upd_$MODE.csv"))
      if "k_" in r["Name"] or "rocprim" in r["Name"] or "fillBuffer" in r["Name"]]   # the update's own launches
print("== $MODE (per update)")
for r in rows[:14]:
    print(f'{r["Name"][:80]:80s} calls {int(r["Calls"])/10:5.1f}  us {float(r["TotalDurationNs"])/1e4:8.1f}')
print("sum us/update", sum(float(r["TotalDurationNs"]) for r in rows)/1e4)
PY
done
