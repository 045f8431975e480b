#!/bin/bash
# the other shapes quoted in DESIGN.md section 5 -> gpurun_out/rNN_bench_*.json     bash tools/bench_shapes.sh r02
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
C="--no-cpu-baseline --graph-leg-steps 0 --render-frames 0"
python bench.py $C --rays 1024 > gpurun_out/${TAG}_bench_1024rays.json 2>/dev/null
python bench.py $C --rays 16384 > gpurun_out/${TAG}_bench_16384rays.json 2>/dev/null
python bench.py $C --rays 65536 --steps 60 > gpurun_out/${TAG}_bench_65536rays.json 2>/dev/null
python bench.py $C --net ff --bound 2 > gpurun_out/${TAG}_bench_ffnet.json 2>/dev/null
python bench.py $C --bound 2 --mode events > gpurun_out/${TAG}_bench_events_bound2.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02_bench_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f.split("/")[-1], round(d["ms_per_step"], 4), round(d["value"] / 1e6, 2), "Mrays/s", round(d["train_ray_samples_per_sec"] / 1e6), "Msamples/s",
          "grid_fwd frac", round(d["roofline"]["frac"], 3) if d.get("roofline") else None, d["step_split"]["steady"]["ms_per_step"])
PY
