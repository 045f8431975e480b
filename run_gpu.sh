#!/bin/bash
bash tools/profile_round.sh r01 > gpurun_out/profile_round.log 2>&1
tail -3 gpurun_out/profile_round.log
R=$GRAFT_REPO_ROOT; cd $R
python bench.py > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log > gpurun_out/bench_default.json
for r in 1024 16384 65536; do python bench.py --no-cpu-baseline --render-frames 0 --rays $r 2>&1 | tail -1 > gpurun_out/bench_rays_$r.json; done
python bench.py --no-cpu-baseline --mode events 2>&1 | tail -1 > gpurun_out/bench_events.json
python bench.py --no-cpu-baseline --mode events --graphs --render-frames 0 2>&1 | tail -1 > gpurun_out/bench_events_graphs.json
python bench.py --no-cpu-baseline --graphs --render-frames 0 2>&1 | tail -1 > gpurun_out/bench_graphs.json
python bench.py --no-cpu-baseline --net ff --bound 2 2>&1 | tail -1 > gpurun_out/bench_ff.json
python tools/bench_kernels.py > gpurun_out/bench_kernels.log 2>&1
python tools/bench_mlp32.py > gpurun_out/mlp32.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -2 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -1
