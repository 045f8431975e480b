#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 30 --warmup 20 --no-cpu-baseline > gpurun_out/bench.log 2>&1
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 20 --no-cpu-baseline --render-frames 0 > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.log 2>&1; find /tmp/prof -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/ \; )
tail -25 gpurun_out/pytest_gpu.log;  tail -1 gpurun_out/bench.log | cut -c1-1800
