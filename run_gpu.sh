#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/smoke.log 2>&1
timeout 600 python bench.py > gpurun_out/bench_default.log 2>&1
tail -3 gpurun_out/pytest_gpu.log; tail -1 gpurun_out/smoke.log; tail -1 gpurun_out/bench_default.log | cut -c1-600
