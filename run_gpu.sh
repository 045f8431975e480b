#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python tools/bench_kernels.py --levels > gpurun_out/kernels.log 2>&1
timeout 600 python bench.py --steps 30 --warmup 20 --no-cpu-baseline > gpurun_out/bench.log 2>&1
tail -5 gpurun_out/pytest_gpu.log; grep -v "level" gpurun_out/kernels.log; grep "level" gpurun_out/kernels.log | head -16;  tail -1 gpurun_out/bench.log | cut -c1-1500
