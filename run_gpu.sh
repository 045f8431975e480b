#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_mlp32.py tests/test_gpu_training.py -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log | cut -c1-300
for i in 1 2 3; do timeout 600 python bench.py --no-cpu-baseline --render-frames 0 2>&1 | tail -1 | cut -c1-160; done
