#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log | cut -c1-250
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --render-frames 0 2>&1 | tail -1 | cut -c1-160; done
timeout 600 python bench.py --no-cpu-baseline --render-frames 0 --mode events 2>&1 | tail -1 | cut -c1-160
python tools/cpu_profile_step.py 2>&1 | grep enqueue
