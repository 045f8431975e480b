#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log | cut -c1-300
python tools/cpu_profile_step.py 2>&1 | grep -v "^$" | cut -c1-150 | sed -n 2,48p
