#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ffmlp.py -m gpu -x -q > gpurun_out/pytest_ffmlp.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_ffmlp.log
timeout 600 python tools/bench_ffmlp.py > gpurun_out/ffmlp.log 2>&1
tail -3 gpurun_out/pytest_ffmlp.log; grep -v amdgpu gpurun_out/ffmlp.log
