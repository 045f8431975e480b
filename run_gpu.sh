#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_training.py -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -12 gpurun_out/pytest_gpu.log | cut -c1-300
