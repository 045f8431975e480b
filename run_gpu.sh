#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_training.py -m gpu -x -q -s > gpurun_out/pytest_train.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_train.log
tail -15 gpurun_out/pytest_train.log
