#!/bin/bash
# first GPU contact: parity tests, smoke, bench, kernel trace
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0))" > gpurun_out/dev.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 600 python bench.py --steps 30 --warmup 20 > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
tail -3 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log; tail -2 gpurun_out/bench.log
