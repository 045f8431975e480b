#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "grid" > gpurun_out/pytest_grid.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_grid.log
tail -5 gpurun_out/pytest_grid.log
timeout 600 python tools/bench_kernels.py 2>&1 | grep -E "grid|level" | head -40
timeout 600 python bench.py --no-cpu-baseline --render-frames 0 2>&1 | tail -1 | cut -c1-200
