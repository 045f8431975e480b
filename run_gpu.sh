#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_mlp32.py -m gpu -x -q 2>&1 | tail -2
python tools/bench_mlp32.py 2>&1 | grep inference | cut -c1-150
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --render-frames 0 2>&1 | tail -1 | cut -c1-160; done
