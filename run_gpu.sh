#!/bin/bash
timeout 600 python tools/bench_ffmlp.py 2>&1 | grep -E "inference" | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_ffmlp.py -m gpu -x -q 2>&1 | tail -2
