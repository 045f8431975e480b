#!/bin/bash
# fixed cost vs per-tile cost of the fp32 MLP kernels: batches of exactly k tiles per wave (1024 resident waves x 32 rows)
for k in 1 2 3 4 5 8; do
  python tools/bench_mlp32.py --B $((32768 * k)) 2>&1 | grep -E "^(sigma|color)" | sed "s/^/k=$k /"
done
