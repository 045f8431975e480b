#!/bin/bash
# round-3 check of the split-bf16 MLP kernels: parity tests in both modes, then kernel times in both modes
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_mlp32.py -m gpu -q 2>&1 | grep -E "^(FAILED|ERROR|[0-9]+ (passed|failed))|assert " | cut -c1-300 | tail -30 > gpurun_out/r03_mlp_tests.log
for P in 0 1; do
  for B in 137851 2097152; do
    echo "precision=$P B=$B"; timeout 300 python tools/bench_mlp32.py --precision $P --B $B 2>&1 | grep -E "^(sigma|color)"
  done
done > gpurun_out/r03_mlp_bench.log 2>&1
cat gpurun_out/r03_mlp_tests.log gpurun_out/r03_mlp_bench.log
