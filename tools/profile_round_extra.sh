#!/bin/bash
# Evidence beyond tools/profile_round.sh, added in round 3:  bash tools/profile_round_extra.sh r03
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
# bench lines: driver shape, default, the other shapes
timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > $OUT/${TAG}_bench_steps20_warmup5.json
timeout 600 python bench.py 2>&1 | tail -1 > $OUT/${TAG}_bench_default.json
for spec in "1024rays:--rays 1024" "16384rays:--rays 16384" "65536rays:--rays 65536" "events:--mode events" \
            "events_bound2:--mode events --bound 2" "ffnet:--net ff --bound 2" "amp_bf16:--amp-bf16" "fp16:--fp16" "fp16_autocast:--fp16-autocast" \
            "noprefetch:--no-prefetch"; do
  name=${spec%%:*}; flags=${spec#*:}
  timeout 600 python bench.py --no-cpu-baseline --render-frames 0 --graph-leg-steps 0 --other-legs 0 --strong-rays 0 $flags 2>&1 | tail -1 > $OUT/${TAG}_bench_$name.json
done
# split-bf16 MLP kernels: accuracy of both arithmetic modes, kernel times, SQ counters
timeout 300 python tools/diag_mlp32_split.py --B 5000 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_mlp32_accuracy.txt
for P in 0 1; do for B in 137851 2097152; do echo "precision=$P B=$B"; timeout 300 python tools/bench_mlp32.py --precision $P --B $B 2>&1 | grep -E "^(sigma|color)"; done; done > $OUT/${TAG}_mlp32_kernels.txt 2>&1
bash tools/pmc_mlp32.sh --precision 1 > /dev/null 2>&1; cp $OUT/pmc_mlp32.json $OUT/${TAG}_pmc_sq_mlp32_split.json
# host side: the step as one library call vs the Python-driven step, and the data-parallel tails on one-rank RCCL
for f in "--rays 64" "--rays 64 --python-step" "--rays 4096" "--rays 4096 --python-step"; do timeout 200 python tools/cpu_profile_step.py --no-profile $f 2>&1 | grep -E "enqueue|host us"; done > $OUT/${TAG}_host_step.txt
timeout 600 python tools/dp_tail_overhead.py 2>&1 | grep -E "ms/step" > $OUT/${TAG}_dp_tail_one_rank_rccl.txt
ls -la $OUT | grep ${TAG}_ | wc -l
