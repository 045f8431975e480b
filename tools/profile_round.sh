#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ on the GPU box:  bash tools/profile_round.sh r01
# (kernel-trace stats and each PMC counter set in its own pass; nothing else traced)
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $R/bench.py --no-cpu-baseline"

rm -rf /tmp/p_stats; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o s -- $BENCH > $OUT/bench_under_stats.log 2>&1
find /tmp/p_stats -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_bench_kernel_stats.csv \;

# live hipEvent timing vs rocprofv3 on the same launches: a run whose timed region is the tail of the process
rm -rf /tmp/p_agree; timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_agree -o a -- $BENCH --render-frames 0 --graph-leg-steps 0 --probe-steps 0 --other-legs 0 --strong-rays 0 > $OUT/bench_under_trace.log 2>&1
python $R/tools/agree.py $(find /tmp/p_agree -name "*kernel_trace.csv") $OUT/bench_under_trace.log $OUT/${TAG}_hipevent_vs_rocprof.json

for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$C; timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/p_$C -o c -- $BENCH --render-frames 0 --graph-leg-steps 0 > $OUT/bench_under_$C.log 2>&1
done
PTS=$(grep '^{"metric"' $OUT/bench_under_FETCH_SIZE.log | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.readline())['roofline']['points_per_launch'])" 2>/dev/null)
# the counters are averaged over EVERY k_grid_fwd dispatch of the process: so is this denominator (bench.py:
# grid_fwd_lifetime; no graph leg in this command -- a replayed graph launches the kernel without the wrapper that counts)
PTS_ALL=$(grep '^{"metric"' $OUT/bench_under_FETCH_SIZE.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline())['grid_fwd_lifetime']; print(d['points_per_launch'], d['launches'])" 2>/dev/null)
python $R/tools/pmc_summary.py $OUT/${TAG}_pmc_hbm_bench.json --meta "command=bench.py --no-cpu-baseline --render-frames 0 --graph-leg-steps 0" --meta "grid_fwd_points_per_launch=$PTS" \
  --meta "grid_fwd_points_per_launch_all_dispatches=${PTS_ALL% *}" --meta "grid_fwd_dispatches_in_process=${PTS_ALL#* }" \
  $(find /tmp/p_FETCH_SIZE -name "*counter_collection.csv") $(find /tmp/p_WRITE_SIZE -name "*counter_collection.csv")

for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/k_$C; timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/k_$C -o c -- python $R/tools/bench_kernels.py > $OUT/kernels_under_$C.log 2>&1
done
python $R/tools/pmc_summary.py $OUT/${TAG}_pmc_hbm_kernels.json --by-grid --meta "command=tools/bench_kernels.py" \
  $(find /tmp/k_FETCH_SIZE -name "*counter_collection.csv") $(find /tmp/k_WRITE_SIZE -name "*counter_collection.csv")

rm -rf /tmp/p_mfma; timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_mfma -o c -- python $R/tools/bench_ffmlp.py --batch 2097152 > $OUT/ffmlp_under_mfma.log 2>&1
python $R/tools/pmc_summary.py $OUT/${TAG}_pmc_mfma_ffmlp.json --by-grid --meta "command=tools/bench_ffmlp.py --batch 2097152" $(find /tmp/p_mfma -name "*counter_collection.csv")
# MFMA-busy of the kernels the bench line's roofline_mfma objects are about (mlp32 training kernels, k_ffnerf_infer), in
# the bench's own command
rm -rf /tmp/p_mfma_bench; timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_mfma_bench -o c -- $BENCH --graph-leg-steps 0 > $OUT/bench_under_mfma.log 2>&1
python $R/tools/pmc_summary.py $OUT/${TAG}_pmc_mfma_bench.json --meta "command=bench.py --no-cpu-baseline --graph-leg-steps 0" $(find /tmp/p_mfma_bench -name "*counter_collection.csv")

# the deferred flush (records -> LDS tile sums -> Adam) against pass B + Adam: traffic of k_grid_tile_adam / k_grid_bwd_*
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/t_$C; timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/t_$C -o c -- python $R/tools/bench_tile_adam.py > $OUT/tile_adam_under_$C.log 2>&1
done
python $R/tools/pmc_summary.py $OUT/${TAG}_pmc_hbm_tile_adam.json --meta "command=tools/bench_tile_adam.py" \
  $(find /tmp/t_FETCH_SIZE -name "*counter_collection.csv") $(find /tmp/t_WRITE_SIZE -name "*counter_collection.csv")

# full-frame render, per kernel (whole-frame schedule, FFMLP nets)
bash $R/tools/render_kstats.sh --net ff > $OUT/${TAG}_render_ffmlp_kernels.txt 2>&1
bash $R/tools/kstats.sh > $OUT/${TAG}_step_kernels_steady.txt 2>&1
bash $R/tools/update_kstats.sh > $OUT/${TAG}_update_extra_state_kernels.txt 2>&1
cp $OUT/${TAG}_*.json $OUT/${TAG}_*.csv $OUT/${TAG}_*.txt $R/gpurun_out/ 2>/dev/null
ls -la $OUT
