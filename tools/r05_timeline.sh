#!/bin/bash
# steady-step timelines (rocprofv3 kernel trace) of the bench's step: bash tools/r05_timeline.sh <tag> [variants: default exact rays64 ...]
TAG=${1:-r05}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
LEAN="--no-cpu-baseline --render-frames 0 --probe-steps 0 --other-legs 0 --strong-rays 0 --steps 64 --warmup 32"
for V in ${@:-default}; do
  case $V in
    default) X="";; nocarry) X=""; export ENERF_NO_CARRY_COUNT=1;; carry256) X=""; export ENERF_MARCH_CARRY_BLOCKS=256;; carry384) X=""; export ENERF_MARCH_CARRY_BLOCKS=384;; carry640) X=""; export ENERF_MARCH_CARRY_BLOCKS=640;; carry768) X=""; export ENERF_MARCH_CARRY_BLOCKS=768;; carry1024) X=""; export ENERF_MARCH_CARRY_BLOCKS=1024;; noprefetch) X="--no-prefetch";; exact) X="--mlp-exact";; rays64) X="--rays 64";; events) X="--mode events --bound 2";; events_nocarry) X="--mode events --bound 2"; export ENERF_NO_CARRY_COUNT=1;; *) X="";;
  esac
  rm -rf /tmp/tl_$V
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$V -o t -- python $R/bench.py $LEAN $X > $OUT/tl_$V.log 2>&1
  python $R/tools/step_timeline.py $(find /tmp/tl_$V -name "*kernel_trace.csv" | head -1) 30 > $OUT/${TAG}_step_timeline_$V.txt 2>&1
  grep '^{"metric"' $OUT/tl_$V.log | cut -c1-160 >> $OUT/${TAG}_step_timeline_$V.txt
  cat $OUT/${TAG}_step_timeline_$V.txt
  unset ENERF_NO_CARRY_COUNT ENERF_MARCH_CARRY_BLOCKS
done
