#!/bin/bash
# Round-5 baseline evidence: driver line, steady-step timelines (default / exact fp32 / 64 rays), drop-in route by kernel,
# TCC hit / request counters of k_grid_fwd.   bash tools/r05_baseline.sh [tag]
TAG=${1:-r05_base}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python $R/bench.py --steps 20 --warmup 5 > $OUT/bench_driver_line.json 2> $OUT/bench_driver_line.err
LEAN="--no-cpu-baseline --render-frames 0 --probe-steps 0 --other-legs 0 --strong-rays 0 --steps 64 --warmup 32"
for V in default exact rays64 rays65536; do
  case $V in
    default) X="";; exact) X="--mlp-exact";; rays64) X="--rays 64";; rays65536) X="--rays 65536";;
  esac
  rm -rf /tmp/tl_$V
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$V -o t -- python $R/bench.py $LEAN $X > $OUT/tl_$V.log 2>&1
  python $R/tools/step_timeline.py $(find /tmp/tl_$V -name "*kernel_trace.csv" | head -1) 30 > $OUT/${TAG}_step_timeline_$V.txt 2>&1
  grep '^{"metric"' $OUT/tl_$V.log | cut -c1-160 >> $OUT/${TAG}_step_timeline_$V.txt
done
# the drop-in route (reference-shaped run_cuda through the pybind modules), by kernel
rm -rf /tmp/di
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/di -o s -- python $R/bench.py --no-cpu-baseline --render-frames 0 --probe-steps 0 --strong-rays 0 --steps 2 --warmup 1 --only-legs dropin_route_rgb --other-legs 200 > $OUT/dropin.log 2>&1
python - > $OUT/${TAG}_dropin_route_kernels.txt <<PY
import csv, glob, re
rows = list(csv.DictReader(open(glob.glob("/tmp/di/**/*kernel_stats.csv", recursive=True)[0])))
print("# bench.py --steps 2 --warmup 1 --only-legs dropin_route_rgb --other-legs 200 under rocprofv3 --kernel-trace --stats; us per leg step = total / 220")
tot = 0
for r in rows[:45]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    t = float(r["TotalDurationNs"]) / 1e3 / 220
    tot += t
    print("%-110s calls/step %6.2f  us/step %8.1f" % (n[:110], int(r["Calls"]) / 220, t))
print("sum %.1f us/step" % tot)
PY
grep '^{"metric"' $OUT/dropin.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(json.dumps(d.get('other_steps')))" >> $OUT/${TAG}_dropin_route_kernels.txt
# L2 counters of the roofline kernel (one pass, counters only)
rm -rf /tmp/tcc
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d /tmp/tcc -o c -- python $R/bench.py $LEAN > $OUT/tcc.log 2>&1
python $R/tools/pmc_summary.py $OUT/${TAG}_pmc_tcc_bench.json --meta "command=bench.py $LEAN" $(find /tmp/tcc -name "*counter_collection.csv") > $OUT/pmc_tcc.log 2>&1
rm -rf /tmp/tcp
timeout 600 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d /tmp/tcp -o c -- python $R/bench.py $LEAN > $OUT/tcp.log 2>&1
python $R/tools/pmc_summary.py $OUT/${TAG}_pmc_tcp_bench.json --meta "command=bench.py $LEAN" $(find /tmp/tcp -name "*counter_collection.csv") > $OUT/pmc_tcp.log 2>&1
ls -la $OUT
