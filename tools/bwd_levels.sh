cd /tmp && export TMPDIR=/tmp
for lv in 0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15; do
  rm -rf /tmp/bs_$lv
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bs_$lv -- python $GRAFT_REPO_ROOT/tools/bwd_split.py --level $lv --reps 20 > /dev/null 2>&1
  f=$(find /tmp/bs_$lv -name "*kernel_stats.csv" | head -1)
  python - "$f" $lv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_grid_bwd_bin" in r["Name"]:
        print("level", sys.argv[2], "bin avg_us", round(float(r["AverageNs"]) / 1e3, 2))
PY
done
