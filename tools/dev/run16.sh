cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
bash tools/r05_timeline.sh r05n default | grep -v "^{"
export ENERF_LIB_PATH=$R/enerf_amd/lib/variants/lib_wt.so
echo "== write-through"
python tools/bench_tile_adam.py 2>&1 | tail -5
bash tools/r05_timeline.sh r05o default | grep -v "^{"
