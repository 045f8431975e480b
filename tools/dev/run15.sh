cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
echo "== paired rounds"; bash tools/update_kstats.sh 2>&1 | grep -A4 "== full" 
bash tools/r05_timeline.sh r05l default | grep "k_grid_fwd\|start-to-start"
export ENERF_LIB_PATH=$R/enerf_amd/lib/variants/lib_fwdseq.so
echo "== sequential"; bash tools/update_kstats.sh 2>&1 | grep -A4 "== full"
bash tools/r05_timeline.sh r05m default | grep "k_grid_fwd\|start-to-start"
unset ENERF_LIB_PATH
timeout 600 python -m pytest tests/test_gpu_ref_gridencoder.py tests/test_gpu_density_update.py -x -q 2>&1 | tail -3
