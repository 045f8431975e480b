cd $GRAFT_REPO_ROOT
bash tools/update_kstats.sh 2>&1 | grep -A6 "== full\|== partial" | head -20
timeout 600 python -m pytest tests/test_gpu_density_update.py -x -q 2>&1 | tail -2
