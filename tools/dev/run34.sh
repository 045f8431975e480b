cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_density_update.py tests/test_gpu_training.py -q 2>&1 | tail -3
bash tools/update_kstats.sh 2>&1 | grep "k_sort\|sum us"
