cd $GRAFT_REPO_ROOT
bash tools/nerf_fwd_residency.sh
python tools/bench_nerf_mlp.py 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_mlp32.py tests/test_gpu_training.py tests/test_gpu_baseline_configs.py -x -q 2>&1 | tail -3
bash tools/r05_timeline.sh r05u default | grep -v "^{"
