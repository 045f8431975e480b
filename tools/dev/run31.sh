cd $GRAFT_REPO_ROOT
python tools/bench_nerf_mlp.py 2>&1 | tail -1
python tools/nerf_fwd_residency.py 256 300 | tail -1
python tools/nerf_fwd_residency.py 768 300 | tail -1
timeout 600 python -m pytest tests/test_gpu_mlp32.py tests/test_gpu_training.py -x -q 2>&1 | tail -2
bash tools/r05_timeline.sh r05w default | grep "start-to-start\|nerf_fwd\|queue 1"
