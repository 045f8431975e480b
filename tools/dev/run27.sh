cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ffmlp.py tests/test_gpu_mlp32.py tests/test_gpu_cuda_ray_vs_reference_fixture.py -q 2>&1 | tail -3
python tools/bench_ffmlp.py --batch 2097152 2>&1 | tail -12
