cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_ref_gridencoder.py tests/test_gpu_mlp32.py -x -q 2>&1 | tail -3
python tools/bench_tile_adam.py 2>&1 | tail -5
bash tools/r05_timeline.sh r05s default | grep -v "^{"
ENERF_LIB_PATH=$GRAFT_REPO_ROOT/enerf_amd/lib/variants/lib_tatime.so python tools/dev/ta_tiles.py 2>&1 | grep "span\|workgroups 1024\|last to finish"
