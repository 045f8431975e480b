cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_steps20.json 2> gpurun_out/bench_steps20.err
tail -c 300 gpurun_out/bench_steps20.err
