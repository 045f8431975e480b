cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r05 > gpurun_out/profile_round_r05.log 2>&1
bash tools/profile_round_extra.sh r05 > gpurun_out/profile_round_extra_r05.log 2>&1
bash tools/r05_timeline.sh r05 default exact rays64 noprefetch > gpurun_out/r05_timelines.log 2>&1
python tools/bench_nerf_mlp.py 2>&1 | grep -v amdgpu > gpurun_out/r05_nerf_mlp_kernels.txt
ls gpurun_out | wc -l
