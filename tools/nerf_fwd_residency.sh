#!/bin/bash
# The residency experiment behind csrc/nerf_mlp.hip's one-workgroup-per-CU rule (run on a GPU box):
#   bash tools/nerf_fwd_residency.sh  ->  gpurun_out/r05_nerf_fwd_residency.txt
# Builds nerf_mlp.hip a second time with -DNERF_SHARED_CU (48 KiB of LDS, no whole-SIMD register claim: up to three
# workgroups of k_nerf_fwd per CU) and runs the same 133 000-sample forward 200 times per grid cap with each library.
R=$(cd $(dirname $0)/.. && pwd)
OUT=$R/gpurun_out/r05_nerf_fwd_residency.txt
mkdir -p $R/gpurun_out
{
  echo "# k_nerf_fwd, 133000 samples (4157 tiles), 200 launches per row; yardstick = the one-launch-per-net kernels"
  echo "# shipped build (84 KiB LDS + whole-SIMD registers: one workgroup per CU whatever the grid)"
  for CAP in 256 512 768; do python $R/tools/nerf_fwd_residency.py $CAP 200; done
  SHARED=$(bash $R/tools/dev/build_variant.sh shared_cu nerf_mlp.hip "-DNERF_SHARED_CU" | tail -1)
  echo "# -DNERF_SHARED_CU build (48 KiB LDS, registers as the compiler allocates them: grid 512 = two, 768 = three workgroups per CU)"
  for CAP in 256 512 768; do ENERF_LIB_PATH=$SHARED python $R/tools/nerf_fwd_residency.py $CAP 200; done
  rm -f $SHARED
} 2>&1 | grep -v '^\[enerf_amd' | tee $OUT
