#!/bin/bash
# The operand hazard behind csrc/mlp32s_ops.h's operand_ready barrier, reproduced (run on a GPU box):
#   bash tools/nerf_fwd_residency.sh  ->  gpurun_out/r05_mfma_operand_hazard.txt
# Runs the 133 000-sample two-net forward 400 times per row at three workgroups per CU (grid cap 768) with the library as
# shipped, with the barrier compiled out (-DMLP32S_NO_OPERAND_BARRIER), and with the barrier compiled out in a build whose
# MFMAs are padded apart (-mllvm -amdgpu-mfma-padding-ratio=100: the scheduler then places the conversions right in front
# of the MFMAs that read them), and the last one again with the barrier back in.
R=$(cd $(dirname $0)/.. && pwd)
OUT=$R/gpurun_out/r05_mfma_operand_hazard.txt
mkdir -p $R/gpurun_out
{
  echo "# k_nerf_fwd, 133000 samples (4157 tiles), 400 launches per row; yardstick = the one-launch-per-net kernels"
  echo "# as shipped (barrier in split8 / exact8; forward at up to three workgroups per CU)"
  for CAP in 256 512 768; do python $R/tools/nerf_fwd_residency.py $CAP 400; done
  V=$(bash $R/tools/dev/build_variant.sh nobar nerf_mlp.hip "-DMLP32S_NO_OPERAND_BARRIER" | tail -1)
  echo "# barrier compiled out"
  for CAP in 256 512 768; do ENERF_LIB_PATH=$V python $R/tools/nerf_fwd_residency.py $CAP 400; done
  V2=$(bash $R/tools/dev/build_variant.sh nobarpad nerf_mlp.hip "-DMLP32S_NO_OPERAND_BARRIER -mllvm -amdgpu-mfma-padding-ratio=100" | tail -1)
  echo "# barrier compiled out, MFMAs padded apart"
  for CAP in 256 512 768; do ENERF_LIB_PATH=$V2 python $R/tools/nerf_fwd_residency.py $CAP 400; done
  V3=$(bash $R/tools/dev/build_variant.sh barpad nerf_mlp.hip "-mllvm -amdgpu-mfma-padding-ratio=100" | tail -1)
  echo "# barrier in, MFMAs padded apart"
  ENERF_LIB_PATH=$V3 python $R/tools/nerf_fwd_residency.py 768 400
  rm -f $V $V2 $V3
} 2>&1 | grep -v '^\[enerf_amd' | grep -v amdgpu.ids | tee $OUT
