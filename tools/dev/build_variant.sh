#!/bin/bash
# A/B builds of one translation unit: bash tools/dev/build_variant.sh <name> <file.hip> "<-D flags>"  -> gpurun_out/../lib_<name>.so
# (links the variant object with the other, already built objects of enerf_amd/lib/obj)
set -e
NAME=$1; SRC=$2; DEFS=$3
R=$(cd $(dirname $0)/../.. && pwd)
OBJ=$R/enerf_amd/lib/obj
mkdir -p $R/enerf_amd/lib/variants
EXTRA=""
case $SRC in mlp32s.hip|mlp32s_f16.hip|nerf_mlp.hip|nerf_mlp_bwd.hip|ffmlp.hip) EXTRA="-mllvm -amdgpu-mfma-vgpr-form -DENERF_MFMA_VGPR_FORM";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function $EXTRA $DEFS -c $R/enerf_amd/csrc/$SRC -o /tmp/variant_$NAME.o
OTHERS=$(ls $OBJ/*.o | grep -v "/${SRC%.hip}.o" | grep -v "/${REPLACES:-none}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS /tmp/variant_$NAME.o -ldl -o $R/enerf_amd/lib/variants/lib_$NAME.so
echo $R/enerf_amd/lib/variants/lib_$NAME.so
