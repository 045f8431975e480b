#!/bin/bash
# SQ counters of the fp32 MLP kernels at training-batch size: bash tools/pmc_mlp32.sh  (-> gpurun_out/pmc_mlp32.json)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
i=0
FILES=""
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VALU" \
           "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"; do
  i=$((i+1))
  rm -rf /tmp/pm_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pm_$i -o c -- python $R/tools/bench_mlp32.py "$@" > $R/gpurun_out/pmc_mlp32_$i.log 2>&1
  FILES="$FILES $(find /tmp/pm_$i -name '*counter_collection.csv')"
done
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_mlp32.json --meta "command=tools/bench_mlp32.py $*" $FILES
python - <<PY
import json
d=json.load(open("$R/gpurun_out/pmc_mlp32.json"))
for k,v in d.items():
    if isinstance(v, dict) and "mlp32" in k:
        print(k)
        print("   ", {a: round(b) if isinstance(b,(int,float)) else b for a,b in v.items()})
PY
