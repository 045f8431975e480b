#!/bin/bash
# SQ counters of the kernels matching a name fragment: bash tools/pmc_sq.sh <fragment> <python script + args>
R=${GRAFT_REPO_ROOT:-$(pwd)}
FRAG=$1; shift
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
i=0
FILES=""
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VALU" \
           "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
           "SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rm -rf /tmp/pq_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pq_$i -o c -- python "$@" > $R/gpurun_out/pmc_sq_$i.log 2>&1
  FILES="$FILES $(find /tmp/pq_$i -name '*counter_collection.csv')"
done
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_sq.json --meta "command=$*" $FILES > /dev/null
python - <<PY
import json
d=json.load(open("$R/gpurun_out/pmc_sq.json"))
for k,v in d.items():
    if isinstance(v, dict) and "$FRAG" in k:
        print(k)
        w=v.get("SQ_WAVES_avg",1)
        for a,b in sorted(v.items()):
            if a.endswith("_avg"): print(f"   {a[:-4]:28s} {b:14.0f}   per wave {b/w:12.1f}")
PY
