#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_r01; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/p_mfma; timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_mfma -o c -- python $R/tools/bench_ffmlp.py --batch 2097152 > $OUT/ffmlp_under_mfma.log 2>&1
python $R/tools/pmc_summary.py $OUT/r01_pmc_mfma_ffmlp.json --by-grid --meta "command=tools/bench_ffmlp.py --batch 2097152" $(find /tmp/p_mfma -name "*counter_collection.csv")
cd $R; python tools/bench_ffmlp.py > gpurun_out/ffmlp.log 2>&1; tail -5 gpurun_out/ffmlp.log
