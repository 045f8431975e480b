#!/bin/bash
# Round-6 measurements of the carried march and the round's side questions, one GPU call:  bash tools/r06_measure.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06
mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -x -q > $OUT/gpu_suite_3.txt 2>&1; tail -3 $OUT/gpu_suite_3.txt
bash tools/r05_timeline.sh r06 default nocarry carry256 carry1024 > /dev/null 2>&1
for v in default nocarry carry256 carry1024; do echo "== $v"; cat $OUT/r06_step_timeline_$v.txt | cut -c1-150; done
for mode in carry nocarry; do
  if [ $mode = nocarry ]; then export ENERF_NO_CARRY_COUNT=1; else unset ENERF_NO_CARRY_COUNT; fi
  python bench.py --steps 20 --warmup 5 > $OUT/bench_3_$mode.json 2> $OUT/bench_3_$mode.err
  python -c "
import json
d=json.loads([l for l in open('$OUT/bench_3_$mode.json') if l.startswith('{\"metric\"')][-1])
print('$mode', d['value'], d['ms_per_step'], d['step_split'], d['roofline']['frac'], d['roofline']['table_backward']['frac'], d['host_enqueue_ms_per_step'])"
done
unset ENERF_NO_CARRY_COUNT
python tools/sweep_ab.py 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep_ab.txt
# what -amdgpu-mfma-vgpr-form buys: the five 16-bit MFMA units built in the default (AGPR) form
for v in shipped agpr_form; do
  if [ $v = agpr_form ]; then export ENERF_LIB_PATH=$R/enerf_amd/lib/variants/lib_agpr_form.so; else unset ENERF_LIB_PATH; fi
  echo "== $v"; python tools/bench_nerf_mlp.py 2>&1 | grep -v amdgpu.ids | tail -8; python tools/bench_ffmlp.py --batch 2097152 2>&1 | grep -v amdgpu.ids | tail -6
done > $OUT/agpr_form_cost.txt 2>&1
unset ENERF_LIB_PATH
cat $OUT/agpr_form_cost.txt
