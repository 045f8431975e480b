#!/bin/bash
# A/B of an environment switch on one box: bash tools/ab_env.sh VAR   (runs the training bench with VAR unset / VAR=0, 3x)
for i in 1 2 3; do
for v in "" "0"; do
  env $1=${v:-1} python bench.py --no-cpu-baseline --graph-leg-steps 0 --render-frames 0 --probe-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$1=${v:-1}', round(d['ms_per_step'],4), round(d['step_split']['steady']['ms_per_step'],4))"
done; done
