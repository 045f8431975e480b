#!/bin/bash
# A/B: where the side-stream march of the next batch is released (same box, alternating)
for i in 1 2; do for at in forward mlp_backward; do
  python bench.py --no-cpu-baseline --graph-leg-steps 0 --render-frames 0 --prefetch-at $at 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$at', round(d['ms_per_step'],4), round(d['step_split']['steady']['ms_per_step'],4), (d.get('roofline_mfma') or {}).get('frac'))"
done; done
