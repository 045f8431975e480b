#!/bin/bash
# side-stream march: workgroups of the count pass against ms/step at several batch sizes (development aid)
for R in 2048 8192 16384 65536; do
  for B in 256 512 1024 2048 4096 16384; do
    python bench.py --rays $R --march-bg-blocks $B --no-cpu-baseline --render-frames 0 --graph-leg-steps 0 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('rays $R blocks $B: %.4f ms/step' % d['ms_per_step'])"
  done
done
