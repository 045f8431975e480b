for i in 1 2 3; do
for f in "" "--no-march-clip"; do python bench.py --no-cpu-baseline --graph-leg-steps 0 --render-frames 0 $f 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$f', round(d['ms_per_step'],4))"; done; done
