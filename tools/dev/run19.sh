cd $GRAFT_REPO_ROOT
LEAN="--no-cpu-baseline --render-frames 0 --probe-steps 0 --other-legs 0 --strong-rays 0 --steps 64 --warmup 32"
for m in 1 0 1 0; do
ENERF_FLAG_HANDOVER=$m python bench.py $LEAN 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('handover $m: %.4f ms/step' % d['ms_per_step'], {k: round(v['ms_per_step'],4) if v['ms_per_step'] else None for k,v in d['step_split'].items()}, 'host', round(d['host_enqueue_ms_per_step'],4))"
done
