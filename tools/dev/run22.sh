cd $GRAFT_REPO_ROOT
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --other-legs 0 --render-frames 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('bench: %.4f ms/step' % d['ms_per_step'], {k: round(v['ms_per_step'],4) if v['ms_per_step'] else None for k,v in d['step_split'].items()}, d['roofline']['table_backward']['tile_adam_ms'], d['roofline']['table_backward']['binning_pass_ms'])"
done
