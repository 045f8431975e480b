cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --other-legs 0 --render-frames 0 --strong-rays 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('bench: %.4f ms/step' % d['ms_per_step'], {k: round(v['ms_per_step'],4) if v['ms_per_step'] else None for k,v in d['step_split'].items()})"
done
timeout 600 python -m pytest tests/test_gpu_training.py -x -q 2>&1 | tail -2
