cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4
python tools/bench_nerf_mlp.py 2>&1 | tail -2
bash tools/r05_timeline.sh r05v default | grep -v "^{"
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_v.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_v.json').read().strip().splitlines()[-1])
print('bench: %.4f ms/step' % d['ms_per_step'], {k: round(v['ms_per_step'],4) if v['ms_per_step'] else None for k,v in d['step_split'].items()})
for k,v in d['other_steps'].items(): print('  ', k, round(v.get('ms_per_step',0),4))
print('ffmlp', d['roofline_mfma_ffmlp']['frac'], 'ffnerf', d['render']['ffmlp_nets']['roofline_mfma']['frac'], 'mfma', d['roofline_mfma']['frac'], d['roofline_mfma']['kernel_ms_per_step'])
print('render', d['render']['ms_per_frame'], d['render']['ffmlp_nets']['ms_per_frame'])
PY
