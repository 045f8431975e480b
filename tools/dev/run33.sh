cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_density_update.py tests/test_gpu_training.py tests/test_gpu_cuda_ray_vs_reference_fixture.py -q 2>&1 | tail -3
bash tools/update_kstats.sh 2>&1 | grep -A16 "== partial"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --render-frames 0 --strong-rays 0 --only-legs after_step_256_rgb 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('bench: %.4f ms/step' % d['ms_per_step'], 'after_256', d['other_steps'].get('after_step_256_rgb',{}).get('ms_per_step'))"
