cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_steps20_warmup5.json 2> gpurun_out/bench_final.err
python bench.py > gpurun_out/r05_bench_default.json 2>> gpurun_out/bench_final.err
python - <<'PY'
import json
for f in ("r05_bench_steps20_warmup5.json","r05_bench_default.json"):
    d=json.loads(open('gpurun_out/'+f).read().strip().splitlines()[-1])
    print(f, '%.4f ms/step' % d['ms_per_step'], {k: round(v['ms_per_step'],4) if v['ms_per_step'] else None for k,v in d['step_split'].items()}, 'roofline', round(d['roofline']['frac'],3), 'tb', round(d['roofline']['table_backward']['frac'],3), 'ffmlp', round(d['roofline_mfma_ffmlp']['frac'],3), 'exact', round(d['value_exact_fp32']/1e6,2), 'cpu', d['cpu_baseline']['value'] if d.get('cpu_baseline') else None)
PY
