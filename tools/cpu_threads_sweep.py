"""cpu_baseline thread sweep on the GPU box's host cores: bench.py's CPU leg (reference PyTorch route + C oracle) at
16 / 32 / 64 / all threads, 6 s each -> JSON on stdout (committed as profiles/rNN_cpu_threads_sweep.json)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for t in (8, 16, 32, 64, 0):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-baseline-only", "--cpu-threads", str(t),
                        "--cpu-budget-s", "6"], capture_output=True, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    d = json.loads(lines[-1]) if lines else {"error": r.stderr[-300:]}
    out[str(t) if t else "all"] = {k: d.get(k) for k in ("value", "cores", "cores_total", "cpu_model", "error") if k in d}
print(json.dumps(out, indent=1))
