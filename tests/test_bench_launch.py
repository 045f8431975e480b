"""bench.py launches itself at N > 1 (`python bench.py --gpus N` with no launcher environment spawns the N ranks through
`torch.distributed.run` on 127.0.0.1 and a free port) and still runs under the driver's own torchrun line.  What is
checked here without a GPU is the launch contract -- rendezvous, barrier, MAX over ranks, ONE JSON line from rank 0,
exit code -- through `--launch-check`, which does no compute (the hot path has no CPU form and bench.py must not grow
one).  On the GPU box the same self-launch runs the real two-rank step (gloo, both ranks on the one device)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _json_lines(stdout):
    return [json.loads(ln) for ln in stdout.splitlines() if ln.startswith("{")]


def test_self_launch_two_ranks_gloo():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--launch-check"],
                       capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout                      # rank 0 only
    j = lines[0]
    assert j["n_gpus"] == 2 and j["backend"] == "gloo" and j["max_over_ranks"] == 2.0
    assert j["strong_rays_per_rank"] == 32768             # configs[3]: 65 536 rays over the ranks


def test_driver_torchrun_form_still_works():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--backend", "gloo",
                        "--launch-check"], capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2


def test_world_size_mismatch_is_refused():
    env = _env()
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], capture_output=True, text=True, env=env,
                       timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_self_launch_real_step_two_ranks_on_one_gpu():
    """The whole bench (weak region + the configs[3] `strong` leg) through the self-launch: two ranks, gloo, both on
    device 0 -- the N > 1 step, its tuning and both timed regions end to end on the one GPU a test box has."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--force-device", "0", "--steps", "6",
                        "--warmup", "18", "--render-frames", "0", "--no-comm-tune", "--probe-steps", "0",
                        "--strong-rays", "16384", "--strong-steps", "4"],
                       capture_output=True, text=True, env=_env(), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout[-2000:]
    j = lines[0]
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["value"] > 0
    assert j["config"]["global_rays"] == 2 * 4096
    st = j["strong"]
    assert "error" not in st, st
    assert st["scaling"] == "strong" and st["global_rays"] == 16384 and st["rays_per_gpu"] == 8192 and st["value"] > 0
