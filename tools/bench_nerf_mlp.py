"""Timing and run-to-run bit-stability of the two-nets-in-one-launch kernels (csrc/nerf_mlp.hip) beside the one-launch-per-net
route they replace, at the 4096-ray training batch.  usage: python tools/bench_nerf_mlp.py [--rows 133000] [--runs 200]"""
import argparse, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enerf_amd import _lib, fused_network as fn
from enerf_amd.network import NeRFNetwork

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=133000)
ap.add_argument("--runs", type=int, default=200)
a = ap.parse_args()
lib = _lib.lib()
DEV = "cuda"
N = a.rows
torch.manual_seed(1)
m = NeRFNetwork(encoding="hashgrid", bound=3, cuda_ray=True, out_dim_color=3).to(DEV)
m.encoder.embeddings.data.uniform_(-1, 1)
x = torch.rand(N, 3, device=DEV) * 6 - 3
d = torch.nn.functional.normalize(torch.randn(N, 3, device=DEV), dim=-1)
gs, gc = torch.randn(N, device=DEV), torch.randn(N, 3, device=DEV)
params = fn.network_params(m)
cfg, offs = fn.network_cfg(m), fn.encoder_offsets(m)


def step():
    s, c, sv = fn.nerf_forward(x, d, cfg, True, params[0], offs, *params[1:])
    g = fn.nerf_backward(sv, gs, gc, raw=True)
    return s, c, g[1], sv


for fused in (0, 1):
    lib.enerf_debug_nerf_mlp_fused(fused)
    s0, c0, dw0, sv0 = step()
    bad = 0
    for it in range(a.runs):
        s1, c1, dw1, _ = step()
        bad += int((s1 != s0).sum()) + int((c1 != c0).sum()) + int((dw1 != dw0).sum())
    torch.cuda.synchronize()
    _lib.prof.reset(); _lib.prof.enable(True, only=("ffmlp_fwd", "ffmlp_bwd", "mlp_reduce"))
    for it in range(50):
        step()
    torch.cuda.synchronize()
    _lib.prof.enable(False)
    f, nf = _lib.prof.read("ffmlp_fwd"); b, nb = _lib.prof.read("ffmlp_bwd"); r, nr = _lib.prof.read("mlp_reduce")
    print(f"{'both nets per launch' if fused else 'one net per launch  '}: forward {f / 50 * 1e3:6.1f} us ({nf // 50} launches)  backward {b / 50 * 1e3:6.1f} us "
          f"({nb // 50})  dW reduce {r / 50 * 1e3:5.1f} us ({nr // 50})  | {a.runs} repeats of {N} rows: {bad} values differ from the first run")
lib.enerf_debug_nerf_mlp_fused(1)
