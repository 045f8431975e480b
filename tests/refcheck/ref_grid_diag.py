"""The reference's own grid encoder (oracle/_ref/_ref_gridencoder: gridencoder.cu built for gfx950, oracle/build_ref.py)
beside this library's kernels and the C oracle on the same MI355X: per-level differences of the forward, the backward,
the Jacobian, and kernel times.  TEST INFRASTRUCTURE (lives under tests/: it runs the checker's kernels).

    gpurun -- 'python tests/refcheck/ref_grid_diag.py > gpurun_out/ref_grid_diag.txt'
"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O, build_ref as br   # noqa: E402
from enerf_amd import ext as E   # noqa: E402
from enerf_amd.ext import build as eb   # noqa: E402

DEV = "cuda"
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3     # us


def main():
    eb.build(verbose=False); E.activate()
    prod = importlib.import_module("_gridencoder")
    ref = br.load("gridencoder")
    for bound in (2, 3):
        offsets, pls = O.grid_offsets(desired_resolution=2048 * bound)
        S = float(np.log2(pls)); L = 16; C = 2
        rng = np.random.default_rng(70 + bound)
        emb = rng.uniform(-1, 1, (int(offsets[-1]), C)).astype(np.float32)
        B = 100000
        x = rng.uniform(0, 1, (B, 3)).astype(np.float32)
        o_out, o_jac = O.grid_encode_forward(x, emb, offsets, S, 16, True, 0)
        ce, cx, co = cu(emb), cu(x), cu(offsets)
        outs = {}
        for name, m in (("reference", ref), ("this", prod)):
            out = torch.empty(L, B, C, device=DEV); jac = torch.empty(B, L * 3 * C, device=DEV)
            m.grid_encode_forward(cx, ce, co, out, B, 3, C, L, S, 16, True, jac, 0)
            outs[name] = (out.cpu().numpy(), jac.cpu().numpy())
        print(f"bound {bound}: forward, max |difference| per level (random table in [-1, 1], 100 000 points)")
        print("  level   scale(host)      ref-oracle   this-oracle     ref-this    jac ref-oracle  jac this-oracle")
        for l in range(L):
            sc, res = O.grid_level_params(l, np.float32(S), 16)
            a = np.abs(outs["reference"][0][l] - o_out[l]).max()
            b = np.abs(outs["this"][0][l] - o_out[l]).max()
            c = np.abs(outs["reference"][0][l] - outs["this"][0][l]).max()
            jr = outs["reference"][1].reshape(B, L, 3, C)[:, l]; jt = outs["this"][1].reshape(B, L, 3, C)[:, l]
            jo = o_jac.reshape(B, L, 3, C)[:, l]
            print(f"  {l:5d} {float(sc):13.6f} {a:13.3e} {b:13.3e} {c:13.3e} {np.abs(jr - jo).max():13.3e} {np.abs(jt - jo).max():13.3e}"
                  f"   bit-equal rows ref/oracle {np.mean(np.all(outs['reference'][0][l] == o_out[l], axis=1)):.4f}")
        g = rng.normal(size=(L, B, C)).astype(np.float32)
        ge_o, gi_o = O.grid_encode_backward(g, x, emb, offsets, S, 16, o_jac, 0)
        for name, m in (("reference", ref), ("this", prod)):
            gemb = torch.zeros_like(ce); gin = torch.zeros(B, 3, device=DEV)
            m.grid_encode_backward(cu(g), cx, ce, co, gemb, B, 3, C, L, S, 16, True, cu(outs[name][1]), gin, 0)
            d = np.abs(gemb.cpu().numpy() - ge_o)
            print(f"  backward {name:10s}: max |grad_emb - oracle| {d.max():.3e} (max |grad| {np.abs(ge_o).max():.3f}), "
                  f"grad_inputs max diff {np.abs(gin.cpu().numpy() - gi_o).max():.3e} (max {np.abs(gi_o).max():.1f})")
        # half table
        eh = ce.half()
        for name, m in (("reference", ref), ("this", prod)):
            out = torch.empty(L, B, C, device=DEV, dtype=torch.half); dummy = torch.empty(1, device=DEV, dtype=torch.half)
            m.grid_encode_forward(cx, eh, co, out, B, 3, C, L, S, 16, False, dummy, 0)
            outs[name + "_h"] = out.float().cpu().numpy()
        print(f"  half table forward: max |ref - this| {np.abs(outs['reference_h'] - outs['this_h']).max():.3e}, "
              f"equal entries {np.mean(outs['reference_h'] == outs['this_h']):.5f}")
        for B2 in (133120, 2097152):
            x2 = torch.rand(B2, 3, device=DEV)
            g2 = torch.randn(L, B2, C, device=DEV)
            for name, m in (("reference", ref), ("this", prod)):
                out = torch.empty(L, B2, C, device=DEV); dummy = torch.empty(1, device=DEV)
                t_f = timeit(lambda: m.grid_encode_forward(x2, ce, co, out, B2, 3, C, L, S, 16, False, dummy, 0))
                gemb = torch.zeros_like(ce)
                t_b = timeit(lambda: m.grid_encode_backward(g2, x2, ce, co, gemb, B2, 3, C, L, S, 16, False, dummy, dummy, 0))
                print(f"  {B2:8d} points  {name:10s} forward {t_f:8.1f} us   backward {t_b:8.1f} us")


if __name__ == "__main__":
    main()
