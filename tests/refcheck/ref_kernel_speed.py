"""The reference's own kernels (oracle/_ref: raymarching.cu / shencoder.cu / gridencoder.cu built for gfx950 with default flags) timed beside
this library's kernels on the same MI355X, same inputs, BASELINE shapes.  TEST INFRASTRUCTURE (it lives under tests/ because it
runs the checker's kernels): the product never loads oracle/_ref.  Both sides are called through their pybind modules (same signatures), hipEvents around 20
back-to-back launches after 3 warm-ups; outputs are the ones tests/test_gpu_ref_kernels.py proves equal.

    gpurun -- 'python tests/refcheck/ref_kernel_speed.py > gpurun_out/ref_kernel_speed.txt'
"""
import importlib
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O, build_ref as br   # noqa: E402
from util import synthetic_density_grid, camera_rays   # noqa: E402
from enerf_amd import ext as E   # noqa: E402
from enerf_amd.ext import build as eb   # noqa: E402

DEV, H = "cuda", 128
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
    prod = {n: importlib.import_module(n) for n in E.MODULES}
    ref_rm, ref_sh = br.load("raymarching"), br.load("shencoder")
    rows = []
    for bound, N in ((3, 4096), (3, 65536), (2, 307200)):
        C = 1 + math.ceil(math.log2(bound))
        bits = cu(O.packbits(synthetic_density_grid(bound, H).reshape(-1), 0.01))
        o, d = camera_rays(N, 5, bound)
        co, cd = cu(o), cu(d)
        aabb = cu(np.array([-bound] * 3 + [bound] * 3, np.float32))
        nears, fars = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
        res = {}
        for name, rm in (("reference", ref_rm), ("this", prod["_raymarching"])):
            t_nf = timeit(lambda: rm.near_far_from_aabb(co, cd, aabb, N, 0.2, nears, fars))
            # training march with a sample budget (the steady-state call: M from the mean count, no host read-back)
            counter = torch.zeros(2, dtype=torch.int32, device=DEV)
            rays = torch.empty(N, 3, dtype=torch.int32, device=DEV)
            M0 = N * 1024 if N <= 4096 else N * 256
            xyzs, dirs, deltas = (torch.empty(M0, 3, device=DEV), torch.empty(M0, 3, device=DEV), torch.empty(M0, 2, device=DEV))

            def march():
                counter.zero_()
                rm.march_rays_train(co, cd, bits, float(bound), 0.0, 1024, N, C, H, M0, nears, fars, xyzs, dirs, deltas, rays,
                                    counter, 1)
            t_march = timeit(march)
            tot = int(counter[0].item())
            assert tot + 128 < M0, (tot, M0)            # nothing dropped: every table row points inside the buffers
            m = tot + 128 - tot % 128
            g = torch.Generator(device=DEV).manual_seed(1)
            sig = torch.rand(m, device=DEV, generator=g) * 20
            rgb = torch.rand(m, 3, device=DEV, generator=g)
            ws, dp, im = torch.empty(N, device=DEV), torch.empty(N, device=DEV), torch.empty(N, 3, device=DEV)
            if name == "reference":        # its table is in atomic order: compositing works on any order
                pass
            dl = deltas[:m].contiguous()
            t_cf = timeit(lambda: rm.composite_rays_train_forward(sig, rgb, dl, rays, m, N, ws, dp, im))
            g_ws, g_im = torch.randn(N, device=DEV, generator=g), torch.randn(N, 3, device=DEV, generator=g)
            gs, gc = torch.zeros(m, device=DEV), torch.zeros(m, 3, device=DEV)
            t_cb = timeit(lambda: rm.composite_rays_train_backward(g_ws, g_im, sig, rgb, dl, rays, ws, im, m, N, gs, gc))
            # one inference round: every ray alive, 8 steps
            alive = torch.arange(N, dtype=torch.int32, device=DEV)
            Mi = N * 8
            gx, gd, gl = torch.empty(Mi, 3, device=DEV), torch.empty(Mi, 3, device=DEV), torch.empty(Mi, 2, device=DEV)
            rt = nears.clone()
            t_mi = timeit(lambda: rm.march_rays(N, 8, alive, rt, co, cd, float(bound), 0.0, 1024, C, H, bits, nears, fars, gx, gd,
                                                gl, 0))
            res[name] = dict(samples=tot, near_far=t_nf, march_train=t_march, comp_fwd=t_cf, comp_bwd=t_cb, march_inf=t_mi)
        rows.append((bound, N, res))
    print("kernel times in us (hipEvents, 20 launches each; march_rays_train includes the 8-byte counter clear both sides do)")
    for bound, N, res in rows:
        r, t = res["reference"], res["this"]
        print(f"bound {bound}, {N} rays, {t['samples']} samples marched:")
        for k in ("near_far", "march_train", "comp_fwd", "comp_bwd", "march_inf"):
            print(f"   {k:12s} reference {r[k]:9.1f}   this library {t[k]:9.1f}   x{r[k] / t[k]:5.2f}")
    for B in (133120, 2097152):
        v = torch.nn.functional.normalize(torch.randn(B, 3, device=DEV), dim=-1)
        y = torch.empty(B, 16, device=DEV); j = torch.empty(B, 48, device=DEV)
        gi = torch.zeros(B, 3, device=DEV); gr = torch.randn(B, 16, device=DEV)
        out = {}
        for name, sh in (("reference", ref_sh), ("this", prod["_shencoder"])):
            out[name] = (timeit(lambda: sh.sh_encode_forward(v, y, B, 3, 4, False, j)),
                         timeit(lambda: sh.sh_encode_forward(v, y, B, 3, 4, True, j)),
                         timeit(lambda: sh.sh_encode_backward(gr, v, B, 3, 4, j, gi)))
        for i, k in enumerate(("sh_fwd", "sh_fwd+jac", "sh_bwd")):
            print(f"sh degree 4, {B} dirs: {k:10s} reference {out['reference'][i]:9.1f}   this library {out['this'][i]:9.1f}   "
                  f"x{out['reference'][i] / out['this'][i]:5.2f}")
    # grid encoder (gridencoder.cu, see oracle/build_ref.py for how it is built): BASELINE's bound-3 table; points as the
    # training step has them (consecutive samples of 4096 rays) and uniformly random ones (no locality at all)
    ref_ge = br.load("gridencoder")
    offsets, pls = O.grid_offsets(desired_resolution=2048 * 3)
    S = float(np.log2(pls)); L = 16; Cf = 2
    g = torch.Generator(device=DEV).manual_seed(3)
    emb = (torch.rand(int(offsets[-1]), Cf, device=DEV, generator=g) * 2 - 1) * 1e-4
    co = cu(offsets)
    for B, kind in ((133120, "ray-ordered"), (133120, "random"), (2097152, "ray-ordered"), (2097152, "random")):
        if kind == "random":
            x = torch.rand(B, 3, device=DEV, generator=g)
        else:
            K = 65 if B == 133120 else 64
            R = B // K
            o = torch.rand(R, 1, 3, device=DEV, generator=g) * 0.4 + 0.3
            d = torch.nn.functional.normalize(torch.randn(R, 1, 3, device=DEV, generator=g), dim=-1)
            t = (torch.arange(K, device=DEV)[None, :, None] + torch.rand(R, 1, 1, device=DEV, generator=g)) / 1024
            x = (o + d * t).clamp(0, 1).reshape(-1, 3)[:B].contiguous()
        gr = torch.randn(L, B, Cf, device=DEV, generator=g)
        dummy = torch.empty(1, device=DEV)
        out = {}
        for name, ge in (("reference", ref_ge), ("this", prod["_gridencoder"])):
            y = torch.empty(L, B, Cf, device=DEV); gemb = torch.zeros_like(emb)
            out[name] = (timeit(lambda: ge.grid_encode_forward(x, emb, co, y, B, 3, Cf, L, S, 16, False, dummy, 0)),
                         timeit(lambda: ge.grid_encode_backward(gr, x, emb, co, gemb, B, 3, Cf, L, S, 16, False, dummy, dummy, 0)))
        for i, k in enumerate(("grid_fwd", "grid_bwd")):
            print(f"grid L16 F2 T2^19 bound 3, {B} {kind} points: {k:9s} reference {out['reference'][i]:9.1f}   this library "
                  f"{out['this'][i]:9.1f}   x{out['reference'][i] / out['this'][i]:5.2f}")


if __name__ == "__main__":
    main()
