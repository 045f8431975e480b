"""Mint tests/golden/ref_kernels_gfx950.npz: inputs and outputs of the REFERENCE's own kernels (oracle/_ref, see
oracle/build_ref.py) run on an MI355X.  TEST INFRASTRUCTURE ONLY.  Run on the GPU box:

    gpurun -- 'python -B oracle/mint_ref_gpu.py gpurun_out/ref_kernels_gfx950.npz'

and copy the file to tests/golden/.  The CPU suite (tests/test_oracle_vs_ref_kernels_golden.py) then holds the C oracle
to these arrays: bit for bit on everything the marcher emits, 1e-5 on compositing / SH.  Data only: arrays the kernels
read and wrote, in ray order (the reference assigns rows by atomics; the canonical per-ray order is what is stored).
"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
DEV = "cuda"
H = 128


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def per_ray(rays, N):
    t = rays[np.argsort(rays[:, 0], kind="stable")]
    assert np.array_equal(t[:, 0], np.arange(N))
    return t


def gather(buf, table):
    rows = [np.arange(o, o + n) for _, o, n in table if n > 0]
    return buf[np.concatenate(rows)] if rows else buf[:0]


def main(out):
    from oracle import build_ref as br
    from util import synthetic_density_grid, camera_rays
    rm, sh = br.load("raymarching"), br.load("shencoder")
    z = {}
    for bound in (2, 3):
        C = 1 + math.ceil(math.log2(bound))
        grid = synthetic_density_grid(bound, H)
        bits = torch.empty(C * H ** 3 // 8, dtype=torch.uint8, device=DEV)
        rm.packbits(cu(grid.reshape(-1)), C * H ** 3 // 8, 0.01, bits)
        N = 96
        o, d = camera_rays(N, 500 + bound, bound)
        d[5] = (1.0, 0.0, 0.0); o[6] = (9.0, 9.0, 9.0)
        aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
        nears, fars = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
        rm.near_far_from_aabb(cu(o), cu(d), cu(aabb), N, 0.2, nears, fars)
        k = f"b{bound}_"
        z[k + "bits"] = bits.cpu().numpy(); z[k + "o"] = o; z[k + "d"] = d
        z[k + "nears"] = nears.cpu().numpy(); z[k + "fars"] = fars.cpu().numpy()
        for tag, dt_gamma, perturb in (("plain", 0.0, 0), ("gamma", 1.0 / 256, 0), ("jitter", 0.0, 1)):
            M = N * 1024
            xyzs, dirs = torch.zeros(M, 3, device=DEV), torch.zeros(M, 3, device=DEV)
            deltas = torch.zeros(M, 2, device=DEV)
            rays = torch.full((N, 3), -1, dtype=torch.int32, device=DEV)
            counter = torch.zeros(2, dtype=torch.int32, device=DEV)
            rm.march_rays_train(cu(o), cu(d), bits, float(bound), dt_gamma, 1024, N, C, H, M, nears, fars, xyzs, dirs,
                                deltas, rays, counter, perturb)
            t = per_ray(rays.cpu().numpy(), N)
            z[k + tag + "_counts"] = t[:, 2].copy()
            z[k + tag + "_counter"] = counter.cpu().numpy()
            z[k + tag + "_xyzs"] = gather(xyzs.cpu().numpy(), t)
            z[k + tag + "_deltas"] = gather(deltas.cpu().numpy(), t)
            if tag == "plain":
                # compositing on the canonical (ray-ordered) sample list
                tot = int(t[:, 2].sum()); m = tot + 128 - tot % 128
                off = np.concatenate([[0], np.cumsum(t[:, 2])[:-1]]).astype(np.int32)
                canon = np.stack([np.arange(N, dtype=np.int32), off, t[:, 2]], 1).astype(np.int32)
                dl = np.zeros((m, 2), np.float32); dl[:tot] = z[k + tag + "_deltas"]
                rng = np.random.default_rng(bound)
                sig = (rng.random(m) * 25).astype(np.float32); rgb = rng.random((m, 3)).astype(np.float32)
                ws, dp, im = torch.empty(N, device=DEV), torch.empty(N, device=DEV), torch.empty(N, 3, device=DEV)
                rm.composite_rays_train_forward(cu(sig), cu(rgb), cu(dl), cu(canon), m, N, ws, dp, im)
                g_ws = rng.standard_normal(N).astype(np.float32); g_im = rng.standard_normal((N, 3)).astype(np.float32)
                gs, gc = torch.zeros(m, device=DEV), torch.zeros(m, 3, device=DEV)
                rm.composite_rays_train_backward(cu(g_ws), cu(g_im), cu(sig), cu(rgb), cu(dl), cu(canon), ws, im, m, N,
                                                 gs, gc)
                for n, v in (("sig", sig), ("rgb", rgb), ("g_ws", g_ws), ("g_im", g_im), ("ws", ws), ("depth", dp),
                             ("image", im), ("gs", gs), ("gc", gc)):
                    z[k + "comp_" + n] = v.cpu().numpy() if isinstance(v, torch.Tensor) else v
        # one inference round: 8 steps for every ray from its near plane, jittered and not
        for tag, perturb in (("inf", 0), ("infj", 3)):
            alive = torch.arange(N, dtype=torch.int32, device=DEV)
            rt = nears.clone()
            Mi = N * 8
            gx, gd, gl = torch.zeros(Mi, 3, device=DEV), torch.zeros(Mi, 3, device=DEV), torch.zeros(Mi, 2, device=DEV)
            rm.march_rays(N, 8, alive, rt, cu(o), cu(d), float(bound), 0.0, 1024, C, H, bits, nears, fars, gx, gd, gl, perturb)
            z[k + tag + "_xyzs"] = gx.cpu().numpy(); z[k + tag + "_deltas"] = gl.cpu().numpy()
            if perturb == 0:
                rng = np.random.default_rng(10 + bound)
                sig = (rng.random(Mi) * 30).astype(np.float32); rgb = rng.random((Mi, 3)).astype(np.float32)
                ws, dp, im = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV), torch.zeros(N, 3, device=DEV)
                rm.composite_rays(N, 8, alive, rt, cu(sig), cu(rgb), gl, ws, dp, im)
                for n, v in (("sig", sig), ("rgb", rgb), ("ws", ws), ("depth", dp), ("image", im), ("rt", rt)):
                    z[k + tag + "_" + n] = v.cpu().numpy() if isinstance(v, torch.Tensor) else v
    g = torch.Generator(device=DEV).manual_seed(4)
    v = torch.randn(96, 3, generator=g, device=DEV)
    v[:48] = torch.nn.functional.normalize(v[:48], dim=-1)
    z["sh_dirs"] = v.cpu().numpy()
    for degree in range(1, 9):
        y = torch.empty(96, degree * degree, device=DEV); j = torch.empty(96, 3 * degree * degree, device=DEV)
        sh.sh_encode_forward(v, y, 96, 3, degree, True, j)
        grad = torch.randn(96, degree * degree, generator=g, device=DEV)
        gi = torch.zeros(96, 3, device=DEV)              # the kernel accumulates; the wrapper hands it zeros
        sh.sh_encode_backward(grad, v, 96, 3, degree, j, gi)
        z[f"sh{degree}_y"] = y.cpu().numpy(); z[f"sh{degree}_dy_dx"] = j.cpu().numpy()
        z[f"sh{degree}_grad"] = grad.cpu().numpy(); z[f"sh{degree}_gi"] = gi.cpu().numpy()
    torch.cuda.synchronize()
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    np.savez_compressed(out, **z)
    print(f"[mint_ref_gpu] {out}: {len(z)} arrays, {os.path.getsize(out) / 1024:.0f} KiB")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/ref_kernels_gfx950.npz")
