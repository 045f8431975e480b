"""Per-kernel microbenchmarks on the GPU (development aid; numbers quoted in DESIGN.md come from here + rocprofv3).
Usage: python tools/bench_kernels.py [--rays 4096] [--bound 3] [--levels]"""
import argparse
import ctypes
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from enerf_amd import _lib, scene, raymarching  # noqa: E402
from enerf_amd.backends import _gridencoder as ge  # noqa: E402
from enerf_amd.gridencoder import GridEncoder  # noqa: E402


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
    return a.elapsed_time(b) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--bound", type=int, default=3)
    ap.add_argument("--levels", action="store_true")
    a = ap.parse_args()
    dev = "cuda"
    bound = a.bound
    C = 1 + math.ceil(math.log2(bound))
    grid = scene.density_grid(bound, dev)
    bits = raymarching.packbits(grid, 0.01)
    g = torch.Generator(device=dev).manual_seed(1)
    (ro, rd), _ = scene.training_batch(0, a.rays, dev, generator=g)
    aabb = torch.tensor([-bound] * 3 + [bound] * 3, dtype=torch.float32, device=dev)
    nears, fars = raymarching.near_far_from_aabb(ro, rd, aabb, 0.2)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(ro, rd, bound, bits, C, 128, nears, fars, counter, -1, True,
                                                            128, True)
    M = xyzs.shape[0]
    print(f"rays={a.rays} samples={int(counter[0])} M={M} occupied={float((rays[:,2]>0).float().mean()):.2f}")

    def march():
        counter.zero_()
        raymarching.march_rays_train(ro, rd, bound, bits, C, 128, nears, fars, counter, M - 128, True, 128, False)
    print(f"march_rays_train (incl. zero-fill of outputs): {timeit(march):.3f} ms")

    enc = GridEncoder(desired_resolution=2048 * bound).to(dev)
    enc.embeddings.data.uniform_(-1, 1)
    x01 = ((xyzs + bound) / (2 * bound)).contiguous()
    xr = torch.rand_like(x01)
    S = float(torch.log2(torch.tensor(enc.per_level_scale, dtype=torch.float64)))
    L, Cc = 16, 2
    dummy = torch.empty(1, device=dev)
    lib = _lib.lib()
    lib.enerf_debug_grid_level_mask.argtypes = [ctypes.c_uint32]
    for name, pts in (("ray samples", x01), ("uniform random", xr)):
        B = pts.shape[0]
        out = torch.empty(B, L * Cc, device=dev)
        gout = torch.randn(B, L * Cc, device=dev)
        gemb = torch.zeros_like(enc.embeddings)

        def fwd():
            ge.grid_encode_forward(pts, enc.embeddings.data, enc.offsets, out, B, 3, Cc, L, S, 16, False, dummy, 0,
                                   layout=1)

        def bwd():
            ge.grid_encode_backward(gout, pts, enc.embeddings.data, enc.offsets, gemb, B, 3, Cc, L, S, 16, False, dummy,
                                    dummy, 0, layout=1)
        tf, tb = timeit(fwd), timeit(bwd)
        print(f"[{name}] B={B} grid fwd {tf:.3f} ms ({B*1164/tf/1e6:.0f} GB/s alg) | bwd {tb:.3f} ms "
              f"({B*1164/tb/1e6:.0f} GB/s alg)")
        if a.levels:
            for l in range(L):
                lib.enerf_debug_grid_level_mask(1 << l)
                print(f"    level {l:2d}: fwd {timeit(fwd, 10):.4f} ms  bwd {timeit(bwd, 10):.4f} ms")
            lib.enerf_debug_grid_level_mask(0xffffffff)
        out0 = torch.empty(L, B, Cc, device=dev)

        def fwd0():
            ge.grid_encode_forward(pts, enc.embeddings.data, enc.offsets, out0, B, 3, Cc, L, S, 16, False, dummy, 0,
                                   layout=0)
        print(f"    layout [L,B,C]: fwd {timeit(fwd0):.3f} ms")
    # big batch (density-grid update size)
    Bb = 2 * 1024 * 1024
    pb = torch.rand(Bb, 3, device=dev)
    outb = torch.empty(Bb, L * Cc, device=dev)
    tfb = timeit(lambda: ge.grid_encode_forward(pb, enc.embeddings.data, enc.offsets, outb, Bb, 3, Cc, L, S, 16, False,
                                                dummy, 0, layout=1), 5)
    print(f"[2M uniform] grid fwd {tfb:.3f} ms ({Bb*1164/tfb/1e6:.0f} GB/s alg)")

    # compositing
    sig = torch.rand(M, device=dev) * 20
    rgb = torch.rand(M, 3, device=dev)
    from enerf_amd.backends import _raymarching as rb
    N = a.rays
    ws = torch.empty(N, device=dev); dp = torch.empty(N, device=dev); im = torch.empty(N, 3, device=dev)
    tcf = timeit(lambda: rb.composite_rays_train_forward(sig, rgb, deltas, rays, M, N, ws, dp, im))
    gs = torch.zeros(M, device=dev); gc = torch.zeros(M, 3, device=dev)
    gw = torch.randn(N, device=dev); gi = torch.randn(N, 3, device=dev)
    tcb = timeit(lambda: rb.composite_rays_train_backward(gw, gi, sig, rgb, deltas, rays, ws, im, M, N, gs, gc))
    print(f"composite fwd {tcf*1e3:.1f} us, bwd {tcb*1e3:.1f} us")


if __name__ == "__main__":
    main()
