"""march_rays_train, fixed step: the wave-per-ray lattice marcher (chunk log) against the one-thread-per-ray walk (run
log) over the ray counts of the path -- count + scan, write, with and without the occupied-box test -- on the training
cameras (random pixels of one pose) and on a whole 640 x 480 frame (coherent rays).  Outputs of the two routes are compared
bit for bit while they are timed.      gpurun -- 'python tools/march_route_sweep.py > gpurun_out/march_route_sweep.txt'"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from enerf_amd import _lib, scene                          # noqa: E402
from enerf_amd.backends import _raymarching as rb          # noqa: E402
from enerf_amd.network import NeRFNetwork                  # noqa: E402

DEV = "cuda"


def timeit(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    lib = _lib.lib()
    H = 128
    print("us per call; c = count + scan, w = write; box = occupied-box test on (what the training step and the frame use)")
    for bound, kind, N in ((3, "train", 4096), (3, "train", 8192), (3, "train", 16384), (3, "train", 32768),
                           (3, "train", 65536), (3, "train", 131072), (2, "frame", 307200), (2, "train", 307200)):
        C = 1 + math.ceil(math.log2(bound))
        m = NeRFNetwork(encoding="hashgrid", bound=bound, cuda_ray=True, out_dim_color=3).to(DEV)
        scene.install_occupancy(m)
        bits = m.density_bitfield
        if kind == "frame":
            inds = torch.arange(scene.H * scene.W, device=DEV)[:N]
            ro, rd = scene.pixel_rays(scene.pose(3), inds, DEV)
        else:
            g = torch.Generator(device=DEV).manual_seed(7)
            (ro, rd), _ = scene.training_batch(0, N, DEV, generator=g)
        ro, rd = ro.reshape(-1, 3).contiguous(), rd.reshape(-1, 3).contiguous()
        nears, fars = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
        aabb = torch.tensor([-bound] * 3 + [bound] * 3, dtype=torch.float32, device=DEV)
        rb.near_far_from_aabb(ro, rd, aabb, N, 0.2, nears, fars)
        rb.occupied_box_update(bits, C, H, bound)
        M = N * 160
        outs = {}
        line = f"bound {bound} {kind:5s} {N:7d} rays:"
        for route, thr in (("wave", 0x7fffffff), ("thread", 1)):
            lib.enerf_debug_march_thread_min_rays(thr)
            for box in (0, 4):
                rays = torch.empty(N, 3, dtype=torch.int32, device=DEV)
                counter = torch.zeros(2, dtype=torch.int32, device=DEV)
                xyzs, dirs, deltas = (torch.empty(M, 3, device=DEV), torch.empty(M, 3, device=DEV), torch.empty(M, 2, device=DEV))
                args = (ro, rd, bits, bound, 0.0, 1024, N, C, H)

                def count():
                    rb.march_rays_train_count(*args, nears, fars, rays, counter, 1, box | 8)

                def write():
                    rb.march_rays_train_write(*args, M, nears, fars, xyzs, dirs, deltas, rays, counter, 1, 1)
                tc = timeit(count)
                tw = timeit(write)
                tot = int(counter[0])
                assert tot + 128 < M, (tot, M)
                outs[(route, box)] = (rays.clone(), xyzs[:tot].clone(), deltas[:tot].clone(), tot)
                line += f"  {route}{'+box' if box else ''}: c {tc:7.1f} w {tw:6.1f}"
        ref = outs[("wave", 0)]
        for k, v in outs.items():
            assert v[3] == ref[3] and torch.equal(v[0], ref[0]) and torch.equal(v[1], ref[1]) and torch.equal(v[2], ref[2]), k
        print(line + f"   ({ref[3]} samples, routes bit-identical)")
        del m
    lib.enerf_debug_march_thread_min_rays(65536)


if __name__ == "__main__":
    main()
