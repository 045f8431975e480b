"""Full-frame (640x480) inference loop timing: python tools/bench_render.py [--net ff] [--mult 8] [--frames 5]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from enerf_amd import scene  # noqa: E402
from enerf_amd.backends import _raymarching as rb  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--net", default="linear")
    ap.add_argument("--bound", type=int, default=2)
    ap.add_argument("--mult", type=int, default=8)
    ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--wave-max-rays", type=int, default=0, help="tuning: wave-per-ray march up to this many rays")
    a = ap.parse_args()
    if a.net == "ff":
        from enerf_amd.network_ff import NeRFNetwork
    else:
        from enerf_amd.network import NeRFNetwork
    dev = "cuda"
    if a.wave_max_rays:
        from enerf_amd import _lib
        _lib.lib().enerf_debug_march_wave_max_rays(a.wave_max_rays)
    torch.manual_seed(0)
    m = NeRFNetwork(encoding="hashgrid", bound=a.bound, cuda_ray=True, out_dim_color=3).to(dev).eval()
    scene.install_occupancy(m)
    m.infer_batch_mult = a.mult
    inds = torch.arange(scene.H * scene.W, device=dev)
    ro, rd = scene.pixel_rays(scene.pose(3), inds, dev)
    with torch.no_grad():
        m.render(ro, rd, staged=False, bg_color=None, perturb=False)
        torch.cuda.synchronize()
        rb.STATS.update(infer_samples=0, infer_calls=0)
        t0 = time.perf_counter()
        for _ in range(a.frames):
            m.render(ro, rd, staged=False, bg_color=None, perturb=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"net={a.net} mult={a.mult}: {dt/a.frames*1e3:.2f} ms/frame, {rb.STATS['infer_samples']/dt/1e6:.0f} Msamples/s, "
          f"{rb.STATS['infer_calls']/a.frames:.0f} iterations/frame, {rb.STATS['infer_samples']/a.frames/1e6:.2f} M samples/frame")


if __name__ == "__main__":
    main()
