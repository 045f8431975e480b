"""FFMLP / fused fp32 MLP microbenchmark (BASELINE config 5 shape: 307 200 samples, sigma net 32->64x2->16 and
colour net 32->64x3->16, bf16 MFMA).  Reports time, TFLOP/s against the 2.5 PF dense bf16 peak and HBM GB/s."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from enerf_amd.backends import _ffmlp as ff  # noqa: E402
from enerf_amd import _lib as L  # noqa: E402


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
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=0, help="only this batch size (default: 307200 and 2097152)")
    a = ap.parse_args()
    dev = "cuda"
    for dtype in (torch.bfloat16, torch.float16):
        for B in ((a.batch,) if a.batch else (307200, 2 * 1024 * 1024)):
            for name, k in (("sigma", 2), ("color", 3)):
                nW = 64 * (32 + 64 * (k - 1) + 16)
                W = (torch.rand(nW, device=dev) - 0.5).to(dtype)
                x = (torch.rand(B, 32, device=dev) - 0.5).to(dtype)
                out = torch.empty(B, 16, device=dev, dtype=dtype)
                fb = torch.empty(k, B, 64, device=dev, dtype=dtype)
                ib = torch.empty(B, 64, device=dev, dtype=dtype)
                flops = 2.0 * B * (32 * 64 + (k - 1) * 64 * 64 + 64 * 16)
                t_inf = timeit(lambda: ff.ffmlp_inference(x, W, B, 32, 16, 64, k, 0, 6, ib, out))
                t_fwd = timeit(lambda: ff.ffmlp_forward(x, W, B, 32, 16, 64, k, 0, 6, fb, out))
                g = torch.randn(B, 16, device=dev).to(dtype)
                bb = torch.zeros(k, B, 64, device=dev, dtype=dtype)
                gi = torch.zeros(B, 32, device=dev, dtype=dtype)
                gw = torch.zeros_like(W)
                t_bwd = timeit(lambda: ff.ffmlp_backward(g, x, W, fb, B, 32, 16, 64, k, 0, 6, True, bb, gi, gw))
                # the reference's data flow (forward_buffer / backward_buffer written and read back) beside the default
                prev = L.lib().enerf_ffmlp_recompute(0)
                t_fwd_b = timeit(lambda: ff.ffmlp_forward(x, W, B, 32, 16, 64, k, 0, 6, fb, out))
                t_bwd_b = timeit(lambda: ff.ffmlp_backward(g, x, W, fb, B, 32, 16, 64, k, 0, 6, True, bb, gi, gw))
                L.lib().enerf_ffmlp_recompute(prev)
                by_inf = B * (64 + 32)
                by_fwd = B * (64 + 32 + k * 128)
                print(f"{str(dtype)[6:]:9s} B={B:8d} {name}: inference {t_inf*1e3:7.1f} us "
                      f"({flops/t_inf/1e9:7.1f} TFLOP/s = {flops/t_inf/1e9/2500*100:4.1f}% of 2.5 PF, "
                      f"{by_inf/t_inf/1e6:6.0f} GB/s) | train fwd {t_fwd*1e3:7.1f} us ({by_fwd/t_fwd/1e6:6.0f} GB/s) | "
                      f"bwd {t_bwd*1e3:7.1f} us ({3*flops/t_bwd/1e9:6.1f} TFLOP/s useful) | buffered flow: fwd "
                      f"{t_fwd_b*1e3:7.1f} us, bwd {t_bwd_b*1e3:7.1f} us ({B*(64+32+k*128)/t_fwd_b/1e6:5.0f} / "
                      f"{B*(2*k*128+k*128+64+64+32)/t_bwd_b/1e6:5.0f} GB/s)")
    # fused fp32 MLP
    lib = L.lib()
    for B in ((a.batch,) if a.batch else (131072, 2 * 1024 * 1024)):
        for name, nh, od in (("sigma32", 1, 16), ("color32", 2, 3)):
            nW = 64 * 32 + (nh - 1) * 4096 + od * 64
            W = torch.rand(nW, device=dev) - 0.5
            x = torch.rand(B, 32, device=dev) - 0.5
            y = torch.empty(B, od, device=dev)
            fb = torch.empty(nh, B, 64, device=dev)
            s = L.stream_handle()
            t = timeit(lambda: lib.enerf_mlp32_forward(x.data_ptr(), W.data_ptr(), B, 32, od, nh, 0, 6, fb.data_ptr(),
                                                       y.data_ptr(), 0, 0, None, s))
            flops = 2.0 * B * (32 * 64 + (nh - 1) * 4096 + 64 * 32)
            print(f"fp32 B={B:8d} {name}: train fwd {t*1e3:7.1f} us ({flops/t/1e9:6.1f} TFLOP/s of 157 peak)")


if __name__ == "__main__":
    main()
