"""Microbenchmark of the fused fp32 MLP kernels at training-batch size (development aid).
Usage: python tools/bench_mlp32.py [--B 137856]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from enerf_amd import _lib as L  # noqa: E402
from enerf_amd.fused_mlp import pad32  # noqa: E402


def timeit(fn, n=30, warm=5):
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=137856 - 5)
    ap.add_argument("--wgrad-blocks", type=int, default=0)
    ap.add_argument("--unfused-backward", action="store_true")
    ap.add_argument("--fwd-blocks", type=int, default=0)
    ap.add_argument("--bwd-blocks", type=int, default=0)
    ap.add_argument("--all-shapes", action="store_true")
    ap.add_argument("--recompute", type=int, default=1, help="enerf_mlp32_recompute: 1 = the backward recomputes the activations")
    ap.add_argument("--precision", type=int, default=1, help="enerf_mlp32_precision: 0 = fp32 MFMA, 1 = split-bf16")
    a = ap.parse_args()
    L.lib().enerf_mlp32_precision(a.precision)
    L.lib().enerf_mlp32_recompute(a.recompute)
    if a.unfused_backward:
        L.lib().enerf_debug_mlp32_fused_backward(0)
    L.lib().enerf_debug_mlp32_grid_caps(a.fwd_blocks, a.bwd_blocks)
    if a.wgrad_blocks:
        L.lib().enerf_debug_mlp32_wgrad_blocks(a.wgrad_blocks)
    B, Bp, dev = a.B, pad32(a.B), "cuda"
    lib = L.lib()
    s = L.stream_handle()
    cfgs = (("sigma 32-64-16 (level-major x)", 1, 16, 1), ("color 32-64-64-3", 2, 3, 0))
    if a.all_shapes:
        cfgs = tuple((f"nh={nh} out={out} xl={xl}", nh, out, xl) for nh in (1, 2, 3) for out in (3, 16) for xl in (0, 1))
    for name, nh, out, xl in cfgs:
        nw = 64 * 32 + (nh - 1) * 64 * 64 + out * 64
        W = (torch.rand(nw, device=dev) - 0.5) * 0.3
        X = torch.rand(16, Bp, 2, device=dev) if xl else torch.rand(B, 32, device=dev)
        fb = torch.empty(nh, Bp, 64, device=dev)
        bb = torch.empty(nh, Bp, 64, device=dev)
        Y = torch.empty(B, out, device=dev)
        dY = torch.randn(B, out, device=dev)
        dX = torch.empty_like(X)
        dW = torch.zeros(nw, device=dev)
        t_inf = timeit(lambda: lib.enerf_mlp32_forward(X.data_ptr(), W.data_ptr(), B, 32, out, nh, 0, 6, None, Y.data_ptr(), xl, 0, None, s))
        t_fwd = timeit(lambda: lib.enerf_mlp32_forward(X.data_ptr(), W.data_ptr(), B, 32, out, nh, 0, 6, fb.data_ptr(), Y.data_ptr(), xl, 0, None, s))
        t_bwd = timeit(lambda: lib.enerf_mlp32_backward(dY.data_ptr(), X.data_ptr(), W.data_ptr(), fb.data_ptr(), B, 32, out, nh, 0,
                                                        bb.data_ptr(), dX.data_ptr(), dW.data_ptr(), xl, 0, None, 0, None, None, 0, s))
        macs = 32 * 64 + (nh - 1) * 64 * 64 + 64 * out
        print(f"{name:34s} B={B}: inference {t_inf:6.1f} us | train fwd {t_fwd:6.1f} us | bwd (act+w+reduce) {t_bwd:6.1f} us"
              f" | fwd {2 * macs * B / t_fwd / 1e6:6.1f} TFLOP/s, bwd {4 * macs * B / t_bwd / 1e6:6.1f} TFLOP/s")
        if hasattr(lib, "enerf_debug_mlp_phases"):      # built with ENERF_DEFINES=-DENERF_MLP_TIMING
            import ctypes

            def phases(title, call, names, reps=20):
                arr = (ctypes.c_ulonglong * 16)()
                lib.enerf_debug_mlp_phases(arr, 1)
                for _ in range(reps):
                    call()
                torch.cuda.synchronize()
                lib.enerf_debug_mlp_phases(arr, 0)
                v = [float(x) for x in arr]
                nwg, tot = v[15] / reps, sum(v[:len(names)])
                print(f"   {title}: {nwg:.0f} workgroups, s_memtime ticks per wave and share")
                for k, nm in enumerate(names):
                    print(f"     {nm:20s} {v[k] / reps / nwg / 4:10.0f}  {100 * v[k] / tot:5.1f} %")

            phases("training forward", lambda: lib.enerf_mlp32_forward(X.data_ptr(), W.data_ptr(), B, 32, out, nh, 0, 6,
                                                                       fb.data_ptr(), Y.data_ptr(), xl, 0, None, s),
                   ["weights -> LDS", "fragments", "x + layer 1", "hidden", "out + stores", "exit"])
            phases("fused backward", lambda: lib.enerf_mlp32_backward(dY.data_ptr(), X.data_ptr(), W.data_ptr(),
                                                                      fb.data_ptr(), B, 32, out, nh, 0, bb.data_ptr(),
                                                                      dX.data_ptr(), dW.data_ptr(), xl, 0, None, 0, None,
                                                                      None, 0, s),
                   ["set-up / loop", "issue loads", "out dgrad (+wait)", "dWout", "hidden dgrad", "hidden wgrad", "X tile",
                    "dW0 (+wait X)", "input dgrad + dX", "acc -> LDS", "partial store"])
    print("kernel split via rocprofv3 --kernel-trace --stats")


if __name__ == "__main__":
    main()
