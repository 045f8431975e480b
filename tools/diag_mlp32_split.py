"""Accuracy of the mlp32 kernels' two arithmetic modes against an fp64 statement of the same stack (development aid):
forward error, ReLU decisions that differ from fp64's (a pre-activation within the forward error of zero), and the
gradient errors with those samples set aside.   python tools/diag_mlp32_split.py [--B 5000]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from enerf_amd import _lib as L  # noqa: E402
from enerf_amd.fused_mlp import pad32  # noqa: E402


def fb_rows(fb, nh, Bp):
    """tile-native forward buffer -> [nh, Bp, 64] row-major (csrc/mlp32.hip store_tile_fb)."""
    t = fb.view(nh, Bp // 32, 2, 4, 2, 32, 4)            # layer, tile, ib, g, h, j, r
    return t.permute(0, 1, 5, 2, 3, 4, 6).reshape(nh, Bp, 64)   # neuron = 32 ib + 8 g + 4 h + r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=5000)
    a = ap.parse_args()
    lib, s, dev = L.lib(), L.stream_handle(), "cuda"
    lib.enerf_mlp32_recompute(0)             # this tool reads the forward buffer (the default leaves it unwritten)
    torch.manual_seed(5)
    for name, nh, out, xl in (("sigma 32-64-16 xl=1", 1, 16, 1), ("colour 32-64-64-3 xl=0", 2, 3, 0),
                              ("32-64-16 xl=0", 1, 16, 0)):
        B, Bp = a.B, pad32(a.B)
        dims = [32] + [64] * nh + [out]
        ws = [(torch.rand(dims[k + 1], dims[k], device=dev) * 2 - 1) * (3.0 / dims[k]) ** 0.5 for k in range(len(dims) - 1)]
        W = torch.cat([w.reshape(-1) for w in ws]).contiguous()
        xr = torch.rand(B, 32, device=dev) * 2 - 1
        if xl:
            X = torch.zeros(16, Bp, 2, device=dev)
            X[:, :B] = xr.view(B, 16, 2).permute(1, 0, 2)
        else:
            X = xr.contiguous()
        dY = torch.randn(B, out, device=dev)
        # fp64 reference
        h = xr.double()
        acts, pres = [], []
        for k, w in enumerate(ws):
            h = h @ w.double().t()
            if k < nh:
                pres.append(h)
                h = torch.relu(h)
                acts.append(h)
        yref = h
        for mode in (0, 1):
            lib.enerf_mlp32_precision(mode)
            fb = torch.zeros(nh * Bp * 64, device=dev)
            bb = torch.empty(nh, Bp, 64, device=dev)
            Y = torch.empty(B, out, device=dev)
            dX = torch.zeros_like(X)
            dW = torch.zeros_like(W)
            L.check(lib.enerf_mlp32_forward(X.data_ptr(), W.data_ptr(), B, 32, out, nh, 0, 6, fb.data_ptr(), Y.data_ptr(), xl,
                                            0, None, s), "fwd")
            L.check(lib.enerf_mlp32_backward(dY.data_ptr(), X.data_ptr(), W.data_ptr(), fb.data_ptr(), B, 32, out, nh, 0,
                                             bb.data_ptr(), dX.data_ptr(), dW.data_ptr(), xl, 0, None, 0, None, None, 0, s),
                    "bwd")
            torch.cuda.synchronize()
            fr = fb_rows(fb, nh, Bp)[:, :B]
            flips = torch.zeros(B, dtype=torch.bool, device=dev)
            nflip = 0
            for l in range(nh):
                d = (fr[l] > 0) != (pres[l] > 0)
                nflip += int(d.sum())
                flips |= d.any(dim=1)
                aerr = float((fr[l].double() - acts[l]).abs().max())
                print(f"  [{name} mode {mode}] layer {l}: max |act - fp64| = {aerr:.2e} (scale {float(acts[l].abs().max()):.2f}), "
                      f"ReLU decisions differing from fp64: {int(d.sum())}")
            # gradients in fp64 WITH the device's ReLU decisions (what a correct kernel computes given its own masks)
            g = dY.double()
            gws = [None] * len(ws)
            ins = [xr.double()] + acts
            for k in range(len(ws) - 1, -1, -1):
                gws[k] = g.t() @ ins[k]
                g = g @ ws[k].double()
                if k > 0:
                    g = g * (fr[k - 1] > 0)
            dxr = (dX[:, :B].permute(1, 0, 2).reshape(B, 32) if xl else dX).double()
            print(f"  [{name} mode {mode}] y: max err {float((Y.double() - yref).abs().max()):.2e} / scale "
                  f"{float(yref.abs().max()):.2f};  dX (own masks): max err {float((dxr - g).abs().max()):.2e} / max "
                  f"{float(g.abs().max()):.2f};  samples with a flipped ReLU: {int(flips.sum())} ({nflip} units)")
            off = 0
            for k, w in enumerate(ws):
                got = dW[off:off + w.numel()].view_as(w).double()
                off += w.numel()
                print(f"      dW{k}: max err {float((got - gws[k]).abs().max()):.2e} / max {float(gws[k].abs().max()):.2e} "
                      f"= {float((got - gws[k]).abs().max() / gws[k].abs().max()):.2e}")
    lib.enerf_mlp32_precision(1)
    lib.enerf_mlp32_recompute(1)


if __name__ == "__main__":
    main()
