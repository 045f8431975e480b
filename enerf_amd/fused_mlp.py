"""Fused fp32 evaluation of an nn.Linear / ReLU stack (64-wide hidden layers, no bias) on the MI355X matrix cores.

`fused_mlp(x, weights, activation)` computes exactly what `for W in weights: x = relu(x @ W.T)` (last layer linear)
computes, through enerf_mlp32_forward / enerf_mlp32_backward (csrc/mlp32.hip: v_mfma_f32_32x32x2_f32, exact fp32 fma
chains), with autograd support for x and every weight.  The nn.Linear modules keep owning the parameters, so
state_dicts are unchanged.  Used by enerf_amd.network.NeRFNetwork for CUDA fp32 inputs; anything else (CPU tensors in
the oracle-backed tests, autocast) takes the plain torch path.
"""
import torch
from torch.autograd import Function

from . import _lib as L


def supported(x, weights):
    if not (x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled()):
        return False
    n = len(weights)
    if n < 2 or n > 4:
        return False
    if weights[0].shape[0] != 64 or weights[0].shape[1] > 32 or weights[-1].shape[0] > 32:
        return False
    if x.shape[1] != weights[0].shape[1] and x.shape[1] != 32:
        return False
    return all(tuple(w.shape) == (64, 64) for w in weights[1:-1]) and weights[-1].shape[1] == 64


class _FusedMLP32(Function):
    @staticmethod
    def forward(ctx, x, activation, *weights):
        B0 = x.shape[0]
        in_dim = weights[0].shape[1]          # x may already be padded to 32 columns (extra columns are ignored)
        num_hidden = len(weights) - 1
        out_dim = weights[-1].shape[0]
        dev = x.device
        # pad the batch to a multiple of 32 and the input width to 32 (zero columns / zero weight columns)
        B = (B0 + 31) // 32 * 32
        if B != B0 or x.shape[1] != 32 or not x.is_contiguous():
            xp = torch.zeros(B, 32, dtype=torch.float32, device=dev)
            xp[:B0, :in_dim] = x[:, :in_dim]
        else:
            xp = x
        w0 = weights[0]
        if in_dim != 32:
            w0 = torch.nn.functional.pad(w0, (0, 32 - in_dim))
        blob = torch.cat([w0.reshape(-1)] + [w.reshape(-1) for w in weights[1:]]).contiguous()
        # hidden activations are only written out when a backward pass can follow (under no_grad every
        # needs_input_grad entry is False)
        train = any(ctx.needs_input_grad)
        fb = torch.empty(num_hidden, B, 64, dtype=torch.float32, device=dev) if train else None
        y = torch.empty(B, out_dim, dtype=torch.float32, device=dev)
        L.check(L.lib().enerf_mlp32_forward(xp.data_ptr(), blob.data_ptr(), B, 32, out_dim, num_hidden, activation, 6,
                                            fb.data_ptr() if fb is not None else None, y.data_ptr(),
                                            L.stream_handle()), "mlp32_forward")
        if train:
            ctx.save_for_backward(xp, blob, fb)
            ctx.meta = (B0, in_dim, out_dim, num_hidden, activation, [tuple(w.shape) for w in weights],
                        ctx.needs_input_grad[0], x.shape[1])
        return y[:B0] if B != B0 else y

    @staticmethod
    def backward(ctx, gy):
        xp, blob, fb = ctx.saved_tensors
        B0, in_dim, out_dim, num_hidden, activation, shapes, need_dx, x_cols = ctx.meta
        B = xp.shape[0]
        dev = xp.device
        if B != B0:
            g = torch.zeros(B, out_dim, dtype=torch.float32, device=dev)
            g[:B0] = gy
        else:
            g = gy.float().contiguous()
        bb = torch.empty(num_hidden, B, 64, dtype=torch.float32, device=dev)
        dx = torch.empty(B, 32, dtype=torch.float32, device=dev) if need_dx else None
        dw = torch.zeros_like(blob)
        L.check(L.lib().enerf_mlp32_backward(g.data_ptr(), xp.data_ptr(), blob.data_ptr(), fb.data_ptr(), B, 32,
                                             out_dim, num_hidden, activation, bb.data_ptr(),
                                             dx.data_ptr() if dx is not None else None, dw.data_ptr(),
                                             L.stream_handle()), "mlp32_backward")
        grads, off = [], 0
        for k, shp in enumerate(shapes):
            if k == 0:
                gw = dw[off:off + 64 * 32].view(64, 32)[:, :shp[1]]
                off += 64 * 32
            else:
                n = shp[0] * shp[1]
                gw = dw[off:off + n].view(shp)
                off += n
            grads.append(gw)
        gx = dx[:B0, :x_cols] if need_dx else None      # columns >= in_dim carry zero weight -> zero gradient
        return (gx, None) + tuple(grads)


def fused_mlp(x, weights, activation="relu"):
    """x [B, in<=32] fp32 CUDA; weights = list of [out, in] matrices (hidden width 64, last out <= 32)."""
    return _FusedMLP32.apply(x, 0 if activation == "relu" else 6, *weights)
