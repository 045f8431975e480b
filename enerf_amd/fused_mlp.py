"""Fused fp32 evaluation of an nn.Linear / ReLU stack (64-wide hidden layers, no bias) on the MI355X matrix cores.

`fused_mlp(x, weights, activation)` computes exactly what `for W in weights: x = relu(x @ W.T)` (last layer linear)
computes, through enerf_mlp32_forward / enerf_mlp32_backward (csrc/mlp32.hip: v_mfma_f32_32x32x2_f32, exact fp32 fma
chains), with autograd support for x and every weight.  The nn.Linear modules keep owning the parameters, so
state_dicts are unchanged.  Used by enerf_amd.network.NeRFNetwork for CUDA fp32 inputs; anything else (CPU tensors in
the oracle-backed tests, autocast) takes the plain torch path.

The input is either the usual [B, in<=32] matrix or, with `x_layout=1`, the level-major [16, Bp, 2] tensor the grid
encoder produces with `layout=2` (Bp = B rounded up to 32): the encoding then goes from the gather kernel into the
matrix cores, and its gradient back into the scatter kernel, without ever being transposed into rows.
Batches are ragged on the device side -- nothing is padded or copied here.
"""
import torch
from torch.autograd import Function

from . import _lib as L


def _weights_ok(weights):
    n = len(weights)
    if n < 2 or n > 4:
        return False
    if weights[0].shape[0] != 64 or weights[0].shape[1] > 32 or weights[-1].shape[0] > 32:
        return False
    return all(tuple(w.shape) == (64, 64) for w in weights[1:-1]) and weights[-1].shape[1] == 64


# False: network.py's nets run as the plain nn.Linear loop of the reference (torch's GEMMs) -- what the reference-native
# comparison routes of tests/refcheck/ set
ENABLED = True


def supported(x, weights):
    """Row-major input [B, in] (in == weights[0].shape[1], or already 32 columns wide)."""
    if not (ENABLED and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled()):
        return False
    if not _weights_ok(weights):
        return False
    return x.shape[1] == weights[0].shape[1] or x.shape[1] == 32


def supported_level_major(device, dtype, weights):
    """Level-major input: the first layer must take exactly 32 columns (16 levels x 2 features)."""
    return (ENABLED and device.type == "cuda" and dtype == torch.float32 and not torch.is_autocast_enabled()
            and _weights_ok(weights) and weights[0].shape[1] == 32)


def pad32(n):
    return (n + 31) // 32 * 32


class _FusedMLP32(Function):
    @staticmethod
    def forward(ctx, x, activation, x_layout, batch, train, *weights):
        in_dim = weights[0].shape[1]
        num_hidden = len(weights) - 1
        out_dim = weights[-1].shape[0]
        dev = x.device
        if x_layout == 1:
            B0 = batch
            assert x.dim() == 3 and x.shape[0] == 16 and x.shape[2] == 2 and x.shape[1] == pad32(B0) and in_dim == 32
            xp = x.contiguous()
        else:
            B0 = x.shape[0]
            # rows narrower than 32 columns are widened once (zero columns meet zero weight columns)
            if x.shape[1] != 32 or not x.is_contiguous():
                xp = torch.zeros(B0, 32, dtype=torch.float32, device=dev)
                xp[:, :in_dim] = x[:, :in_dim]
            else:
                xp = x
        Bp = pad32(B0)
        w0 = weights[0]
        if in_dim != 32:
            w0 = torch.nn.functional.pad(w0, (0, 32 - in_dim))
        blob = torch.cat([w0.reshape(-1)] + [w.reshape(-1) for w in weights[1:]]).contiguous()
        # `train` (decided by the caller, where grad mode is still visible): hidden activations are only written out
        # when a backward pass can follow
        fb = torch.empty(num_hidden, Bp, 64, dtype=torch.float32, device=dev) if train else None
        y = torch.empty(B0, out_dim, dtype=torch.float32, device=dev)
        if B0 > 0:
            L.check(L.lib().enerf_mlp32_forward(xp.data_ptr(), blob.data_ptr(), B0, 32, out_dim, num_hidden,
                                                activation, 6, fb.data_ptr() if fb is not None else None,
                                                y.data_ptr(), x_layout, 0, None, L.stream_handle()), "mlp32_forward")
        if train:
            ctx.save_for_backward(xp, blob, fb)
            ctx.meta = (B0, in_dim, out_dim, num_hidden, activation, [tuple(w.shape) for w in weights],
                        ctx.needs_input_grad[0], x.shape[1], x_layout)
        return y

    @staticmethod
    def backward(ctx, gy):
        xp, blob, fb = ctx.saved_tensors
        B0, in_dim, out_dim, num_hidden, activation, shapes, need_dx, x_cols, x_layout = ctx.meta
        Bp = pad32(B0)
        dev = xp.device
        g = gy.float().contiguous()
        bb = torch.empty(num_hidden, Bp, 64, dtype=torch.float32, device=dev)
        dx = None
        if need_dx:
            dx = torch.empty(16, Bp, 2, dtype=torch.float32, device=dev) if x_layout == 1 else \
                torch.empty(B0, 32, dtype=torch.float32, device=dev)
        dw = torch.zeros_like(blob)
        if B0 > 0:
            L.check(L.lib().enerf_mlp32_backward(g.data_ptr(), xp.data_ptr(), blob.data_ptr(), fb.data_ptr(), B0, 32,
                                                 out_dim, num_hidden, activation, bb.data_ptr(),
                                                 dx.data_ptr() if dx is not None else None, dw.data_ptr(), x_layout,
                                                 0, None, 0, None, None, 0, L.stream_handle()), "mlp32_backward")
        elif dx is not None:
            dx.zero_()
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
        gx = None
        if need_dx:       # row-major: columns >= in_dim carry zero weight -> zero gradient
            gx = dx if x_layout == 1 else dx[:, :x_cols]
        return (gx, None, None, None, None) + tuple(grads)


def fused_mlp(x, weights, activation="relu", x_layout=0, batch=None):
    """x [B, in<=32] fp32 CUDA (or [16, pad32(batch), 2] with x_layout=1); weights = list of [out, in] matrices
    (hidden width 64, last out <= 32).  Returns [B, out]."""
    train = torch.is_grad_enabled() and (x.requires_grad or any(w.requires_grad for w in weights))
    return _FusedMLP32.apply(x, 0 if activation == "relu" else 6, x_layout, batch, train, *weights)
