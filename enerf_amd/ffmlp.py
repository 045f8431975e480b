"""Host-side mirror of ffmlp/ffmlp.py (reference): `ffmlp_forward` autograd function and the `FFMLP` module
(single flat parameter `weights`, same layout, same `manual_seed(42)` U(+-sqrt(3/hidden)) init, same 128-row padding).

Storage dtype: the reference runs fp16 (custom_fwd(cast_inputs=torch.half)) with fp16 accumulation; here the module
computes in `FFMLP.compute_dtype` (bf16 by default -- MFMA 64-wide bf16 tiles with fp32 accumulation; fp16 is also
accepted by the kernels for bit-layout compatibility with the reference's wrapper).
"""
import math

import torch
import torch.nn as nn
from torch.autograd import Function

from .backends import _ffmlp as _backend

_ACTIVATIONS = ("relu", "exponential", "sine", "sigmoid", "squareplus", "softplus")      # ffmlp.h's enum, then None = 6
_ROW_GROUP = 128                     # the kernels' batch granularity (ffmlp.py:157)


class _ffmlp_forward(Function):
    """(inputs [B,in], flat weights) -> outputs [B,out_padded].  Training keeps the post-activation buffer of every
    hidden layer for the backward; inference uses one scratch row block (ffmlp.py:24-83 of the reference)."""

    @staticmethod
    def forward(ctx, inputs, weights, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                inference=False, calc_grad_inputs=False, compute_dtype=torch.bfloat16):
        rows, dev = inputs.shape[0], inputs.device
        shape = (input_dim, output_dim, hidden_dim, num_layers, activation, output_activation)
        x = inputs.to(compute_dtype).contiguous()
        w = weights.to(compute_dtype).contiguous()
        y = torch.empty(rows, output_dim, device=dev, dtype=compute_dtype)
        if inference:
            scratch = torch.empty(rows, hidden_dim, device=dev, dtype=compute_dtype)
            _backend.ffmlp_inference(x, w, rows, *shape, scratch, y)
            return y
        hidden = torch.empty(num_layers, rows, hidden_dim, device=dev, dtype=compute_dtype)
        _backend.ffmlp_forward(x, w, rows, *shape, hidden, y)
        ctx.save_for_backward(x, w, hidden)
        ctx.shape, ctx.want_dx, ctx.dtypes = shape, bool(calc_grad_inputs), (inputs.dtype, weights.dtype)
        return y

    @staticmethod
    def backward(ctx, grad):
        x, w, hidden = ctx.saved_tensors
        num_layers, hidden_dim = ctx.shape[3], ctx.shape[2]
        rows = grad.shape[0]
        dy = grad.to(x.dtype).contiguous()
        # the kernels accumulate into all three: zero-filled, as the reference allocates them
        dx = torch.zeros_like(x) if ctx.want_dx else torch.zeros(1, device=dy.device, dtype=dy.dtype)
        dw = torch.zeros_like(w)
        scratch = torch.zeros(num_layers, rows, hidden_dim, device=dy.device, dtype=dy.dtype)
        _backend.ffmlp_backward(dy, x, w, hidden, rows, *ctx.shape, ctx.want_dx, scratch, dx, dw)
        in_dtype, w_dtype = ctx.dtypes
        return (dx.to(in_dtype) if ctx.want_dx else None, dw.to(w_dtype)) + (None,) * 9


ffmlp_forward = _ffmlp_forward.apply


def convert_activation(act):
    return _ACTIVATIONS.index(act) if act in _ACTIVATIONS else 6


def _check(ok, message):
    if not ok:
        raise AssertionError(message)


class FFMLP(nn.Module):
    compute_dtype = torch.bfloat16

    def __init__(self, input_dim, output_dim, hidden_dim, num_layers, activation="relu"):
        super().__init__()
        # the reference's limits and messages (ffmlp.py:110-113)
        _check(hidden_dim in (16, 32, 64, 128, 256),
               f"FFMLP only support hidden_dim in [16, 32, 64, 128, 256], but got {hidden_dim}")
        _check(input_dim > 0 and input_dim % 16 == 0, f"FFMLP input_dim should be 16 * m (m  > 0), but got {input_dim}")
        _check(output_dim <= 16, f"FFMLP current only supports output dim <= 16, but got {output_dim}")
        _check(num_layers >= 2, f"FFMLP num_layers should be larger than 2 (3 matmuls), but got {num_layers}")
        self.input_dim, self.output_dim, self.hidden_dim, self.num_layers = input_dim, output_dim, hidden_dim, num_layers
        self.activation, self.output_activation = convert_activation(activation), convert_activation("none")
        self.tensorcore_width = 16
        self.padded_output_dim = 16 * -(-output_dim // 16)
        # one flat vector: [hidden x in | (num_layers - 1) x hidden x hidden | padded_out x hidden], W[out][in] row-major
        self.num_parameters = hidden_dim * (input_dim + (num_layers - 1) * hidden_dim + self.padded_output_dim)
        self.weights = nn.Parameter(torch.zeros(self.num_parameters))
        self.reset_parameters()
        _backend.allocate_splitk(num_layers + 1)

    def cleanup(self):
        _backend.free_splitk()

    def __repr__(self):
        return (f"FFMLP: input_dim={self.input_dim} output_dim={self.output_dim} hidden_dim={self.hidden_dim} "
                f"num_layers={self.num_layers} activation={self.activation}")

    def reset_parameters(self):
        torch.manual_seed(42)   # the reference reseeds the global RNG here (ffmlp.py:142); kept for identical init
        bound = math.sqrt(3 / self.hidden_dim)
        self.weights.data.uniform_(-bound, bound)

    def forward(self, inputs):
        """inputs [B, input_dim] -> [B, output_dim]; B is padded by 128 - B % 128 zero rows (always >= 1 row group,
        ffmlp.py:157-159)."""
        rows, width = inputs.shape
        fill = _ROW_GROUP - rows % _ROW_GROUP
        padded = torch.cat([inputs, inputs.new_zeros(fill, width)], dim=0)
        y = ffmlp_forward(padded, self.weights, self.input_dim, self.padded_output_dim, self.hidden_dim,
                          self.num_layers, self.activation, self.output_activation, not self.training,
                          inputs.requires_grad, self.compute_dtype)
        return y[:rows, :self.output_dim]
