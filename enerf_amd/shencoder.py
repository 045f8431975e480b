"""Spherical-harmonics direction encoding above the C ABI: `sh_encode` / `SHEncoder`, the interface of the reference's
shencoder/sphere_harmonics.py (same names, arguments, defaults and output layout) so that encoding.py and the networks
use it unchanged.  Kernels: csrc/shencoder.hip through backends/_shencoder.py."""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .backends import _shencoder as _backend

MAX_DEGREE = 8


class _sh_encoder(Function):
    """y[b, l*l + l + m] = Y_l^m(x[b]) for l < degree, x taken as given (not normalised).  With `calc_grad_inputs` the
    forward also stores the Jacobian [B, 3 * degree^2] and the backward contracts it with the incoming gradient."""

    @staticmethod
    def forward(ctx, inputs, degree, calc_grad_inputs=False):
        x = inputs.to(torch.half) if torch.is_autocast_enabled() else inputs      # reference: cast_inputs=torch.half
        x = x.contiguous()
        rows, dim = x.shape
        new = lambda *shape: torch.empty(*shape, dtype=x.dtype, device=x.device)  # noqa: E731
        y = new(rows, degree * degree)
        jac = new(rows, dim * degree * degree) if calc_grad_inputs else new(1)
        _backend.sh_encode_forward(x, y, rows, dim, degree, calc_grad_inputs, jac)
        if calc_grad_inputs:
            ctx.save_for_backward(x, jac)
        ctx.degree = degree if calc_grad_inputs else 0
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, grad):
        if ctx.degree == 0:
            return None, None, None
        x, jac = ctx.saved_tensors
        dx = torch.zeros_like(x)                          # the kernel accumulates into it
        _backend.sh_encode_backward(grad.to(x.dtype).contiguous(), x, x.shape[0], x.shape[1], ctx.degree, jac, dx)
        return dx, None, None


sh_encode = _sh_encoder.apply


class SHEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        if input_dim != 3:
            raise AssertionError("SH encoder only support input dim == 3")
        if not 1 <= degree <= MAX_DEGREE:
            raise AssertionError("SH encoder only supports degree in [1, 8]")
        self.input_dim, self.degree, self.output_dim = input_dim, degree, degree * degree

    def __repr__(self):
        return f"SHEncoder: input_dim={self.input_dim} degree={self.degree}"

    def forward(self, inputs, size=1):
        """inputs [..., 3] in [-size, size] -> [..., degree^2]; the only preprocessing is the division by `size`."""
        flat = (inputs / size).reshape(-1, self.input_dim)
        y = sh_encode(flat, self.degree, flat.requires_grad)
        return y.reshape(*inputs.shape[:-1], self.output_dim)
