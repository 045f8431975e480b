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


class _ffmlp_forward(Function):
    @staticmethod
    def forward(ctx, inputs, weights, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                inference=False, calc_grad_inputs=False, compute_dtype=torch.bfloat16):
        B = inputs.shape[0]
        ctx.in_dtype = inputs.dtype
        ctx.w_dtype = weights.dtype
        inputs = inputs.to(compute_dtype).contiguous()
        weights = weights.to(compute_dtype).contiguous()
        outputs = torch.empty(B, output_dim, device=inputs.device, dtype=compute_dtype)
        if not inference:
            forward_buffer = torch.empty(num_layers, B, hidden_dim, device=inputs.device, dtype=compute_dtype)
            _backend.ffmlp_forward(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation,
                                   output_activation, forward_buffer, outputs)
            ctx.save_for_backward(inputs, weights, outputs, forward_buffer)
            ctx.dims = (input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                        calc_grad_inputs)
        else:
            inference_buffer = torch.empty(B, hidden_dim, device=inputs.device, dtype=compute_dtype)
            _backend.ffmlp_inference(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation,
                                     output_activation, inference_buffer, outputs)
        return outputs

    @staticmethod
    def backward(ctx, grad):
        inputs, weights, outputs, forward_buffer = ctx.saved_tensors
        input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, calc_grad_inputs = ctx.dims
        B = grad.shape[0]
        grad = grad.to(inputs.dtype).contiguous()
        if calc_grad_inputs:
            grad_inputs = torch.zeros_like(inputs)
        else:
            grad_inputs = torch.zeros(1, device=grad.device, dtype=grad.dtype)
        grad_weights = torch.zeros_like(weights)
        backward_buffer = torch.zeros(num_layers, B, hidden_dim, device=grad.device, dtype=grad.dtype)
        _backend.ffmlp_backward(grad, inputs, weights, forward_buffer, B, input_dim, output_dim, hidden_dim,
                                num_layers, activation, output_activation, calc_grad_inputs, backward_buffer,
                                grad_inputs, grad_weights)
        gw = grad_weights.to(ctx.w_dtype)
        gi = grad_inputs.to(ctx.in_dtype) if calc_grad_inputs else None
        return gi, gw, None, None, None, None, None, None, None, None, None


ffmlp_forward = _ffmlp_forward.apply


def convert_activation(act):
    return {"relu": 0, "exponential": 1, "sine": 2, "sigmoid": 3, "squareplus": 4, "softplus": 5}.get(act, 6)


class FFMLP(nn.Module):
    compute_dtype = torch.bfloat16

    def __init__(self, input_dim, output_dim, hidden_dim, num_layers, activation="relu"):
        super().__init__()
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.hidden_dim = hidden_dim
        self.num_layers = num_layers
        self.activation = convert_activation(activation)
        self.output_activation = convert_activation("none")
        self.tensorcore_width = 16

        assert hidden_dim in [16, 32, 64, 128, 256], f"FFMLP only support hidden_dim in [16, 32, 64, 128, 256], but got {hidden_dim}"
        assert input_dim > 0 and input_dim % 16 == 0, f"FFMLP input_dim should be 16 * m (m  > 0), but got {input_dim}"
        assert output_dim <= 16, f"FFMLP current only supports output dim <= 16, but got {output_dim}"
        assert num_layers >= 2, f"FFMLP num_layers should be larger than 2 (3 matmuls), but got {num_layers}"

        self.padded_output_dim = int(math.ceil(output_dim / 16)) * 16
        self.num_parameters = hidden_dim * (input_dim + hidden_dim * (num_layers - 1) + self.padded_output_dim)
        self.weights = nn.Parameter(torch.zeros(self.num_parameters))
        self.reset_parameters()
        _backend.allocate_splitk(self.num_layers + 1)

    def cleanup(self):
        _backend.free_splitk()

    def __repr__(self):
        return (f"FFMLP: input_dim={self.input_dim} output_dim={self.output_dim} hidden_dim={self.hidden_dim} "
                f"num_layers={self.num_layers} activation={self.activation}")

    def reset_parameters(self):
        torch.manual_seed(42)   # the reference reseeds the global RNG here (ffmlp.py:142); kept for identical init
        std = math.sqrt(3 / self.hidden_dim)
        self.weights.data.uniform_(-std, std)

    def forward(self, inputs):
        """inputs [B, input_dim] -> [B, output_dim]; B is padded by 128 - B % 128 zero rows (always >= 1 row group,
        ffmlp.py:157-159)."""
        B, C = inputs.shape
        pad = 128 - (B % 128)
        if pad > 0:
            inputs = torch.cat([inputs, torch.zeros(pad, C, dtype=inputs.dtype, device=inputs.device)], dim=0)
        outputs = ffmlp_forward(inputs, self.weights, self.input_dim, self.padded_output_dim, self.hidden_dim,
                                self.num_layers, self.activation, self.output_activation, not self.training,
                                inputs.requires_grad, self.compute_dtype)
        if B != outputs.shape[0] or self.padded_output_dim != self.output_dim:
            outputs = outputs[:B, :self.output_dim]
        return outputs
