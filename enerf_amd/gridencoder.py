"""Host-side mirror of gridencoder/grid.py (reference): `grid_encode` autograd function and the `GridEncoder`
module (same constructor arguments, attribute / parameter / buffer names and init, so state_dicts interchange).

MI355X-side difference, invisible to callers: when the backend advertises `layout=` support (the HIP backend does),
the kernel writes the [B, L*C] result directly and reads the [B, L*C] gradient directly, so the two full
permute+copy passes of grid.py:52 and grid.py:70 disappear.  `GridEncoder.forward_level_major` additionally exposes
the kernel-native [L, Bp, C] tensor for consumers that can read it in place (enerf_amd.fused_mlp / fused_network).
"""
import inspect

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from .backends import _gridencoder as _backend

_gridtype_to_id = {"hash": 0, "tiled": 1}

# The backward kernels *add* into the gradient buffer they are given.  When the embedding table is a leaf parameter
# whose .grad already exists (zeroed by the optimizer), they can add straight into it and hand autograd None for that
# input: this removes a 52 MB zero-fill and a 156 MB grad += pass from every step.  It also bypasses AccumulateGrad, so
# DistributedDataParallel's reducer hooks and register_post_accumulate_grad_hook callbacks would never fire, and
# torch.autograd.grad(...) would mutate .grad: OPT-IN for autograd-driven loops (set True only by a loop that owns the
# step and uses neither).  Loops that drive the backward themselves, without autograd (fused_render.backward_raw), own
# the gradient buffer and always use it (param_grad_target(owner=True)).
ACCUMULATE_INTO_PARAM_GRAD = False


def param_grad_target(param, dtype, owner=False):
    """-> param.grad if the backward may add straight into it, else None (see ACCUMULATE_INTO_PARAM_GRAD)."""
    g = param.grad
    if (g is None or not param.is_leaf or g.dtype != dtype or not g.is_contiguous() or g.shape != param.shape):
        return None
    if owner:
        return g
    if not ACCUMULATE_INTO_PARAM_GRAD or param._backward_hooks or getattr(param, "_post_accumulate_grad_hooks", None):
        return None
    return g


_layout_support = {}


def _supports_layout():
    """Does the active backend's grid_encode_forward take `layout=`?  (cached per backend function: tests swap the
    module-level `_backend` for the CPU oracle's)"""
    fn = _backend.grid_encode_forward
    r = _layout_support.get(fn)
    if r is None:
        try:
            r = "layout" in inspect.signature(fn).parameters
        except (TypeError, ValueError):
            r = False
        _layout_support[fn] = r
    return r


class _grid_encode(Function):
    @staticmethod
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False,
                gridtype=0, level_major=False):
        """inputs [B,D] fp32 in [0,1]; embeddings [rows,C]; offsets [L+1] int32 -> [B, L*C]   (grid.py:19-58)
        level_major=True (HIP backend only) returns the kernel-native [L, Bp, C] tensor instead, Bp = B rounded up
        to 32 with zero pad rows -- the input format of fused_mlp(..., x_layout=1)."""
        inputs = inputs.float().contiguous()
        B, D = inputs.shape
        L = offsets.shape[0] - 1
        C = embeddings.shape[1]
        S = np.log2(per_level_scale)   # python float; the kernel side takes it as fp32 (grid.py:32)
        H = base_resolution

        # half-precision tables only under autocast and only for even C (grid.py:36-39)
        if torch.is_autocast_enabled() and C % 2 == 0:
            embeddings = embeddings.to(torch.half)
        embeddings = embeddings.contiguous()

        direct = _supports_layout()
        if level_major:
            if not direct:
                raise RuntimeError("level_major output needs a grid backend with layout support")
            outputs = torch.empty(L, (B + 31) // 32 * 32, C, device=inputs.device, dtype=embeddings.dtype)
        elif direct:
            outputs = torch.empty(B, L * C, device=inputs.device, dtype=embeddings.dtype)
        else:
            outputs = torch.empty(L, B, C, device=inputs.device, dtype=embeddings.dtype)
        if calc_grad_inputs:
            dy_dx = torch.empty(B, L * D * C, device=inputs.device, dtype=embeddings.dtype)
        else:
            dy_dx = torch.empty(1, device=inputs.device, dtype=embeddings.dtype)

        if direct:
            _backend.grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, calc_grad_inputs,
                                         dy_dx, gridtype, layout=2 if level_major else 1)
        else:
            _backend.grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, calc_grad_inputs,
                                         dy_dx, gridtype)
            outputs = outputs.permute(1, 0, 2).reshape(B, L * C)

        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.dims = (B, D, C, L, S, H, gridtype)
        ctx.calc_grad_inputs = calc_grad_inputs
        ctx.direct = direct
        ctx.level_major = level_major
        return outputs

    @staticmethod
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H, gridtype = ctx.dims
        calc_grad_inputs = ctx.calc_grad_inputs

        grad = grad.to(embeddings.dtype)
        target = param_grad_target(embeddings, embeddings.dtype) if ctx.direct else None
        direct_param = target is not None
        grad_embeddings = target if direct_param else torch.zeros_like(embeddings)
        if calc_grad_inputs:
            grad_inputs = torch.zeros_like(inputs, dtype=embeddings.dtype)
        else:
            grad_inputs = torch.zeros(1, device=inputs.device, dtype=embeddings.dtype)

        if ctx.direct:
            grad = grad.contiguous()
            _backend.grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H,
                                          calc_grad_inputs, dy_dx, grad_inputs, gridtype,
                                          layout=2 if ctx.level_major else 1)
        else:
            grad = grad.view(B, L, C).permute(1, 0, 2).contiguous()
            _backend.grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H,
                                          calc_grad_inputs, dy_dx, grad_inputs, gridtype)

        if direct_param:
            grad_embeddings = None
        if calc_grad_inputs:
            return grad_inputs.to(inputs.dtype), grad_embeddings, None, None, None, None, None, None
        return None, grad_embeddings, None, None, None, None, None, None


grid_encode = _grid_encode.apply


def level_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size):
    """Row offset of every level (grid.py:113-123): min(2^log2_hashmap_size, (ceil(base*s^i)+1)^D) rounded up to 8."""
    offsets, offset = [], 0
    max_params = 2 ** log2_hashmap_size
    for i in range(num_levels):
        resolution = int(np.ceil(base_resolution * per_level_scale ** i))
        params_in_level = min(max_params, (resolution + 1) ** input_dim)
        params_in_level = int(np.ceil(params_in_level / 8) * 8)
        offsets.append(offset)
        offset += params_in_level
    offsets.append(offset)
    return np.array(offsets, dtype=np.int32)


class GridEncoder(nn.Module):
    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None, gridtype="hash"):
        super().__init__()
        if desired_resolution is not None:   # overrides per_level_scale (grid.py:96-97)
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))

        self.input_dim = input_dim
        self.num_levels = num_levels
        self.level_dim = level_dim
        self.per_level_scale = per_level_scale
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype = gridtype
        self.gridtype_id = _gridtype_to_id[gridtype]
        self.max_params = 2 ** log2_hashmap_size

        if level_dim % 2 != 0:
            print("[WARN] detected HashGrid level_dim % 2 != 0, which will cause very slow backward is also enabled fp16! (maybe fix later)")

        offsets = level_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size)
        self.register_buffer("offsets", torch.from_numpy(offsets))
        self.n_params = int(offsets[-1]) * level_dim
        self.embeddings = nn.Parameter(torch.empty(int(offsets[-1]), level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        std = 1e-4
        self.embeddings.data.uniform_(-std, std)

    def __repr__(self):
        return (f"GridEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"resolution={self.base_resolution} -> "
                f"{int(round(self.base_resolution * self.per_level_scale ** (self.num_levels - 1)))} "
                f"per_level_scale={self.per_level_scale:.4f} params={tuple(self.embeddings.shape)} "
                f"gridtype={self.gridtype}")

    def forward(self, inputs, bound=1):
        """inputs [..., input_dim] in [-bound, bound] -> [..., num_levels*level_dim]"""
        inputs = (inputs + bound) / (2 * bound)
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.input_dim)
        outputs = grid_encode(inputs, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution,
                              inputs.requires_grad, self.gridtype_id)
        return outputs.view(prefix_shape + [self.output_dim])

    def forward_level_major(self, inputs, bound=1):
        """inputs [B, input_dim] in [-bound, bound] -> ([num_levels, Bp, level_dim], B): the same features as
        forward(), in the kernel's own level-major order with the batch padded to a multiple of 32 (zero rows).
        Feeds enerf_amd.fused_mlp.fused_mlp(..., x_layout=1, batch=B) without a transpose in either direction."""
        inputs = ((inputs + bound) / (2 * bound)).view(-1, self.input_dim)
        outputs = grid_encode(inputs, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution,
                              inputs.requires_grad, self.gridtype_id, True)
        return outputs, inputs.shape[0]
