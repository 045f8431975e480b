"""One autograd node for `NeRFNetwork.forward` on the MI355X fp32 path (nerf/network.py:104-132 of the reference):

    x -> (x + bound) / (2 bound) -> hash grid -> sigma MLP -> [trunc_exp | geo_feat]
    d -> SH -------------------------------------+-> colour MLP -> sigmoid

The kernels are the library's own entry points (grid_encode_forward/backward, sh_encode_forward, enerf_mlp32_*); what
this node removes is the glue between them -- the normalisation kernels, the permute of the encoding, trunc_exp and
sigmoid as separate elementwise passes (forward and backward), the [SH | geo_feat] concatenation, the slicing /
re-packing of gradients -- about two dozen small launches and as many torch dispatches per call, which is what bounds
a 4096-ray step once the big kernels are fast.  Data flow:

  * the grid kernel normalises its input itself (affine = (bound, 1/(2 bound)), rounded like torch's two kernels) and
    writes the level-major [16, Bp, 2] tensor the sigma MLP reads;
  * the sigma MLP writes its 16 outputs into columns 0..15 of a [B, 32] buffer and exp(column 0) into `sigma`;
    the SH kernel writes its 16 outputs into columns 16..31 of the same buffer, which is the colour MLP's input
    (its first-layer weight columns are permuted to match: column 0 -- the raw density -- gets zero weight);
  * the colour MLP applies the sigmoid; its backward takes d(rgb) and the saved rgb, the sigma MLP's backward takes
    columns 0..15 of the colour MLP's input gradient in place (row stride 32) with column 0 replaced by
    d(sigma) * exp(clamp(h0, -15, 15)), and hands its input gradient to the grid backward in level-major order.

Same parameters (nn.Linear weights, `encoder.embeddings`), same values to fp32 round-off (the colour net's first layer
sums its 31 products in a different order).  Anything it does not cover (CPU tensors, autocast, other widths, inputs
that need gradients, disable_view_direction) takes the unfused route in network.py.
"""
import numpy as np
import torch
from torch.autograd import Function

from . import _lib as L
from . import gridencoder as _ge
from .backends import _gridencoder as _gb
from .fused_mlp import pad32

ENABLED = True


def supported(net, x, d):
    from .gridencoder import GridEncoder
    from .shencoder import SHEncoder
    enc, encd = net.encoder, net.encoder_dir
    if not (ENABLED and x.is_cuda and x.dtype == torch.float32 and d.dtype == torch.float32 and x.dim() == 2
            and d.dim() == 2 and not torch.is_autocast_enabled() and not net.disable_view_direction):
        return False
    if x.requires_grad or d.requires_grad:
        return False
    if not (isinstance(enc, GridEncoder) and enc.num_levels == 16 and enc.level_dim == 2 and enc.input_dim == 3
            and enc.embeddings.dtype == torch.float32 and _ge._supports_layout()):
        return False
    if not (isinstance(encd, SHEncoder) and encd.degree == 4):
        return False
    s, c = net.sigma_net, net.color_net
    return (len(s) == 2 and tuple(s[0].weight.shape) == (64, 32) and tuple(s[1].weight.shape) == (16, 64)
            and len(c) == 3 and tuple(c[0].weight.shape) == (64, 31) and tuple(c[1].weight.shape) == (64, 64)
            and c[2].weight.shape[1] == 64 and c[2].weight.shape[0] <= 32)


class _FusedNeRF(Function):
    @staticmethod
    def forward(ctx, x, d, cfg, train, embeddings, offsets, ws0, ws1, wc0, wc1, wc2):
        bound, per_level_scale, base_resolution, gridtype = cfg
        x = x.contiguous()
        d = d.contiguous()
        B = x.shape[0]
        Bp = pad32(B)
        dev = x.device
        lib = L.lib()
        out_c = wc2.shape[0]
        sigma = torch.empty(B, dtype=torch.float32, device=dev)
        rgb = torch.empty(B, out_c, dtype=torch.float32, device=dev)
        if B == 0:
            return sigma, rgb
        S = float(np.log2(per_level_scale))
        affine = (float(bound), float(np.float32(1.0) / np.float32(2 * bound)))   # torch: (x + b) * (1.0f / (2b))
        emb = embeddings.contiguous()
        feats = torch.empty(16, Bp, 2, dtype=torch.float32, device=dev)
        dummy = feats                                   # dy_dx is not computed (calc_grad_inputs = False)
        _gb.grid_encode_forward(x, emb, offsets, feats, B, 3, 2, 16, S, base_resolution, False, dummy, gridtype,
                                layout=2, affine=affine)
        stream = L.stream_handle()
        h32 = torch.empty(B, 32, dtype=torch.float32, device=dev)
        blob_s = torch.cat([ws0.reshape(-1), ws1.reshape(-1)])
        fb_s = torch.empty(1, Bp, 64, dtype=torch.float32, device=dev) if train else None
        L.check(lib.enerf_mlp32_forward(feats.data_ptr(), blob_s.data_ptr(), B, 32, 16, 1, 0, 6,
                                        fb_s.data_ptr() if train else None, h32.data_ptr(), 1, 32, sigma.data_ptr(),
                                        stream), "mlp32_forward(sigma)")
        L.check(lib.enerf_sh_encode_forward_strided(d.data_ptr(), h32.data_ptr() + 64, B, 4, 32, stream),
                "sh_encode_forward_strided")
        # colour net input columns: [raw density (zero weight) | geo_feat 15 | SH 16]
        blob_c = torch.cat([wc0.new_zeros(64, 1), wc0[:, 16:], wc0[:, :16]], dim=1).reshape(-1)
        blob_c = torch.cat([blob_c, wc1.reshape(-1), wc2.reshape(-1)])
        fb_c = torch.empty(2, Bp, 64, dtype=torch.float32, device=dev) if train else None
        L.check(lib.enerf_mlp32_forward(h32.data_ptr(), blob_c.data_ptr(), B, 32, out_c, 2, 0, 3,
                                        fb_c.data_ptr() if train else None, rgb.data_ptr(), 0, 0, None, stream),
                "mlp32_forward(color)")
        if train:
            ctx.save_for_backward(x, emb, offsets, feats, h32, fb_s, fb_c, blob_s, blob_c, rgb)
            ctx.cfg = (B, S, base_resolution, gridtype, affine, out_c)
            ctx.embeddings_param = embeddings
        return sigma, rgb

    @staticmethod
    def backward(ctx, g_sigma, g_rgb):
        x, emb, offsets, feats, h32, fb_s, fb_c, blob_s, blob_c, rgb = ctx.saved_tensors
        B, S, H, gridtype, affine, out_c = ctx.cfg
        Bp = pad32(B)
        dev = x.device
        lib = L.lib()
        stream = L.stream_handle()
        g_sigma = torch.zeros(B, dtype=torch.float32, device=dev) if g_sigma is None else g_sigma.float().contiguous()
        g_rgb = torch.zeros(B, out_c, dtype=torch.float32, device=dev) if g_rgb is None else g_rgb.float().contiguous()

        bb_c = torch.empty(2, Bp, 64, dtype=torch.float32, device=dev)
        dx32 = torch.empty(B, 32, dtype=torch.float32, device=dev)
        dw_c = torch.zeros_like(blob_c)
        L.check(lib.enerf_mlp32_backward(g_rgb.data_ptr(), h32.data_ptr(), blob_c.data_ptr(), fb_c.data_ptr(), B, 32,
                                         out_c, 2, 0, bb_c.data_ptr(), dx32.data_ptr(), dw_c.data_ptr(), 0, 0,
                                         rgb.data_ptr(), out_c, None, None, 0, stream), "mlp32_backward(color)")
        bb_s = torch.empty(1, Bp, 64, dtype=torch.float32, device=dev)
        dfeat = torch.empty(16, Bp, 2, dtype=torch.float32, device=dev)
        dw_s = torch.zeros_like(blob_s)
        L.check(lib.enerf_mlp32_backward(dx32.data_ptr(), feats.data_ptr(), blob_s.data_ptr(), fb_s.data_ptr(), B, 32,
                                         16, 1, 0, bb_s.data_ptr(), dfeat.data_ptr(), dw_s.data_ptr(), 1, 32, None, 0,
                                         g_sigma.data_ptr(), h32.data_ptr(), 32, stream), "mlp32_backward(sigma)")

        param = ctx.embeddings_param
        direct = (_ge.ACCUMULATE_INTO_PARAM_GRAD and param.is_leaf and param.grad is not None
                  and param.grad.dtype == torch.float32 and param.grad.is_contiguous()
                  and param.grad.shape == param.shape and not param._backward_hooks)
        g_emb = param.grad if direct else torch.zeros_like(emb)
        _gb.grid_encode_backward(dfeat, x, emb, offsets, g_emb, B, 3, 2, 16, S, H, False, dfeat, dfeat, gridtype,
                                 layout=2, affine=affine)

        g_ws0 = dw_s[:2048].view(64, 32)
        g_ws1 = dw_s[2048:].view(16, 64)
        g0 = dw_c[:2048].view(64, 32)
        g_wc0 = torch.cat([g0[:, 16:], g0[:, 1:16]], dim=1)
        g_wc1 = dw_c[2048:2048 + 4096].view(64, 64)
        g_wc2 = dw_c[2048 + 4096:].view(out_c, 64)
        return (None, None, None, None, None if direct else g_emb, None, g_ws0, g_ws1, g_wc0, g_wc1, g_wc2)


def forward(net, x, d):
    """sigma [N], rgb [N, out_dim_color] for x [N,3] in [-bound, bound], d [N,3]."""
    enc = net.encoder
    params = (enc.embeddings, net.sigma_net[0].weight, net.sigma_net[1].weight, net.color_net[0].weight,
              net.color_net[1].weight, net.color_net[2].weight)
    train = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    cfg = (net.bound, enc.per_level_scale, enc.base_resolution, enc.gridtype_id)
    return _FusedNeRF.apply(x, d, cfg, train, enc.embeddings, enc.offsets, *params[1:])
