"""Inference of nerf/network_ff.py's NeRFNetwork (hash grid -> sigma FFMLP -> exp ; SH + geo_feat -> colour FFMLP ->
sigmoid) as two launches: the level-major grid encode and csrc/ffnerf.hip.  The op-by-op route (enerf_amd/network_ff.py)
spends 40 % of a frame in conversions, padding cats and slicing around its two FFMLP kernels; this path has none.
Used by NeRFNetwork.forward when nothing needs a gradient; same roundings as the op-by-op route (tests)."""
import numpy as np
import torch

from . import _lib as L
from .backends import _gridencoder as _gb
from .fused_mlp import pad32

ENABLED = True
_DTYPES = {torch.bfloat16: 2, torch.float16: 1}


def _architecture_supported(net):
    from .ffmlp import FFMLP
    from .gridencoder import GridEncoder
    from .shencoder import SHEncoder
    enc, encd = getattr(net, "encoder", None), getattr(net, "encoder_dir", None)
    s, c = getattr(net, "sigma_net", None), getattr(net, "color_net", None)
    if not (isinstance(enc, GridEncoder) and enc.num_levels == 16 and enc.level_dim == 2 and enc.input_dim == 3
            and enc.embeddings.dtype == torch.float32 and isinstance(encd, SHEncoder) and encd.degree == 4):
        return False
    if not (isinstance(s, FFMLP) and isinstance(c, FFMLP)):
        return False
    relu, none = 0, 6
    return (s.input_dim == 32 and s.hidden_dim == 64 and s.num_layers == 2 and s.output_dim == 16
            and c.input_dim == 32 and c.hidden_dim == 64 and c.num_layers == 3 and c.output_dim == 3
            and int(s.activation) == relu and int(c.activation) == relu and int(s.output_activation) == none
            and int(c.output_activation) == none and s.compute_dtype in _DTYPES and c.compute_dtype == s.compute_dtype
            and s.weights.dtype == torch.float32 and c.weights.dtype == torch.float32)


def supported(net, x, d):
    if not (ENABLED and x.is_cuda and x.dtype == torch.float32 and d.dtype == torch.float32 and x.dim() == 2
            and d.dim() == 2 and x.shape[0] == d.shape[0] and not torch.is_autocast_enabled()):
        return False
    if torch.is_grad_enabled() and (net.training or x.requires_grad or d.requires_grad
                                    or any(p.requires_grad for p in net.parameters())):
        return False                                   # someone may want a gradient: the autograd route serves it
    ok = net.__dict__.get("_ffnerf_arch_ok")           # (cached on the net, not by id(net): ids are reused after a delete)
    if ok is None:
        ok = net.__dict__["_ffnerf_arch_ok"] = _architecture_supported(net)
    return ok


@torch.no_grad()
def forward(net, x, d, out=None):
    """sigma [N], rgb [N,3] (fp32 tensors holding 16-bit-rounded values) for x [N,3] in [-bound, bound], d [N,3];
    `out` = (sigma, rgb) contiguous fp32 tensors to write into."""
    x = x.contiguous()
    d = d.contiguous()
    B = x.shape[0]
    dev = x.device
    if out is None:
        sigma = torch.empty(B, dtype=torch.float32, device=dev)
        rgb = torch.empty(B, 3, dtype=torch.float32, device=dev)
    else:
        sigma, rgb = out
        assert sigma.shape == (B,) and rgb.shape == (B, 3) and sigma.is_contiguous() and rgb.is_contiguous() \
            and sigma.dtype == rgb.dtype == torch.float32
    if B == 0:
        return sigma, rgb
    enc = net.encoder
    bound = net.bound
    feats = torch.empty(16, pad32(B), 2, dtype=torch.float32, device=dev)
    affine = (float(bound), float(np.float32(1.0) / np.float32(2 * bound)))   # torch: (x + b) * (1.0f / (2b))
    _gb.grid_encode_forward(x, enc.embeddings.contiguous(), enc.offsets, feats, B, 3, 2, 16,
                            float(np.log2(enc.per_level_scale)), enc.base_resolution, False, feats, enc.gridtype_id,
                            layout=2, affine=affine)
    L.check(L.lib().enerf_ffnerf_inference(feats.data_ptr(), d.data_ptr(), net.sigma_net.weights.data_ptr(),
                                           net.color_net.weights.data_ptr(), B, _DTYPES[net.sigma_net.compute_dtype],
                                           sigma.data_ptr(), rgb.data_ptr(), L.stream_handle()), "ffnerf_inference")
    return sigma, rgb
