"""`NeRFNetwork` with nn.Linear MLPs -- re-statement of nerf/network.py of the reference (hashgrid -> sigma MLP ->
trunc_exp; SH(dir) (+) geo_feat -> colour MLP -> sigmoid).  Module / parameter names match the reference so
checkpoints interchange (`encoder.embeddings`, `encoder.offsets`, `sigma_net.N.weight`, `color_net.N.weight`).
Quirk kept: the colour MLP's hidden width is `hidden_dim`, not `hidden_dim_color` (network.py:68,73).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused_mlp, fused_network
from .activation import trunc_exp
from .encoding import get_encoder
from .gridencoder import GridEncoder, _supports_layout
from .renderer import NeRFRenderer


def _mlp(in_dim, hidden, out_dim, n):
    layers = []
    for l in range(n):
        layers.append(nn.Linear(in_dim if l == 0 else hidden, out_dim if l == n - 1 else hidden, bias=False))
    return nn.ModuleList(layers)


def _run_mlp(layers, h):
    """relu(... relu(h W0^T) ...) W_last^T.  CUDA fp32 inputs take the fused matrix-core path (same parameters, same
    math to fp32 round-off); everything else is the plain nn.Linear loop of the reference."""
    weights = [layer.weight for layer in layers]
    if h.dim() == 2 and fused_mlp.supported(h, weights):
        return fused_mlp.fused_mlp(h, weights, "relu")
    for l, layer in enumerate(layers):
        h = layer(h)
        if l != len(layers) - 1:
            h = F.relu(h, inplace=True)
    return h


class NeRFNetwork(NeRFRenderer):
    def __init__(self, encoding="hashgrid", encoding_dir="sphere_harmonics", encoding_bg="hashgrid", num_layers=2,
                 hidden_dim=64, geo_feat_dim=15, num_layers_color=3, hidden_dim_color=64, num_layers_bg=2,
                 hidden_dim_bg=64, bound=1, disable_view_direction=False, out_dim_color=3, **kwargs):
        super().__init__(bound, **kwargs)
        self.disable_view_direction = disable_view_direction
        self.out_dim_color = out_dim_color

        self.num_layers = num_layers
        self.hidden_dim = hidden_dim
        self.geo_feat_dim = geo_feat_dim
        self.encoder, self.in_dim = get_encoder(encoding, desired_resolution=2048 * bound)
        self.sigma_net = _mlp(self.in_dim, hidden_dim, 1 + geo_feat_dim, num_layers)

        self.num_layers_color = num_layers_color
        self.hidden_dim_color = hidden_dim_color
        self.encoder_dir, self.in_dim_dir = get_encoder(encoding_dir)
        self.color_net = _mlp(self.in_dim_dir + geo_feat_dim, hidden_dim, out_dim_color, num_layers_color)

        if self.bg_radius > 0:
            self.num_layers_bg = num_layers_bg
            self.hidden_dim_bg = hidden_dim_bg
            self.encoder_bg, self.in_dim_bg = get_encoder(encoding_bg, input_dim=2, num_levels=4,
                                                          log2_hashmap_size=19, desired_resolution=2048)
            self.bg_net = _mlp(self.in_dim_bg + self.in_dim_dir, hidden_dim_bg, out_dim_color, num_layers_bg)
        else:
            self.bg_net = None

    def _sigma_mlp(self, x):
        """sigma_net(encoder(x)).  On the fused CUDA path the 16 x 2 hash-grid features stay in the gather kernel's
        level-major order all the way into the matrix-core MLP (and their gradient all the way back)."""
        enc = self.encoder
        weights = [layer.weight for layer in self.sigma_net]
        if isinstance(enc, GridEncoder) and enc.num_levels == 16 and enc.level_dim == 2 and x.dim() == 2 \
                and fused_mlp.supported_level_major(x.device, enc.embeddings.dtype, weights) and _supports_layout():
            feats, n = enc.forward_level_major(x, bound=self.bound)
            return fused_mlp.fused_mlp(feats, weights, "relu", x_layout=1, batch=n)
        return _run_mlp(self.sigma_net, enc(x, bound=self.bound))

    def _color_input(self, d, geo_feat):
        """[dir encoding | geo_feat] (network.py:123); on the fused CUDA path the 31 columns are written straight into a
        32-wide buffer (one zero column) so the fused MLP needs no extra padding pass."""
        e = self._dir_features(d)
        if e.dim() == 2 and e.shape[1] + geo_feat.shape[1] == 31 and e.dtype == geo_feat.dtype \
                and fused_mlp.supported(e.new_empty((0, 32)), [layer.weight for layer in self.color_net]):
            return torch.cat([e, geo_feat, torch.zeros_like(geo_feat[:, :1])], dim=-1)
        return torch.cat([e, geo_feat], dim=-1)

    def _dir_features(self, d):
        e = self.encoder_dir(d)
        return e * 0 if self.disable_view_direction else e * 1

    def forward(self, x, d):
        """x [N,3] in [-bound,bound], d [N,3] -> sigma [N], color [N,out_dim_color]   (network.py:104-132)"""
        if fused_network.supported(self, x, d):
            return fused_network.forward(self, x, d)
        h = self._sigma_mlp(x)
        sigma = trunc_exp(h[..., 0])
        geo_feat = h[..., 1:]
        h = _run_mlp(self.color_net, self._color_input(d, geo_feat))
        return sigma, torch.sigmoid(h)

    def forward_into(self, x, d, sigma_out, rgb_out):
        """forward() into caller-owned fp32 buffers (the whole-frame renderer's sample arrays)."""
        if fused_network.supported(self, x, d) and not torch.is_grad_enabled() and sigma_out.is_contiguous() \
                and rgb_out.is_contiguous():
            fused_network.forward_into(self, x, d, sigma_out, rgb_out)
        else:
            sigma, rgb = self(x, d)
            sigma_out.copy_(sigma)
            rgb_out.copy_(rgb)

    def density(self, x):
        h = self._sigma_mlp(x)
        return {"sigma": trunc_exp(h[..., 0]), "geo_feat": h[..., 1:]}

    def background(self, x, d):
        h = torch.cat([self._dir_features(d), self.encoder_bg(x)], dim=-1)
        return torch.sigmoid(_run_mlp(self.bg_net, h))

    def color(self, x, d, mask=None, geo_feat=None, **kwargs):
        """Masked colour query (network.py:171-199): rows where mask is False stay zero."""
        if mask is not None:
            rgbs = torch.zeros(mask.shape[0], self.out_dim_color, dtype=x.dtype, device=x.device)
            if not mask.any():
                return rgbs
            x, d, geo_feat = x[mask], d[mask], geo_feat[mask]
        h = torch.sigmoid(_run_mlp(self.color_net, self._color_input(d, geo_feat)))
        if mask is not None:
            rgbs[mask] = h.to(rgbs.dtype)
            return rgbs
        return h

    def get_params(self, lr):
        params = [
            {"params": self.encoder.parameters(), "lr": lr},
            {"params": self.sigma_net.parameters(), "lr": lr},
            {"params": self.encoder_dir.parameters(), "lr": lr},
            {"params": self.color_net.parameters(), "lr": lr},
        ]
        if self.bg_radius > 0:
            params.append({"params": self.encoder_bg.parameters(), "lr": lr})
            params.append({"params": self.bg_net.parameters(), "lr": lr})
        return params
