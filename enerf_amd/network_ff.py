"""`NeRFNetwork` on fully-fused MLPs -- re-statement of nerf/network_ff.py of the reference.

sigma_net = FFMLP(32 -> 64 x num_layers -> 16 [1 sigma + 15 geo]); color_net = FFMLP(32 [16 SH + 15 geo + 1 zero pad]
-> 64 x num_layers_color -> 16 [3 rgb]).  The reference forwards **kwargs to NeRFRenderer.__init__ and therefore
cannot be constructed from main_nerf.py (it passes out_dim_color / disable_view_direction); this class accepts and
stores out_dim_color (3) so `render(staged=True)` also works.
"""
import torch

from . import fused_network, fused_network_ff
from .activation import trunc_exp
from .encoding import get_encoder
from .ffmlp import FFMLP
from .renderer import NeRFRenderer


class NeRFNetwork(NeRFRenderer):
    def __init__(self, encoding="hashgrid", encoding_dir="sphere_harmonics", num_layers=2, hidden_dim=64,
                 geo_feat_dim=15, num_layers_color=3, hidden_dim_color=64, bound=1, out_dim_color=3,
                 disable_view_direction=False, **kwargs):
        super().__init__(bound, **kwargs)
        assert out_dim_color == 3, "the fully-fused colour net has 3 outputs (network_ff.py:44-49)"
        self.out_dim_color = 3
        self.disable_view_direction = disable_view_direction
        self.num_layers = num_layers
        self.hidden_dim = hidden_dim
        self.geo_feat_dim = geo_feat_dim
        self.encoder, self.in_dim = get_encoder(encoding, desired_resolution=2048 * bound)
        self.sigma_net = FFMLP(input_dim=self.in_dim, output_dim=1 + self.geo_feat_dim, hidden_dim=self.hidden_dim,
                               num_layers=self.num_layers)

        self.num_layers_color = num_layers_color
        self.hidden_dim_color = hidden_dim_color
        self.encoder_dir, self.in_dim_color = get_encoder(encoding_dir)
        self.in_dim_color += self.geo_feat_dim + 1   # pad 31 -> 32 (network_ff.py:42)
        self.color_net = FFMLP(input_dim=self.in_dim_color, output_dim=3, hidden_dim=self.hidden_dim_color,
                               num_layers=self.num_layers_color)

    def _color_in(self, d, geo_feat):
        d = self.encoder_dir(d)
        p = torch.zeros_like(geo_feat[..., :1])
        return torch.cat([d.to(geo_feat.dtype), geo_feat, p], dim=-1)

    def forward(self, x, d):
        if fused_network_ff.supported(self, x, d):
            return fused_network_ff.forward(self, x, d)           # inference: grid encode + one MFMA kernel
        if fused_network.supported(self, x, d):
            # training: one autograd node over the bf16-operand MFMA kernels (csrc/mlp32s.hip, precision 2), fp32 master
            # weights, the activations of the five hidden layers kept for a fused dgrad + wgrad backward per net
            return fused_network.forward(self, x, d)
        h = self.sigma_net(self.encoder(x, bound=self.bound))
        sigma = trunc_exp(h[..., 0])
        geo_feat = h[..., 1:]
        h = self.color_net(self._color_in(d, geo_feat))
        return sigma, torch.sigmoid(h)

    def forward_into(self, x, d, sigma_out, rgb_out):
        """forward() into caller-owned fp32 buffers (the whole-frame renderer's sample arrays)."""
        if fused_network_ff.supported(self, x, d):
            fused_network_ff.forward(self, x, d, out=(sigma_out, rgb_out))
        else:
            sigma, rgb = self(x, d)
            sigma_out.copy_(sigma)
            rgb_out.copy_(rgb)

    def density(self, x):
        h = self.sigma_net(self.encoder(x, bound=self.bound))
        return {"sigma": trunc_exp(h[..., 0]), "geo_feat": h[..., 1:]}

    def color(self, x, d, mask=None, geo_feat=None, **kwargs):
        if mask is not None:
            rgbs = torch.zeros(mask.shape[0], 3, dtype=x.dtype, device=x.device)
            if not mask.any():
                return rgbs
            x, d, geo_feat = x[mask], d[mask], geo_feat[mask]
        h = torch.sigmoid(self.color_net(self._color_in(d, geo_feat)))
        if mask is not None:
            rgbs[mask] = h.to(rgbs.dtype)
            return rgbs
        return h

    def get_params(self, lr):
        return [
            {"params": self.encoder.parameters(), "lr": lr},
            {"params": self.sigma_net.parameters(), "lr": lr},
            {"params": self.encoder_dir.parameters(), "lr": lr},
            {"params": self.color_net.parameters(), "lr": lr},
        ]
