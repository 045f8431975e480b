import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from enerf_amd import _lib, fused_network as fn
from enerf_amd.network import NeRFNetwork
lib = _lib.lib()
DEV = "cuda"
N = 70001
torch.manual_seed(5)
m = NeRFNetwork(encoding="hashgrid", bound=2, cuda_ray=True, out_dim_color=3).to(DEV)
m.encoder.embeddings.data.uniform_(-1, 1)
x = torch.rand(N, 3, device=DEV) * 4 - 2
d = torch.nn.functional.normalize(torch.randn(N, 3, device=DEV), dim=-1)
params = fn.network_params(m)
cfg, offs = fn.network_cfg(m), fn.encoder_offsets(m)
prev = lib.enerf_debug_nerf_mlp_fused(0)
s0 = torch.empty(N, device=DEV); c0 = torch.empty(N, 3, device=DEV)
fn.nerf_forward(x, d, cfg, False, params[0], offs, *params[1:], out=(s0, c0))
tot = 0
for fused in (1, 0):
    lib.enerf_debug_nerf_mlp_fused(fused)
    nb = 0; mx = 0.0
    for it in range(200):
        s1 = torch.full((N,), float("nan"), device=DEV); c1 = torch.full((N, 3), float("nan"), device=DEV)
        fn.nerf_forward(x, d, cfg, False, params[0], offs, *params[1:], out=(s1, c1))
        dc = (c1 - c0).abs().max(dim=1).values
        nb += int(((dc > 2e-5) | dc.isnan()).sum()); mx = max(mx, float(dc.max()))
    print("fused" if fused else "one launch per net", "200 forwards of 70001 rows: bad rows", nb, "max diff", mx)
