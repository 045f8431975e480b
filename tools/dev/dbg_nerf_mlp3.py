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
s0, c0, sv = fn.nerf_forward(x, d, cfg, True, params[0], offs, *params[1:])
h32 = sv["h32"]
lib.enerf_debug_nerf_mlp_fused(1)
for it in range(3):
    s1 = torch.full((N,), float("nan"), device=DEV); c1 = torch.full((N, 3), float("nan"), device=DEV)
    fn.nerf_forward(x, d, cfg, False, params[0], offs, *params[1:], out=(s1, c1))
    torch.cuda.synchronize()
    ref = torch.stack([h32[:, 1], h32[:, 17], h32[:, 18]], dim=1)
    for k, name in enumerate(("geo1", "sh1", "sh2")):
        dk = (c1[:, k] - ref[:, k]).abs()
        bad = (dk > 1e-6).nonzero().flatten()
        print(it, name, "bad", bad.numel(), "max", float(dk.max()), "lanes", sorted(set((bad % 32).tolist()))[:40], "tiles", sorted(set((bad // 32).tolist()))[:6])
