import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from enerf_amd import _lib, fused_network as fn
from enerf_amd.network import NeRFNetwork
lib = _lib.lib()
DEV = "cuda"
N = 70016
torch.manual_seed(5)
m = NeRFNetwork(encoding="hashgrid", bound=2, cuda_ray=True, out_dim_color=3).to(DEV)
m.encoder.embeddings.data.uniform_(-1, 1)
x = torch.rand(N, 3, device=DEV) * 4 - 2
d = torch.nn.functional.normalize(torch.randn(N, 3, device=DEV), dim=-1)
params = fn.network_params(m)
cfg, offs = fn.network_cfg(m), fn.encoder_offsets(m)
for it in range(3):
    s1 = torch.full((N,), float("nan"), device=DEV); c1 = torch.full((N, 3), float("nan"), device=DEV)
    fn.nerf_forward(x, d, cfg, False, params[0], offs, *params[1:], out=(s1, c1))
    torch.cuda.synchronize()
    t = c1.view(-1, 32, 3)
    for k, name in enumerate(("frags0-7", "frags8-19", "frags20-23")):
        bad = (t[:, :, k] != t[0:1, :, k])
        rows = bad.nonzero()
        print(it, name, "bad entries", int(bad.sum()), "tiles", sorted(set(rows[:, 0].tolist()))[:8], "lanes", sorted(set(rows[:, 1].tolist()))[:40])
