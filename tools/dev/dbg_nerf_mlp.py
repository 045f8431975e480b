import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from enerf_amd import _lib, fused_network as fn
from enerf_amd.network import NeRFNetwork
lib = _lib.lib()
DEV = "cuda"
for N in (4096, 8192, 20000, 70001):
    torch.manual_seed(5)
    m = NeRFNetwork(encoding="hashgrid", bound=2, cuda_ray=True, out_dim_color=3).to(DEV)
    m.encoder.embeddings.data.uniform_(-1, 1)
    x = torch.rand(N, 3, device=DEV) * 4 - 2
    d = torch.nn.functional.normalize(torch.randn(N, 3, device=DEV), dim=-1)
    params = fn.network_params(m)
    cfg, offs = fn.network_cfg(m), fn.encoder_offsets(m)
    out = {}
    for fused in (1, 0):
        prev = lib.enerf_debug_nerf_mlp_fused(fused)
        s, c, sv = fn.nerf_forward(x, d, cfg, True, params[0], offs, *params[1:])
        lib.enerf_debug_nerf_mlp_fused(prev)
        out[fused] = (s.clone(), c.clone())
    torch.cuda.synchronize()
    ds = ((out[1][0] - out[0][0]).abs() / out[0][0].abs().clamp(min=1e-6))
    dc = (out[1][1] - out[0][1]).abs().max(dim=1).values
    bad = (dc > 2e-5).nonzero().flatten()
    print(N, "sigma rel max", float(ds.max()), "rgb max", float(dc.max()), "bad rows", bad.numel(), bad[:20].tolist(),
          "tiles", sorted(set((bad // 32).tolist()))[:10], "lanes", sorted(set((bad % 32).tolist()))[:32])
