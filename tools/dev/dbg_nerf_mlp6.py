import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from enerf_amd import _lib, fused_network as fn
from enerf_amd.network import NeRFNetwork
lib = _lib.lib()
DEV = "cuda"
for N in (70001, 133000):
    torch.manual_seed(1)
    m = NeRFNetwork(encoding="hashgrid", bound=2, cuda_ray=True, out_dim_color=3).to(DEV)
    m.encoder.embeddings.data.uniform_(-1, 1)
    x = torch.rand(N, 3, device=DEV) * 4 - 2
    d = torch.nn.functional.normalize(torch.randn(N, 3, device=DEV), dim=-1)
    params = fn.network_params(m)
    cfg, offs = fn.network_cfg(m), fn.encoder_offsets(m)
    lib.enerf_debug_nerf_mlp_fused(0)
    s0 = torch.empty(N, device=DEV); c0 = torch.empty(N, 3, device=DEV)
    fn.nerf_forward(x, d, cfg, False, params[0], offs, *params[1:], out=(s0, c0))
    lib.enerf_debug_nerf_mlp_fused(1)
    for cap in (256, 512, 768, 1024):
        lib.enerf_debug_mlp32_grid_caps(cap, 0)
        nb = 0; lanes = set(); tiles = []; runs = 0
        for it in range(100):
            s1 = torch.empty(N, device=DEV); c1 = torch.full((N, 3), float("nan"), device=DEV)
            fn.nerf_forward(x, d, cfg, False, params[0], offs, *params[1:], out=(s1, c1))
            dc = (c1 - c0).abs().max(dim=1).values
            bad = ((dc > 2e-5) | dc.isnan()).nonzero().flatten()
            if bad.numel():
                runs += 1; nb += bad.numel(); lanes |= set((bad % 32).tolist()); tiles += (bad // 32).tolist()
        grid = min((N + 127) // 128, cap)
        print(f"N {N} cap {cap} grid {grid}: bad rows {nb} in {runs}/100 runs; lanes {sorted(lanes)[:6]}..{sorted(lanes)[-3:] if lanes else []}; WGs {sorted(set(((t % (grid*4))//4) for t in tiles))[:12]} rounds {sorted(set(t // (grid*4) for t in tiles))}")
    lib.enerf_debug_mlp32_grid_caps(0, 0)
