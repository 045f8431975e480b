"""Density-only sigma head over a sweep-sized batch at different workgroup counts (development aid)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enerf_amd import _lib as L, fused_network
from enerf_amd.network import NeRFNetwork
torch.manual_seed(0)
m = NeRFNetwork(encoding="hashgrid", bound=3, cuda_ray=True, out_dim_color=3).cuda()
x = (torch.rand(3 * 128 ** 3, 3, device="cuda") * 2 - 1) * 3
for blocks in (0, 512, 768, 1024, 1536, 2048):
    L.lib().enerf_debug_mlp32_grid_caps(blocks, 0)
    for _ in range(2):
        fused_network.density_sigma(m, x)
    torch.cuda.synchronize()
    L.prof.reset(); L.prof.enable(True, only=("ffmlp_fwd",))
    for _ in range(5):
        fused_network.density_sigma(m, x)
    torch.cuda.synchronize()
    L.prof.enable(False)
    ms, n = L.prof.read("ffmlp_fwd")
    print(f"fwd blocks {blocks:5d}: sigma head {1e3 * ms / n:7.1f} us")
