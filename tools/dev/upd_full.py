"""Ten full update_extra_state sweeps (development aid for profiler runs)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from enerf_amd.network import NeRFNetwork
from enerf_amd import scene
m = NeRFNetwork(encoding="hashgrid", bound=3, cuda_ray=True, out_dim_color=3).cuda()
scene.install_occupancy(m)
m.train()
for i in range(10):
    m.iter_density = 0
    m.update_extra_state()
torch.cuda.synchronize()
