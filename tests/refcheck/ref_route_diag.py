"""Why is route B of tools/psnr_ab.py slow on the reference's grid kernel?  One short route-B training (learned occupancy,
bound 2, tools/psnr_ab.py's batches) under rocprofv3 --kernel-trace --stats.  TEST INFRASTRUCTURE."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import build_ref as br
import enerf_amd.raymarching as rmod, enerf_amd.gridencoder as gmod, enerf_amd.shencoder as smod
from enerf_amd import fused_network as fn_, fused_render as fr_, density_update as du_
from enerf_amd.network import NeRFNetwork
from enerf_amd.trainer import TrainHarness
from test_gpu_training import _batches
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rmod._backend, gmod._backend, smod._backend = (br.load(n) for n in ("raymarching", "gridencoder", "shencoder"))
fr_.ENABLED = fn_.ENABLED = du_.ENABLED = False
data = _batches(8, 4096, 2, seed=5)
torch.manual_seed(0)
m = NeRFNetwork(encoding="hashgrid", bound=2, cuda_ray=True, out_dim_color=3).cuda()
h = TrainHarness(m, lr=1e-2, occupancy="learned", optimizer=torch.optim.Adam)
h.native_step = h.manual_mse = h.fuse_table_adam = h.prefetch = False
torch.cuda.synchronize(); t0 = time.time()
for i in range(steps):
    h.step_rgb(*data[i % 8])
    if i % 16 == 15:
        torch.cuda.synchronize(); print(i, round((time.time() - t0) / 16 * 1e3, 2), "ms/step", int(m.mean_count), flush=True); t0 = time.time()
