"""Train on the analytic scene with the occupancy grid maintained by update_extra_state itself and report loss, occupancy
and held-out PSNR over time (development aid; the assertion-bearing version is in tests/test_gpu_training.py)."""
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from enerf_amd.network import NeRFNetwork  # noqa: E402
from enerf_amd.trainer import TrainHarness  # noqa: E402
from test_gpu_training import _batches  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
torch.manual_seed(0)
model = NeRFNetwork(encoding="hashgrid", bound=2, cuda_ray=True, out_dim_color=3).cuda()
h = TrainHarness(model, lr=1e-2, occupancy="learned")
data = _batches(32, 4096, 2, seed=5)
held = _batches(1, 16384, 2, seed=77)[0]
t0 = time.time()
for i in range(steps):
    nxt = data[(i + 1) % len(data)]
    loss = h.step_rgb(*data[i % len(data)], next_rays=(nxt[0], nxt[1]))
    if (i + 1) % (steps // 10) == 0:
        model.eval()
        with torch.no_grad():
            img = model.render(held[0], held[1], staged=False, bg_color=None, perturb=False)["image"].reshape(-1, 3)
        model.train()
        psnr = -10 * math.log10(float(((img - held[2]) ** 2).mean()))
        occ = float((model.density_grid > min(model.mean_density, model.density_thresh)).float().mean())
        print(f"step {i + 1:5d}  loss {float(loss):.5f}  held-out PSNR {psnr:5.2f} dB  occupied {100 * occ:5.2f} %  "
              f"samples/step {model.mean_count}  {1e3 * (time.time() - t0) / (i + 1):.3f} ms/step incl. evals")
