"""Soak: N training steps of the bench configuration with periodic evaluation renders; reports step time drift and
device memory (torch's pool) at intervals -- hunting leaks and slow growth.  python tools/soak.py [steps]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from enerf_amd.network import NeRFNetwork  # noqa: E402
from enerf_amd.trainer import TrainHarness  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = NeRFNetwork(encoding="hashgrid", bound=3, cuda_ray=True, out_dim_color=3).to(dev)
h = TrainHarness(model, occupancy="learned", world=1)
batches = bench.build_batches(8, 4096, dev, 0, 3)
held = bench.build_batches(1, 16384, dev, 0, 3)[0]
t0 = time.time()
last = t0
for i in range(steps):
    nxt = batches[(i + 1) % 8]
    loss = h.step_rgb(*batches[i % 8], next_rays=(nxt[0], nxt[1]))
    if (i + 1) % 2000 == 0:
        model.eval()
        with torch.no_grad():
            model.render(held[0], held[1], staged=False, bg_color=None, perturb=False)
        model.train()
    if (i + 1) % 10000 == 0:
        torch.cuda.synchronize()
        now = time.time()
        print(f"step {i + 1}: {1e3 * (now - last) / 10000:.3f} ms/step, loss {float(loss):.5f}, budget {model.mean_count}, "
              f"allocated {torch.cuda.memory_allocated() / 2**20:.0f} MiB, reserved {torch.cuda.memory_reserved() / 2**20:.0f} MiB",
              flush=True)
        last = now
print("soak done")
