import sys, torch
sys.path.insert(0, "/root/repo")
import bench
from enerf_amd.network import NeRFNetwork
from enerf_amd.trainer import TrainHarness
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = NeRFNetwork(encoding="hashgrid", bound=3, cuda_ray=True, out_dim_color=3).to(dev)
h = TrainHarness(model, occupancy="synthetic")
batches = bench.build_batches(8, 4096, dev, 0, 3)
cs = []
for i in range(24):
    h.step_rgb(*batches[i % 8])
    cs.append(int(model.step_counter[model.rendered_counter_slot, 0]))
print(cs[:8], "budget", model.mean_count)
