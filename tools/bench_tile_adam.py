"""Microbenchmark: table gradient flush + Adam, fused (k_grid_tile_adam) vs separate (pass B + k_adam)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from enerf_amd.backends import _gridencoder as ge
from enerf_amd.gridencoder import GridEncoder
from enerf_amd.optim import FusedAdam

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

enc = GridEncoder(desired_resolution=2048 * 3).to("cuda")
p = enc.embeddings; p.grad = torch.zeros_like(p)
opt = FusedAdam([{"params": [p], "lr": 1e-2}], betas=(0.9, 0.99), eps=1e-15)
S = float(np.log2(enc.per_level_scale)); dummy = torch.empty(1, device="cuda")
B = 137856
# ray-like coherent points: short segments
base = torch.rand(B // 32, 1, 3, device="cuda"); d = torch.randn(B // 32, 1, 3, device="cuda") * 0.0005
x = (base + d * torch.arange(32, device="cuda").view(1, 32, 1)).clamp(0, 1).reshape(-1, 3).contiguous()
g = torch.randn(16, (B + 31) // 32 * 32, 2, device="cuda")
bwd = lambda defer: ge.grid_encode_backward(g, x, p.data, enc.offsets, p.grad, B, 3, 2, 16, S, 16, False, dummy, dummy, 0, layout=2, defer=defer, reserve=B)
print("adam dense (k_adam)            : %.1f us" % timeit(lambda: opt.step_now(zero_grads=True)))
print("tile adam, no records          : %.1f us" % timeit(lambda: opt.step_grid_table(p, enc.offsets, 2)))
print("backward (A+B) + adam          : %.1f us" % timeit(lambda: (bwd(False), opt.step_now(zero_grads=True))))
print("backward (A) + tile adam       : %.1f us" % timeit(lambda: (bwd(True), opt.step_grid_table(p, enc.offsets, 2))))
print("backward (A+B) alone           : %.1f us" % timeit(lambda: bwd(False)))
