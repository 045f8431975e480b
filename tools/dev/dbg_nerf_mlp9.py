import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from enerf_amd import _lib, fused_network as fn
from enerf_amd.network import NeRFNetwork
lib = _lib.lib()
DEV = "cuda"
N = 133000
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
shown = 0
for it in range(200):
    s1 = torch.empty(N, device=DEV); c1 = torch.full((N, 3), float("nan"), device=DEV)
    fn.nerf_forward(x, d, cfg, False, params[0], offs, *params[1:], out=(s1, c1))
    dc = (c1 - c0).abs().max(dim=1).values
    bad = ((dc > 2e-5) | dc.isnan()).nonzero().flatten()
    if bad.numel() and shown < 3:
        shown += 1
        t = int(bad[0]) // 32
        rows = torch.arange(t * 32, t * 32 + 32, device=DEV)
        print("iteration", it, "tile", t, "bad rows in tile", [int(b) - t * 32 for b in bad if int(b) // 32 == t])
        logit = lambda p: torch.log(p / (1 - p))
        dl = (logit(c1[rows].double()) - logit(c0[rows].double()))
        for r in range(32):
            print(r, ["%.6f" % v for v in c1[rows[r]].tolist()], ["%.6f" % v for v in c0[rows[r]].tolist()], "dlogit", ["%+.4f" % v for v in dl[r].tolist()])
        # does a bad row equal some other row of the reference?
        b0 = int(bad[0])
        eq = (c0 - c1[b0]).abs().max(dim=1).values
        print("closest reference row to bad row", b0, ":", int(eq.argmin()), float(eq.min()))
