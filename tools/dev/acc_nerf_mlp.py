"""accuracy of the two-nets-in-one-launch kernels vs one launch per net, both against an fp64 nn.Linear loop"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from enerf_amd import _lib, fused_network as fn
from enerf_amd.network import NeRFNetwork
lib = _lib.lib()
DEV = "cuda"
N = 133000
for seed, scale in ((1, 1.0), (2, 1.0), (3, 0.1)):
    torch.manual_seed(seed)
    m = NeRFNetwork(encoding="hashgrid", bound=2, cuda_ray=True, out_dim_color=3).to(DEV)
    m.encoder.embeddings.data.uniform_(-scale, scale)
    x = torch.rand(N, 3, device=DEV) * 4 - 2
    d = torch.nn.functional.normalize(torch.randn(N, 3, device=DEV), dim=-1)
    params = fn.network_params(m)
    cfg, offs = fn.network_cfg(m), fn.encoder_offsets(m)
    with torch.no_grad():
        ws = [p.double() for p in params[1:]]
        feat = m.encoder(x, bound=m.bound).double()
        h = torch.relu(feat @ ws[0].t()) @ ws[1].t()
        sh = m.encoder_dir(d).double()
        hc = torch.relu(torch.relu(torch.cat([sh, h[:, 1:]], dim=1) @ ws[2].t()) @ ws[3].t()) @ ws[4].t()
        sref, cref = torch.exp(h[:, 0]), torch.sigmoid(hc)
    for fused in (1, 0):
        lib.enerf_debug_nerf_mlp_fused(fused)
        s = torch.empty(N, device=DEV); c = torch.empty(N, 3, device=DEV)
        fn.nerf_forward(x, d, cfg, False, params[0], offs, *params[1:], out=(s, c))
        es = ((s.double() - sref).abs() / sref.clamp(min=1e-6))
        ec = (c.double() - cref).abs()
        print(f"seed {seed} emb +-{scale}: {'fused ' if fused else 'per-net'} sigma rel max {float(es.max()):.2e} mean {float(es.mean()):.2e} | rgb abs max {float(ec.max()):.2e} mean {float(ec.mean()):.2e} rms {float((ec**2).mean().sqrt()):.2e}")
    lib.enerf_debug_nerf_mlp_fused(1)
