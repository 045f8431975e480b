"""at::Half tables: the product's grid kernels beside the reference's (oracle/_ref/_ref_gridencoder) level by level, and which
arithmetic each side's Half accumulation follows -- (a) forward / Jacobian: equal entries per level (1.0 since the product
pins the float product: hipcc folds multiply + convert into v_fma_mixlo_f16, which rounds once); (b) grad_inputs: the
reference AS BUILT contracts Half's multiply-add into v_fma_f16 (99.99 % of its entries equal that emulation), its source
rounds the product first (what the product does, 100 %).  TEST INFRASTRUCTURE.
    gpurun -- 'python tests/refcheck/half_diag.py'
"""
import sys, os, importlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from oracle import oracle as O, build_ref as br
from enerf_amd import ext as E
from enerf_amd.ext import build as eb
eb.build(verbose=False); E.activate()
prod = importlib.import_module("_gridencoder")
ref = br.load("gridencoder")
C = 2
offsets, pls = O.grid_offsets(level_dim=C, desired_resolution=2048)
S = float(np.log2(pls)); L = 16; B = 50000
rng = np.random.default_rng(61)
emb = rng.uniform(-1, 1, (int(offsets[-1]), C)).astype(np.float16)
x = rng.uniform(0, 1, (B, 3)).astype(np.float32)
cu = lambda a: torch.from_numpy(a).cuda()
cx, ce, co = cu(x), cu(emb), cu(offsets)
outs = []
for m in (ref, prod):
    out = torch.empty(L, B, C, device="cuda", dtype=torch.half); j = torch.empty(B, L * 3 * C, device="cuda", dtype=torch.half)
    m.grid_encode_forward(cx, ce, co, out, B, 3, C, L, S, 16, True, j, 0)
    outs.append((out, j.view(B, L, 3 * C)))
for l in range(L):
    a, b = outs[0][0][l], outs[1][0][l]
    ja, jb = outs[0][1][:, l], outs[1][1][:, l]
    bad = (a != b).nonzero()
    print(l, "out equal", float((a == b).float().mean()), "max", float((a.float() - b.float()).abs().max()),
          "jac equal", float((ja == jb).float().mean()), "max", float((ja.float() - jb.float()).abs().max()),
          "first bad", (bad[0].tolist(), float(a[tuple(bad[0])]), float(b[tuple(bad[0])])) if len(bad) else None)
g = rng.normal(size=(L, B, C)).astype(np.float16)
res = []
for m in (ref, prod):
    ge = torch.zeros_like(ce); gi = torch.zeros(B, 3, device="cuda", dtype=torch.half)
    m.grid_encode_backward(cu(g), cx, ce, co, ge, B, 3, C, L, S, 16, True, outs[0][1].reshape(B, -1).contiguous(), gi, 0)
    res.append(gi)
a, b = res
print("grad_inputs equal", float((a == b).float().mean()), "max diff", float((a.float() - b.float()).abs().max()), "max", float(a.float().abs().max()),
      "nan/inf ref", int((~torch.isfinite(a)).sum()), "prod", int((~torch.isfinite(b)).sum()))
bad = (a != b).nonzero()
for t in bad[:5]:
    print(t.tolist(), float(a[tuple(t)]), float(b[tuple(t)]))
J = outs[0][1].view(B, L, 3, C)
G = cu(g)
r_fma = torch.zeros(B, 3, device="cuda", dtype=torch.half); r_src = torch.zeros_like(r_fma)
for l in range(L):
    for ch in range(C):
        gg = G[l, :, ch].double()[:, None]; jj = J[:, l, :, ch].double()
        r_fma = (r_fma.double() + gg * jj).half()
        r_src = (r_src.float() + (gg * jj).half().float()).half()
print("reference build == half fma (single rounding):", float((a == r_fma).float().mean()), " == source semantics:", float((a == r_src).float().mean()))
print("product == half fma:", float((b == r_fma).float().mean()), " == source semantics:", float((b == r_src).float().mean()))
