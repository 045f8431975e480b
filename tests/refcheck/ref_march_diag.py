"""Where the reference's marcher (oracle/_ref) and the C oracle differ, if they do: per configuration, how many rays /
samples / by how many ulps.  GPU box only."""
import math, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O, build_ref as br
from util import synthetic_density_grid, camera_rays
rm = br.load("raymarching")
DEV = "cuda"; H = 128
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
for bound, dt_gamma, perturb in [(3, 1 / 128, 1), (3, 0.0, 1), (3, 1 / 128, 0), (2, 1 / 256, 1), (1, 0.0, 1), (2, 0.0, 1)]:
    C = 1 + math.ceil(math.log2(bound)); N = 4096
    bits = O.packbits(synthetic_density_grid(bound, H).reshape(-1), 0.01)
    o, d = camera_rays(N, 100 + bound, bound)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    n_o, f_o = O.near_far_from_aabb(o, d, aabb, 0.2)
    M = N * 1024
    x_o, d_o, l_o, r_o, c_o = O.march_rays_train(o, d, bits, bound, dt_gamma, 1024, C, H, M, n_o, f_o, perturb)
    xyzs, dirs, deltas = torch.zeros(M, 3, device=DEV), torch.zeros(M, 3, device=DEV), torch.zeros(M, 2, device=DEV)
    rays = torch.full((N, 3), -1, dtype=torch.int32, device=DEV); counter = torch.zeros(2, dtype=torch.int32, device=DEV)
    rm.march_rays_train(cu(o), cu(d), cu(bits), float(bound), dt_gamma, 1024, N, C, H, M, cu(n_o), cu(f_o), xyzs, dirs, deltas, rays, counter, perturb)
    r = rays.cpu().numpy(); t = r[np.argsort(r[:, 0], kind="stable")]; x = xyzs.cpu().numpy(); dl = deltas.cpu().numpy()
    bad_rays = 0; bad_samples = 0; maxulp = 0; first = None
    for n in range(N):
        cnt = t[n, 2]; co = r_o[n, 2]
        if cnt != co:
            bad_rays += 1; first = first or (n, "count", cnt, co); continue
        if cnt == 0: continue
        a = x[t[n, 1]:t[n, 1] + cnt]; b = x_o[r_o[n, 1]:r_o[n, 1] + cnt]
        la = dl[t[n, 1]:t[n, 1] + cnt]; lb = l_o[r_o[n, 1]:r_o[n, 1] + cnt]
        ne = (a != b).any(axis=1) | (la != lb).any(axis=1)
        if ne.any():
            bad_rays += 1; bad_samples += int(ne.sum())
            ulp = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64)).max()
            maxulp = max(maxulp, int(ulp))
            if first is None:
                k = int(np.argmax(ne)); first = (n, "sample", k, a[k].tolist(), b[k].tolist(), la[k].tolist(), lb[k].tolist(), float(n_o[n]))
    print(f"bound {bound} dt_gamma {dt_gamma} perturb {perturb}: counters ref {counter.cpu().numpy()} oracle {c_o}; rays differing {bad_rays}/{N}, samples {bad_samples}, max ulp {maxulp}; first {first}", flush=True)
