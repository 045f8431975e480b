import sys, torch
sys.path.insert(0, '.')
from enerf_amd import scene, raymarching
from enerf_amd.network import NeRFNetwork
torch.manual_seed(0)
for bound in (2, 3):
    m = NeRFNetwork(encoding="hashgrid", bound=bound, cuda_ray=True, out_dim_color=3).cuda().eval()
    scene.install_occupancy(m)
    m.infer_batch_mult = 8
    inds = torch.arange(scene.H * scene.W, device='cuda')
    ro, rd = scene.pixel_rays(scene.pose(3), inds, 'cuda')
    orig = raymarching.march_rays
    log = []
    def spy(n_alive, n_step, *a, **k):
        out = orig(n_alive, n_step, *a, **k)
        real = int((out[2][:, 0] > 0).sum())
        log.append((n_alive, n_step, n_alive * n_step, real))
        return out
    raymarching.march_rays = spy
    with torch.no_grad():
        m.render(ro, rd, staged=False, bg_color=None, perturb=False)
    raymarching.march_rays = orig
    tot = sum(l[2] for l in log); real = sum(l[3] for l in log)
    print("bound", bound, "slots", tot, "real", real, "frac", real / tot)
    for l in log: print("  ", l)
