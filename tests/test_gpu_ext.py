"""The pybind11 extension modules (enerf_amd/ext: `_raymarching`, `_gridencoder`, `_shencoder`, `_ffmlp`, the boundary the
reference's own wrappers bind) on the GPU: each module's functions against the CPU oracle, and a whole training render +
backward with enerf_amd's reference-shaped wrappers running on these modules instead of the ctypes face."""
import math

import numpy as np
import pytest
import torch

from oracle import oracle as O
from util import synthetic_density_grid, camera_rays, assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"
H = 128


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.fixture(scope="module")
def ext():
    import importlib
    import sys
    from enerf_amd import ext as e
    from enerf_amd.ext import build as eb
    eb.build(verbose=False)
    e.activate()
    mods = {n: importlib.import_module(n) for n in e.MODULES}
    yield mods
    for n in e.MODULES:
        sys.modules.pop(n, None)


def test_raymarching_module_vs_oracle(ext):
    rm = ext["_raymarching"]
    bound = 2
    C = 1 + math.ceil(math.log2(bound))
    grid = synthetic_density_grid(bound, H)
    bits_ref = O.packbits(grid.reshape(-1), 0.01)
    bits = torch.empty(len(bits_ref), dtype=torch.uint8, device=DEV)
    rm.packbits(cu(grid), len(bits_ref), 0.01, bits)
    assert np.array_equal(bits.cpu().numpy(), bits_ref)
    N = 1500
    o, d = camera_rays(N, 3, bound)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    n_ref, f_ref = O.near_far_from_aabb(o, d, aabb, 0.2)
    nears, fars = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
    rm.near_far_from_aabb(cu(o), cu(d), cu(aabb), N, 0.2, nears, fars)
    assert np.array_equal(nears.cpu().numpy(), n_ref) and np.array_equal(fars.cpu().numpy(), f_ref)
    M = N * 1024
    ref = O.march_rays_train(o, d, bits_ref, bound, 0.0, 1024, C, H, M, n_ref, f_ref, 1)
    xyzs, dirs, deltas = torch.zeros(M, 3, device=DEV), torch.zeros(M, 3, device=DEV), torch.zeros(M, 2, device=DEV)
    rays = torch.empty(N, 3, dtype=torch.int32, device=DEV)
    counter = torch.zeros(2, dtype=torch.int32, device=DEV)
    rm.march_rays_train(cu(o), cu(d), bits, bound, 0.0, 1024, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays, counter, 1)
    tot = int(ref[4][0])
    assert np.array_equal(counter.cpu().numpy(), ref[4]) and np.array_equal(rays.cpu().numpy(), ref[3])
    assert np.array_equal(xyzs[:tot].cpu().numpy(), ref[0][:tot]) and np.array_equal(deltas[:tot].cpu().numpy(), ref[2][:tot])
    rng = np.random.default_rng(0)
    m = tot + 128 - tot % 128
    sig = (rng.random(m) * 20).astype(np.float32); rgb = rng.random((m, 3)).astype(np.float32)
    ws_ref, dp_ref, im_ref = O.composite_rays_train_forward(sig, rgb, ref[2][:m], ref[3])
    ws, dp, im = torch.empty(N, device=DEV), torch.empty(N, device=DEV), torch.empty(N, 3, device=DEV)
    rm.composite_rays_train_forward(cu(sig), cu(rgb), deltas[:m].contiguous(), rays, m, N, ws, dp, im)
    assert_close(ws, ws_ref, rtol=1e-4, atol=1e-6)
    assert_close(im, im_ref, rtol=1e-4, atol=1e-6)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        rm.morton3D(torch.zeros(4, 3, dtype=torch.int32), 4, torch.zeros(4, dtype=torch.int32))
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        rm.morton3D(torch.zeros(4, 3, dtype=torch.int32, device=DEV), 4, torch.zeros(4, dtype=torch.int32))
    with pytest.raises(RuntimeError, match="contiguous"):
        rm.near_far_from_aabb(cu(o).t().contiguous().t(), cu(d), cu(aabb), N, 0.2, nears, fars)     # strided view


def test_encoder_modules_vs_oracle(ext):
    ge, sh = ext["_gridencoder"], ext["_shencoder"]
    from enerf_amd.gridencoder import GridEncoder
    enc = GridEncoder(input_dim=3, num_levels=6, level_dim=2, base_resolution=4, log2_hashmap_size=9,
                      desired_resolution=96).to(DEV)
    enc.embeddings.data.uniform_(-1, 1)
    B, L, Cc = 777, 6, 2
    x = torch.rand(B, 3, device=DEV)
    S = float(np.log2(enc.per_level_scale))
    out = torch.empty(L, B, Cc, device=DEV)
    dummy = torch.empty(1, device=DEV)
    ge.grid_encode_forward(x, enc.embeddings.data, enc.offsets, out, B, 3, Cc, L, S, 4, False, dummy, 0)
    ref, _ = O.grid_encode_forward(x.cpu().numpy(), enc.embeddings.data.cpu().numpy(), enc.offsets.cpu().numpy(), S, 4)
    assert_close(out, ref, rtol=1e-5, atol=2e-6)
    g = torch.randn(L, B, Cc, device=DEV)
    gemb = torch.zeros_like(enc.embeddings.data)
    ge.grid_encode_backward(g, x, enc.embeddings.data, enc.offsets, gemb, B, 3, Cc, L, S, 4, False, dummy, dummy, 0)
    gref, _ = O.grid_encode_backward(g.cpu().numpy(), x.cpu().numpy(), enc.embeddings.data.cpu().numpy(),
                                     enc.offsets.cpu().numpy(), S, 4)
    assert_close(gemb, gref, rtol=1e-4, atol=1e-5)
    v = torch.nn.functional.normalize(torch.randn(500, 3, device=DEV), dim=-1)
    y = torch.empty(500, 16, device=DEV)
    sh.sh_encode_forward(v, y, 500, 3, 4, False, dummy)
    yref, _ = O.sh_encode_forward(v.cpu().numpy(), 4)
    assert_close(y, yref, rtol=1e-4, atol=2e-6)


def test_ffmlp_module_vs_oracle(ext):
    ff = ext["_ffmlp"]
    B, k = 256, 2
    g = torch.Generator(device=DEV).manual_seed(1)
    x = (torch.rand(B, 32, generator=g, device=DEV) - 0.5).to(torch.bfloat16)
    w = ((torch.rand(64 * 32 + 64 * 64 + 16 * 64, generator=g, device=DEV) - 0.5) * 0.4).to(torch.bfloat16)
    fb = torch.empty(k, B, 64, dtype=torch.bfloat16, device=DEV)
    y = torch.empty(B, 16, dtype=torch.bfloat16, device=DEV)
    ff.ffmlp_forward(x, w, B, 32, 16, 64, k, 0, 6, fb, y)
    yref, fbref = O.ffmlp_forward(x.float().cpu().numpy(), w.float().cpu().numpy(), 32, 16, 64, k, 0, 6, rnd=1)
    err = (y.float().cpu().numpy() - yref)
    assert np.abs(err).max() <= 4 * 2.0 ** -8 * max(1.0, np.abs(yref).max())
    y2 = torch.empty_like(y)
    ff.ffmlp_inference(x, w, B, 32, 16, 64, k, 0, 6, torch.empty(1, dtype=torch.bfloat16, device=DEV), y2)
    assert torch.equal(y, y2)
    ff.allocate_splitk(1 << 20)
    ff.free_splitk()


def test_training_render_through_the_extension_modules(ext, monkeypatch):
    """enerf_amd's reference-shaped wrappers (raymarching.py / gridencoder.py / shencoder.py: the reference's autograd
    Functions restated) with `_backend` = the pybind modules -- exactly what the reference's own wrappers would bind --
    against the same render on the CPU oracle backend: counters bit-exact, image / gradients to 1e-4."""
    import enerf_amd.raymarching as rmod, enerf_amd.gridencoder as gmod, enerf_amd.shencoder as smod
    from enerf_amd import fused_network, fused_render, density_update
    from enerf_amd.network import NeRFNetwork
    from oracle import backend as ob
    bound = 2
    bits = O.packbits(synthetic_density_grid(bound, H).reshape(-1), 0.01)
    o, d = camera_rays(160, 55, bound)

    def run(dev, backends):
        with monkeypatch.context() as mp:
            for mod, b in zip((rmod, gmod, smod), backends):
                mp.setattr(mod, "_backend", b)
            mp.setattr(rmod, "_DEVICE", dev)
            mp.setattr(fused_render, "ENABLED", False); mp.setattr(fused_network, "ENABLED", False)
            mp.setattr(density_update, "ENABLED", False)
            mp.setattr(gmod, "_layout_support", {})
            torch.manual_seed(0)
            model = NeRFNetwork(encoding="hashgrid", bound=bound, cuda_ray=True, out_dim_color=3)
            gg = torch.Generator().manual_seed(5)
            model.encoder.embeddings.data.copy_(torch.rand(model.encoder.embeddings.shape, generator=gg) * 2 - 1)
            model.density_bitfield.copy_(torch.from_numpy(bits))
            model.to(dev).train()
            ro, rd = torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev)
            out = model.render(ro, rd, staged=False, bg_color=torch.full((3,), 0.3, device=dev), perturb=True,
                               force_all_rays=True)
            (out["image"] ** 2).sum().backward()
            return (out["image"].detach().cpu(), model.step_counter.cpu().clone(),
                    {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()})

    im_ref, c_ref, g_ref = run("cpu", (ob.raymarching_backend, ob.gridencoder_backend, ob.shencoder_backend))
    im, c, g = run("cuda", (ext["_raymarching"], ext["_gridencoder"], ext["_shencoder"]))
    assert torch.equal(c, c_ref)
    assert_close(im, im_ref, rtol=1e-4, atol=2e-5)
    for n in g_ref:
        assert float((g[n] - g_ref[n]).abs().max()) < 2e-4 * float(g_ref[n].abs().max()) + 1e-7, n
