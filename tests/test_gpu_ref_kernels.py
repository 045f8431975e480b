"""The reference's OWN kernels on the same GPU (oracle/_ref: `raymarching.cu` and `shencoder.cu` built for gfx950 by
oracle/build_ref.py -- the translator that ships inside this image's PyTorch + hipcc, nothing written by us) against the C
oracle and against the product's kernels, same inputs.  This is what pins the oracle's marcher to the reference itself
(rows a1-a11, a15 of SURVEY.md section 8): before it, `march_rays_train` / `march_rays` had only builder-written second
statements.  `gridencoder.cu` has its own file (tests/test_gpu_ref_gridencoder.py; built since round 4, the recipe says
how); `ffmlp` needs CUTLASS + nvcuda::wmma: no stand-in was written, it stays on the pins DESIGN.md section 2 lists.

The reference's marcher hands out sample rows and ray-table rows by atomicAdd, so WHICH rows a ray gets differs from run to
run; every comparison below is per ray (the table sorted by its ray index, each ray's samples read from its own offset).
"""
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
def ref():
    from oracle import build_ref as br
    br.build()
    try:
        return {"rm": br.load("raymarching"), "sh": br.load("shencoder"), "ge": br.load("gridencoder")}
    except ImportError as e:                                     # never built: /root/reference is not on this machine
        pytest.skip(f"oracle/_ref not built: {e}")


@pytest.fixture(scope="module")
def prod():
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


@pytest.fixture(scope="module")
def scenes():
    out = {}
    for bound in (1, 2, 3):
        grid = synthetic_density_grid(bound, H)
        out[bound] = (grid, O.packbits(grid.reshape(-1), 0.01), 1 + math.ceil(math.log2(bound)))
    return out


def _rays(N, seed, bound):
    o, d = camera_rays(N, seed, bound)
    # a few special rays: axis-parallel directions (infinite reciprocals), a ray that misses the box
    d[0] = (1.0, 0.0, 0.0); d[1] = (0.0, -1.0, 0.0); d[2] = (0.0, 0.0, 1.0)
    o[3] = (5.0 * bound, 5.0 * bound, 5.0 * bound); d[3] = (0.0, 1.0, 0.0)
    return o, d, np.array([-bound] * 3 + [bound] * 3, np.float32)


def test_reference_utilities_equal_oracle_and_product(ref, prod):
    rm, pm = ref["rm"], prod["_raymarching"]
    for bound in (1, 2, 3):
        N = 5000
        o, d, aabb = _rays(N, 11 + bound, bound)
        n_o, f_o = O.near_far_from_aabb(o, d, aabb, 0.2)
        n_r, f_r = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
        rm.near_far_from_aabb(cu(o), cu(d), cu(aabb), N, 0.2, n_r, f_r)
        n_p, f_p = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
        pm.near_far_from_aabb(cu(o), cu(d), cu(aabb), N, 0.2, n_p, f_p)
        assert np.array_equal(n_r.cpu().numpy(), n_o) and np.array_equal(f_r.cpu().numpy(), f_o)
        assert torch.equal(n_r, n_p) and torch.equal(f_r, f_p)
        c_o = O.polar_from_ray(o, d, float(bound))
        c_r = torch.empty(N, 2, device=DEV)
        rm.polar_from_ray(cu(o), cu(d), float(bound), N, c_r)
        ok = np.isfinite(c_o).all(axis=1)
        np.testing.assert_allclose(c_r.cpu().numpy()[ok], c_o[ok], rtol=1e-5, atol=1e-5)
    rng = np.random.default_rng(0)
    coords = rng.integers(0, 1024, size=(100003, 3)).astype(np.int32)
    m_r = torch.empty(len(coords), dtype=torch.int32, device=DEV)
    rm.morton3D(cu(coords), len(coords), m_r)
    assert np.array_equal(m_r.cpu().numpy(), O.morton3D(coords))
    back = torch.empty(len(coords), 3, dtype=torch.int32, device=DEV)
    rm.morton3D_invert(m_r, len(coords), back)
    assert np.array_equal(back.cpu().numpy(), coords) and np.array_equal(O.morton3D_invert(m_r.cpu().numpy()), coords)
    grid = (rng.random(3 * H ** 3) * 0.02).astype(np.float32)
    bits = torch.empty(len(grid) // 8, dtype=torch.uint8, device=DEV)
    rm.packbits(cu(grid), len(grid) // 8, 0.01, bits)
    assert np.array_equal(bits.cpu().numpy(), O.packbits(grid, 0.01))


def _per_ray(rays, N):
    """ray table in the order of the ray index (the reference fills it in the order its atomics landed)"""
    rays = np.asarray(rays).reshape(-1, 3)
    order = np.argsort(rays[:, 0], kind="stable")
    t = rays[order]
    assert np.array_equal(t[:, 0], np.arange(N)), "every ray owns exactly one row of the table"
    return t


def _gather(buf, table, width):
    rows = np.concatenate([np.arange(o, o + n) for _, o, n in table if n > 0] or [np.zeros(0, np.int64)]).astype(np.int64)
    return buf[rows].reshape(-1, width)


@pytest.mark.parametrize("bound,dt_gamma,perturb,N", [(1, 0.0, 0, 4096), (2, 0.0, 0, 4096), (3, 0.0, 0, 4096),
                                                      (2, 1.0 / 256, 0, 4096), (3, 1.0 / 128, 1, 4096),
                                                      (3, 0.0, 1, 4096), (1, 0.0, 1, 1777)])
def test_reference_march_rays_train_equals_oracle_and_product(ref, prod, scenes, bound, dt_gamma, perturb, N):
    """every ray's sample count, positions, directions, step sizes and real deltas: reference kernel == C oracle == product
    kernel, bit for bit (BASELINE configs[1] shape at bound 3: 4096 rays, 128^3 x C grid, max_steps 1024)."""
    rm, pm = ref["rm"], prod["_raymarching"]
    grid, bits, C = scenes[bound]
    o, d, aabb = _rays(N, 100 + bound, bound)
    n_o, f_o = O.near_far_from_aabb(o, d, aabb, 0.2)
    M = N * 1024
    x_o, d_o, l_o, r_o, c_o = O.march_rays_train(o, d, bits, bound, dt_gamma, 1024, C, H, M, n_o, f_o, perturb)
    outs = []
    for mod in (rm, pm):
        xyzs, dirs = torch.zeros(M, 3, device=DEV), torch.zeros(M, 3, device=DEV)
        deltas = torch.zeros(M, 2, device=DEV)
        rays = torch.full((N, 3), -1, dtype=torch.int32, device=DEV)
        counter = torch.zeros(2, dtype=torch.int32, device=DEV)
        mod.march_rays_train(cu(o), cu(d), cu(bits), float(bound), dt_gamma, 1024, N, C, H, M, cu(n_o), cu(f_o), xyzs, dirs,
                             deltas, rays, counter, perturb)
        outs.append((xyzs.cpu().numpy(), dirs.cpu().numpy(), deltas.cpu().numpy(), rays.cpu().numpy(),
                     counter.cpu().numpy()))
    t_o = _per_ray(r_o, N)
    assert int(c_o[0]) > 4 * N                                    # a real workload
    for name, (x, dd, dl, rays, counter) in zip(("reference", "product"), outs):
        assert np.array_equal(counter, c_o), (name, counter, c_o)
        t = _per_ray(rays, N)
        assert np.array_equal(t[:, 2], t_o[:, 2]), f"{name}: per-ray sample counts"
        assert np.array_equal(_gather(x, t, 3), _gather(x_o, t_o, 3)), f"{name}: positions"
        assert np.array_equal(_gather(dd, t, 3), _gather(d_o, t_o, 3)), f"{name}: directions"
        assert np.array_equal(_gather(dl, t, 2), _gather(l_o, t_o, 2)), f"{name}: dt / real delta"
    # the product's table is in ray order with offsets = running sums (what the oracle does too)
    assert np.array_equal(outs[1][3], r_o)


def test_reference_march_rays_train_overflowing_budget(ref, scenes):
    """M smaller than the samples marched: the reference drops a ray whose range reaches M (`>=`), counters still count
    everything.  Which rays are dropped depends on the order its atomics landed, so only order-free facts are compared."""
    rm = ref["rm"]
    bound = 2
    grid, bits, C = scenes[bound]
    N = 2048
    o, d, aabb = _rays(N, 77, bound)
    n_o, f_o = O.near_far_from_aabb(o, d, aabb, 0.2)
    full = O.march_rays_train(o, d, bits, bound, 0.0, 1024, C, H, N * 1024, n_o, f_o, 0)
    M = int(full[4][0]) // 2
    x_o, d_o, l_o, r_o, c_o = O.march_rays_train(o, d, bits, bound, 0.0, 1024, C, H, M, n_o, f_o, 0)
    xyzs, dirs, deltas = torch.zeros(M, 3, device=DEV), torch.zeros(M, 3, device=DEV), torch.zeros(M, 2, device=DEV)
    rays = torch.full((N, 3), -1, dtype=torch.int32, device=DEV)
    counter = torch.zeros(2, dtype=torch.int32, device=DEV)
    rm.march_rays_train(cu(o), cu(d), cu(bits), float(bound), 0.0, 1024, N, C, H, M, cu(n_o), cu(f_o), xyzs, dirs, deltas,
                        rays, counter, 0)
    assert np.array_equal(counter.cpu().numpy(), c_o) and int(c_o[0]) == int(full[4][0])
    t = _per_ray(rays.cpu().numpy(), N)
    assert np.array_equal(t[:, 2], _per_ray(r_o, N)[:, 2])
    x = xyzs.cpu().numpy()
    full_t = _per_ray(full[3], N)
    for n, off, num in t[:400]:
        if num == 0:
            continue
        if off + num >= M:                                         # dropped: nothing of it was written
            continue
        fo = full_t[n][1]
        assert np.array_equal(x[off:off + num], full[0][fo:fo + num])


def test_reference_composite_train_equals_oracle_and_product(ref, prod, scenes):
    rm, pm = ref["rm"], prod["_raymarching"]
    bound = 3
    grid, bits, C = scenes[bound]
    N = 4096
    o, d, aabb = _rays(N, 9, bound)
    n_o, f_o = O.near_far_from_aabb(o, d, aabb, 0.2)
    x_o, d_o, l_o, r_o, c_o = O.march_rays_train(o, d, bits, bound, 0.0, 1024, C, H, N * 1024, n_o, f_o, 0)
    tot = int(c_o[0]); m = tot + 128 - tot % 128
    rng = np.random.default_rng(3)
    sig = (rng.random(m) * 25).astype(np.float32); rgb = rng.random((m, 3)).astype(np.float32)
    ws_o, dp_o, im_o = O.composite_rays_train_forward(sig, rgb, l_o[:m], r_o)
    g_ws = rng.standard_normal(N).astype(np.float32); g_im = rng.standard_normal((N, 3)).astype(np.float32)
    gs_o, gc_o = O.composite_rays_train_backward(g_ws, g_im, sig, rgb, l_o[:m], r_o, ws_o, im_o)
    res = []
    for mod in (rm, pm):
        ws, dp, im = torch.empty(N, device=DEV), torch.empty(N, device=DEV), torch.empty(N, 3, device=DEV)
        mod.composite_rays_train_forward(cu(sig), cu(rgb), cu(l_o[:m]), cu(r_o), m, N, ws, dp, im)
        gs, gc = torch.zeros(m, device=DEV), torch.zeros(m, 3, device=DEV)
        mod.composite_rays_train_backward(cu(g_ws), cu(g_im), cu(sig), cu(rgb), cu(l_o[:m]), cu(r_o), ws, im, m, N, gs, gc)
        res.append((ws, dp, im, gs, gc))
    for ws, dp, im, gs, gc in res:
        assert_close(ws, ws_o, rtol=1e-5, atol=1e-6); assert_close(dp, dp_o, rtol=1e-5, atol=1e-5)
        assert_close(im, im_o, rtol=1e-5, atol=1e-6)
        assert_close(gs, gs_o, rtol=1e-4, atol=1e-6); assert_close(gc, gc_o, rtol=1e-5, atol=1e-6)
    # the product against the reference kernel directly
    for a, b in zip(res[0], res[1]):
        assert_close(b, a.cpu().numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("dt_gamma", [0.0, 1.0 / 256])
def test_reference_inference_loop_equals_oracle(ref, scenes, dt_gamma):
    """march_rays / composite_rays / compact_rays for the whole loop of `run_cuda`'s inference branch: every round's samples
    bit-exact against the oracle, same rays terminate, same survivors (the reference compacts through an atomic counter:
    its survivors come in any order, so both sides continue from the oracle's order)."""
    rm = ref["rm"]
    bound = 2
    grid, bits, C = scenes[bound]
    N = 3000
    o, d, aabb = _rays(N, 40, bound)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.2)
    rng = np.random.default_rng(6)
    ws = np.zeros(N, np.float32); dp = np.zeros(N, np.float32); im = np.zeros((N, 3), np.float32)
    alive = np.arange(N, dtype=np.int32); rt = nears.copy()
    g_ws = torch.zeros(N, device=DEV); g_dp = torch.zeros(N, device=DEV); g_im = torch.zeros(N, 3, device=DEV)
    co, cd, cb, cn, cf = cu(o), cu(d), cu(bits), cu(nears), cu(fars)
    n_alive, rounds = N, 0
    g_alive, g_rt = cu(alive), cu(rt)
    for it in range(200):
        if it > 0:
            new_alive, new_t, n_new = O.compact_rays(n_alive, alive, rt)
            cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
            r_alive = torch.zeros(n_alive, dtype=torch.int32, device=DEV); r_t = torch.zeros(n_alive, device=DEV)
            rm.compact_rays(n_alive, r_alive, g_alive, r_t, g_rt, cnt)
            assert int(cnt.item()) == n_new
            got = sorted(zip(r_alive[:n_new].cpu().tolist(), r_t[:n_new].cpu().tolist()))
            want = sorted(zip(new_alive[:n_new].tolist(), new_t[:n_new].tolist()))
            assert got == want
            alive, rt, n_alive = new_alive[:n_new].copy(), new_t[:n_new].copy(), n_new
            g_alive, g_rt = cu(alive), cu(rt)
        if n_alive <= 0:
            break
        n_step = max(min(N // n_alive, 8), 1)
        Mi = n_alive * n_step; Mi += 128 - Mi % 128
        perturb = 0 if it % 2 == 0 else 1
        x, dd, dl = O.march_rays(n_alive, n_step, alive, rt, o, d, bound, dt_gamma, 1024, C, H, bits, nears, fars, Mi, perturb)
        gx = torch.zeros(Mi, 3, device=DEV); gd = torch.zeros(Mi, 3, device=DEV); gl = torch.zeros(Mi, 2, device=DEV)
        rm.march_rays(n_alive, n_step, g_alive, g_rt, co, cd, float(bound), dt_gamma, 1024, C, H, cb, cn, cf, gx, gd, gl, perturb)
        assert np.array_equal(gx.cpu().numpy(), x) and np.array_equal(gl.cpu().numpy(), dl)
        assert np.array_equal(gd.cpu().numpy(), dd)
        sig = (rng.random(Mi) * 30).astype(np.float32); rgb = rng.random((Mi, 3)).astype(np.float32)
        rt_o = rt.copy()
        O.composite_rays(n_alive, n_step, alive, rt_o, sig, rgb, dl, ws, dp, im)
        rm.composite_rays(n_alive, n_step, g_alive, g_rt, cu(sig), cu(rgb), gl, g_ws, g_dp, g_im)
        got_t = g_rt[:n_alive].cpu().numpy()
        assert np.array_equal(got_t < 0, rt_o < 0)
        np.testing.assert_allclose(got_t, rt_o, rtol=1e-6, atol=1e-6)
        rt = rt_o
        g_rt = cu(rt)
        rounds += 1
    assert rounds > 5
    assert_close(g_ws, ws, rtol=1e-4, atol=1e-6); assert_close(g_im, im, rtol=1e-4, atol=1e-6)
    assert_close(g_dp, dp, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("bound,N,perturb", [(3, 65536, 1), (2, 307200, 0)])
def test_reference_march_at_full_sizes_equals_product(ref, prod, scenes, bound, N, perturb):
    """BASELINE's largest shapes -- configs[3]'s 65 536 rays (bound 3) and configs[4]'s 640 x 480 = 307 200 rays (bound 2,
    23 M samples) -- marched by the reference's kernel and by the product's: every ray's sample count and every sample
    bit for bit (no oracle in between: it would take minutes at this size; the oracle is held to the same kernel at 4096
    rays above), then composited by both (1e-5)."""
    rm, pm = ref["rm"], prod["_raymarching"]
    grid, bits, C = scenes[bound]
    o, d, aabb = _rays(N, 900 + bound, bound)
    co, cd, cb = cu(o), cu(d), cu(bits)
    nears, fars = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
    rm.near_far_from_aabb(co, cd, cu(aabb), N, 0.2, nears, fars)
    M = N * 128
    outs = []
    for mod in (rm, pm):
        xyzs, deltas = torch.zeros(M, 3, device=DEV), torch.zeros(M, 2, device=DEV)
        dirs = torch.zeros(M, 3, device=DEV)
        rays = torch.full((N, 3), -1, dtype=torch.int32, device=DEV)
        counter = torch.zeros(2, dtype=torch.int32, device=DEV)
        mod.march_rays_train(co, cd, cb, float(bound), 0.0, 1024, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays, counter,
                             perturb)
        outs.append((xyzs, deltas, rays, counter))
        del dirs
    (x_r, l_r, rays_r, c_r), (x_p, l_p, rays_p, c_p) = outs
    assert torch.equal(c_r, c_p) and int(c_r[0]) + 128 < M and int(c_r[0]) > 30 * N
    # canonical order on the device: sort the reference's table by ray, gather each ray's rows
    order = torch.argsort(rays_r[:, 0].long(), stable=True)
    tr = rays_r[order].long()
    tp = rays_p.long()
    assert torch.equal(tr[:, 0], torch.arange(N, device=DEV)) and torch.equal(tp[:, 0], tr[:, 0])
    assert torch.equal(tr[:, 2], tp[:, 2])                                          # per-ray sample counts
    tot = int(c_r[0])
    ray_of_row = torch.repeat_interleave(torch.arange(N, device=DEV), tp[:, 2])      # product rows are in ray order
    within = torch.arange(tot, device=DEV) - tp[ray_of_row, 1]
    rows_r = tr[ray_of_row, 1] + within
    assert torch.equal(x_r[rows_r], x_p[:tot]) and torch.equal(l_r[rows_r], l_p[:tot])
    # compositing of the same samples by both (each on its own table / row order)
    m = tot + 128 - tot % 128
    g = torch.Generator(device=DEV).manual_seed(5)
    sig_p = torch.rand(m, device=DEV, generator=g) * 25
    rgb_p = torch.rand(m, 3, device=DEV, generator=g)
    sig_r, rgb_r = torch.zeros(M, device=DEV), torch.zeros(M, 3, device=DEV)
    sig_r[rows_r] = sig_p[:tot]; rgb_r[rows_r] = rgb_p[:tot]
    res = []
    for mod, sig, rgb, dl, rays, mm in ((rm, sig_r, rgb_r, l_r, rays_r, M), (pm, sig_p, rgb_p, l_p[:m].contiguous(), rays_p, m)):
        ws, dp, im = torch.empty(N, device=DEV), torch.empty(N, device=DEV), torch.empty(N, 3, device=DEV)
        mod.composite_rays_train_forward(sig, rgb, dl, rays, mm, N, ws, dp, im)
        res.append((ws, dp, im))
    for a, b in zip(res[0], res[1]):
        assert_close(b, a.cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("degree", [1, 2, 3, 4, 5, 6, 7, 8])
def test_reference_sh_encode_equals_oracle_and_product(ref, prod, degree):
    sh, ps = ref["sh"], prod["_shencoder"]
    B = 4099
    g = torch.Generator(device=DEV).manual_seed(degree)
    v = torch.randn(B, 3, generator=g, device=DEV)
    v[: B // 2] = torch.nn.functional.normalize(v[: B // 2], dim=-1)          # on and off the unit sphere
    out_dim = degree * degree
    y_r, y_p = torch.empty(B, out_dim, device=DEV), torch.empty(B, out_dim, device=DEV)
    j_r, j_p = torch.empty(B, 3 * out_dim, device=DEV), torch.empty(B, 3 * out_dim, device=DEV)
    sh.sh_encode_forward(v, y_r, B, 3, degree, True, j_r)
    ps.sh_encode_forward(v, y_p, B, 3, degree, True, j_p)
    y_o, j_o = O.sh_encode_forward(v.cpu().numpy(), degree, True)
    scale = max(1.0, float(np.abs(y_o).max()))
    assert_close(y_r, y_o, rtol=1e-5, atol=2e-6 * scale)
    assert_close(y_p, y_r.cpu().numpy(), rtol=1e-5, atol=2e-6 * scale)
    jscale = max(1.0, float(np.abs(j_o).max()))
    assert_close(j_r, j_o.reshape(B, -1), rtol=1e-5, atol=4e-6 * jscale)
    assert_close(j_p, j_r.cpu().numpy(), rtol=1e-5, atol=4e-6 * jscale)
    grad = torch.randn(B, out_dim, generator=g, device=DEV)
    gi_r, gi_p = torch.zeros(B, 3, device=DEV), torch.zeros(B, 3, device=DEV)      # the reference ACCUMULATES (`+=`, :379)
    sh.sh_encode_backward(grad, v, B, 3, degree, j_r, gi_r)
    ps.sh_encode_backward(grad, v, B, 3, degree, j_p, gi_p)
    gi_o = O.sh_encode_backward(grad.cpu().numpy(), v.cpu().numpy(), degree, j_o)
    gs = max(1.0, float(np.abs(gi_o).max()))
    assert_close(gi_r, gi_o, rtol=1e-4, atol=1e-5 * gs)
    assert_close(gi_p, gi_r.cpu().numpy(), rtol=1e-4, atol=1e-5 * gs)


def test_config4_frame_on_the_reference_inference_kernels_equals_the_whole_frame_pass(ref, monkeypatch):
    """BASELINE configs[4] end to end: the 640 x 480 frame rendered by the product's whole-frame pass (one march, one
    compositing pass) against the reference's round schedule (nerf/renderer.py:330-380) running on the REFERENCE's
    near_far_from_aabb / march_rays / composite_rays / compact_rays kernels, same FFMLP networks evaluated in between.
    Equal up to the round schedule's own hand-over of t (see tests/test_gpu_baseline_configs.py: a pixel in 10^3, by a bf16
    activation flip at most)."""
    import enerf_amd.raymarching as rmod
    from enerf_amd import frame, scene
    from enerf_amd.network_ff import NeRFNetwork
    bound = 2
    torch.manual_seed(0)
    model = NeRFNetwork(encoding="hashgrid", bound=bound, cuda_ray=True).to(DEV).eval()
    model.encoder.embeddings.data.uniform_(-1, 1)
    g = torch.Generator(device=DEV).manual_seed(9)
    model.sigma_net.weights.data.copy_((torch.rand(model.sigma_net.weights.shape, generator=g, device=DEV) - 0.5) * 0.9)
    model.color_net.weights.data.copy_((torch.rand(model.color_net.weights.shape, generator=g, device=DEV) - 0.5) * 0.6)
    bits = O.packbits(synthetic_density_grid(bound, H).reshape(-1), 0.01)
    model.density_bitfield.copy_(torch.from_numpy(bits))
    model.density_scale = 40.0
    inds = torch.arange(scene.H * scene.W)
    o, d = scene.pixel_rays(scene.pose(3), inds, "cpu")
    ro, rd = o[0].contiguous().to(DEV), d[0].contiguous().to(DEV)
    assert ro.shape[0] == 307200
    with torch.no_grad():
        depth, image = frame.render_frame(model, ro, rd, 1)
        with monkeypatch.context() as mp:
            mp.setattr(rmod, "_backend", ref["rm"])              # the wrappers bind the reference's module, as its own do
            d_ref, im_ref = frame.render_rounds(model, ro, rd, 1)
        torch.cuda.synchronize()
    diff = (image - im_ref).abs()
    assert float(diff.max()) < 2e-3 and float((diff > 1e-4).float().mean()) < 2e-3, (float(diff.max()),
                                                                                    float((diff > 1e-4).float().mean()))
    dd = (depth - d_ref).abs()
    dd = dd[torch.isfinite(dd)]
    assert float(dd.max()) < 2e-4 and float((dd > 1e-5).float().mean()) < 2e-3
    assert float(image.std()) > 0.01 and float((image - 1).abs().max()) > 0.1       # a real picture, not the background


@pytest.mark.parametrize("reference_grid", [False, True])
def test_training_render_on_the_reference_kernels_equals_the_fused_product_render(ref, monkeypatch, reference_grid):
    """BASELINE configs[1] end to end (4096 rays, bound 3, jitter on): a training render + backward with the reference-shaped
    wrappers bound to the REFERENCE's raymarching and SH modules (op by op, autograd) against the product's fused route (one
    render node, closed backward, split-bf16 MLP kernels): sample counters bit-exact, image 1e-4, every parameter's gradient
    to the bars of the product's own route-equivalence tests.  reference_grid=True binds the reference's grid encoder as
    well (gridencoder.cu, oracle/build_ref.py): with the nn.Linear nets on torch's GEMMs -- what nerf/network.py runs on --
    every native instruction of that route is then the reference's or PyTorch's; False keeps the grid on this library (the
    round-3 form of the test)."""
    import enerf_amd.raymarching as rmod, enerf_amd.shencoder as smod, enerf_amd.gridencoder as gmod
    from enerf_amd import fused_mlp, fused_network, fused_render, density_update
    from enerf_amd.network import NeRFNetwork
    bound = 3
    bits = O.packbits(synthetic_density_grid(bound, H).reshape(-1), 0.01)
    o, d = camera_rays(4096, 56, bound)

    def run(on_reference):
        with monkeypatch.context() as mp:
            if on_reference:
                mp.setattr(rmod, "_backend", ref["rm"]); mp.setattr(smod, "_backend", ref["sh"])
                if reference_grid:
                    mp.setattr(gmod, "_backend", ref["ge"]); mp.setattr(gmod, "_layout_support", {})
                    mp.setattr(fused_mlp, "ENABLED", False)       # the nets: nn.Linear / ReLU on torch's GEMMs
                mp.setattr(fused_render, "ENABLED", False); mp.setattr(fused_network, "ENABLED", False)
                mp.setattr(density_update, "ENABLED", False)
            torch.manual_seed(0)
            model = NeRFNetwork(encoding="hashgrid", bound=bound, cuda_ray=True, out_dim_color=3)
            gg = torch.Generator().manual_seed(5)
            model.encoder.embeddings.data.copy_(torch.rand(model.encoder.embeddings.shape, generator=gg) * 2 - 1)
            model.density_bitfield.copy_(torch.from_numpy(bits))
            model.to(DEV).train()
            ro, rd = torch.from_numpy(o)[None].to(DEV), torch.from_numpy(d)[None].to(DEV)
            out = model.render(ro, rd, staged=False, bg_color=torch.full((3,), 0.3, device=DEV), perturb=True,
                               force_all_rays=True)
            (out["image"] ** 2).sum().backward()
            torch.cuda.synchronize()
            return (out["image"].detach().cpu(), model.step_counter[0].cpu().clone(),
                    {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()})

    im_r, c_r, g_r = run(True)
    im_p, c_p, g_p = run(False)
    assert torch.equal(c_r, c_p) and int(c_r[0]) > 100_000          # the same ~300 k samples, 4096 rays
    assert_close(im_p, im_r, rtol=1e-4, atol=2e-5)
    for n in g_r:
        scale = float(g_r[n].abs().max())
        assert scale > 0
        assert float((g_p[n] - g_r[n]).abs().max()) < 2e-3 * scale, (n, float((g_p[n] - g_r[n]).abs().max()) / scale)
