"""Pin the CPU oracle (oracle/enerf_oracle.c) to independent statements: canonical PCG32 stream, bit-loop morton,
numpy.packbits, scipy spherical harmonics, torch autograd, and the reference's own cumprod compositing
(tests/golden/ref_composite_vs_run.npz, minted from NeRFRenderer.run by oracle/make_golden.py)."""
import numpy as np
import torch

from oracle import oracle as O
from util import golden


def test_pcg32_canonical_stream():
    # the PCG reference demo: pcg32(seed=42, seq=54) -> 0xa15c02b7 0x7b47f409 0xba1d3330 ...
    u, _ = O.pcg32_stream(42, 54, 6)
    assert [hex(x) for x in u] == ["0xa15c02b7", "0x7b47f409", "0xba1d3330", "0x83d2f293", "0xbfa4784b", "0xcbed606e"]


def test_pcg32_first_floats():
    # SURVEY.md appendix A.10 (transcribed independently in Python during the survey)
    exp_u = [0x0f5deba9, 0xc9828f91, 0x7aa10266, 0x0f5deba9, 0xb2723db7, 0xf2393151]
    exp_f = [0.0600267649, 0.787148356, 0.479019284, 0.0600267649, 0.697055578, 0.946185112]
    for n in range(6):
        u, f = O.pcg32_stream(n, 1, 1)
        assert int(u[0]) == exp_u[n]
        assert abs(float(f[0]) - exp_f[n]) < 1e-9
    assert abs(float(O.pcg32_stream(5, 3, 1)[1][0]) - 0.592513323) < 1e-9
    # pure-python transcription of the generator for a few more seeds
    def py_pcg(seed, seq, n):
        M, mask = 0x5851f42d4c957f2d, (1 << 64) - 1
        state, inc = 0, ((seq << 1) | 1) & mask
        def nxt():
            nonlocal state
            old = state
            state = (old * M + inc) & mask
            xs = (((old >> 18) ^ old) >> 27) & 0xffffffff
            rot = old >> 59
            return ((xs >> rot) | (xs << ((-rot) & 31))) & 0xffffffff
        nxt(); state = (state + seed) & mask; nxt()
        return [nxt() for _ in range(n)]
    for seed, seq in ((4095, 1), (123456, 7), (2 ** 33 + 5, 1)):
        assert list(O.pcg32_stream(seed, seq, 5)[0]) == py_pcg(seed, seq, 5)


def test_morton_bijection_and_bitloop():
    ax = np.arange(128, dtype=np.int32)
    c = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    m = O.morton3D(c)
    assert m.min() == 0 and m.max() == 128 ** 3 - 1 and len(np.unique(m)) == 128 ** 3
    assert (O.morton3D_invert(m) == c).all()
    naive = np.zeros(len(c), np.int64)
    for b in range(10):
        naive |= ((c[:, 0] >> b) & 1).astype(np.int64) << (3 * b)
        naive |= ((c[:, 1] >> b) & 1).astype(np.int64) << (3 * b + 1)
        naive |= ((c[:, 2] >> b) & 1).astype(np.int64) << (3 * b + 2)
    assert (naive == m).all()


def test_packbits_vs_numpy():
    g = np.random.default_rng(0).random(8 * 4096).astype(np.float32)
    g[:8] = 0.5                      # strict '>' : equal is not occupied
    g[8:16] = -1.0                   # untrained cells
    assert (O.packbits(g, 0.5) == np.packbits(g > 0.5, bitorder="little")).all()


def _real_sh_scipy(deg, v):
    try:                                 # SciPy >= 1.15: sph_harm_y(l, m, polar, azimuth)
        from scipy.special import sph_harm_y

        def sph_harm(m, l, azimuth, polar):
            return sph_harm_y(l, m, polar, azimuth)
    except ImportError:
        from scipy.special import sph_harm
    x, y, z = v[:, 0].astype(np.float64), v[:, 1].astype(np.float64), v[:, 2].astype(np.float64)
    theta = np.arctan2(y, x)           # azimuth
    phi = np.arccos(np.clip(z, -1, 1))   # polar
    out = np.zeros((len(v), deg * deg))
    for l in range(deg):
        for m in range(-l, l + 1):
            Y = sph_harm(abs(m), l, theta, phi)
            out[:, l * l + l + m] = Y.real if m == 0 else (np.sqrt(2) * (Y.real if m > 0 else Y.imag))
    return out


def test_sh_vs_scipy_on_unit_sphere():
    v = np.random.default_rng(1).normal(size=(200, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    y, _ = O.sh_encode_forward(v, 8)
    np.testing.assert_allclose(y, _real_sh_scipy(8, v), atol=1e-5)  # v is unit only to fp32 round-off


def test_sh_off_sphere_polynomial_form_and_jacobian():
    # off the unit sphere the basis is the z-polynomial form: Y_2^0 = 0.9462 z^2 - 0.3154, Y_3^0 = 0.3732 z (5 z^2 - 3)
    v = (np.random.default_rng(2).normal(size=(50, 3)) * 0.7).astype(np.float32)
    y, j = O.sh_encode_forward(v, 4, True)
    x_, y_, z_ = v[:, 0].astype(np.float64), v[:, 1].astype(np.float64), v[:, 2].astype(np.float64)
    np.testing.assert_allclose(y[:, 0], 0.28209479177387814, atol=1e-7)
    np.testing.assert_allclose(y[:, 1], -0.48860251190291987 * y_, atol=1e-6)
    np.testing.assert_allclose(y[:, 3], -0.48860251190291987 * x_, atol=1e-6)
    np.testing.assert_allclose(y[:, 4], 1.0925484305920792 * x_ * y_, atol=1e-6)
    np.testing.assert_allclose(y[:, 6], 0.94617469575755997 * z_ * z_ - 0.31539156525251999, atol=1e-6)
    np.testing.assert_allclose(y[:, 8], 0.54627421529603959 * (x_ * x_ - y_ * y_), atol=1e-6)
    np.testing.assert_allclose(y[:, 12], 0.3731763325901154 * z_ * (5 * z_ * z_ - 3), atol=2e-6)
    np.testing.assert_allclose(y[:, 15], 0.59004358992664352 * x_ * (-x_ * x_ + 3 * y_ * y_), atol=2e-6)
    # jacobian vs central differences of the oracle itself in double-ish steps
    eps = 1e-3
    for d in range(3):
        vp, vm = v.copy(), v.copy()
        vp[:, d] += eps
        vm[:, d] -= eps
        fd = (O.sh_encode_forward(vp, 4)[0].astype(np.float64) - O.sh_encode_forward(vm, 4)[0]) / (2 * eps)
        np.testing.assert_allclose(j.reshape(50, 3, 16)[:, d], fd, atol=2e-3)


def _torch_grid_reference(x, emb, offsets, per_level_scale, H):
    """Independent pure-torch multires grid (fp64 interpolation weights) written from the paper + grid.py sizing."""
    B, D = x.shape
    L = len(offsets) - 1
    S = np.float32(np.log2(per_level_scale))
    outs = []
    for l in range(L):
        scale = np.float32(np.float32(np.exp2(np.float32(l) * S)) * np.float32(H) - np.float32(1.0))
        res = int(np.ceil(scale)) + 1
        size = int(offsets[l + 1] - offsets[l])
        pos = x.double() * float(scale) + 0.5
        pg = torch.floor(pos)
        fr = pos - pg
        pg = pg.long()
        acc = torch.zeros(B, emb.shape[1], dtype=torch.float64)
        for idx in range(8):
            w = torch.ones(B, dtype=torch.float64)
            c = []
            for d in range(3):
                if idx & (1 << d):
                    w = w * fr[:, d]
                    c.append(pg[:, d] + 1)
                else:
                    w = w * (1 - fr[:, d])
                    c.append(pg[:, d])
            stride, index, hashed = 1, torch.zeros(B, dtype=torch.long), False
            for d in range(3):
                if stride <= size:
                    index = index + c[d] * stride
                    stride *= (res + 1)
            if stride > size:
                primes = [1, 2654435761, 805459861]
                index = torch.zeros(B, dtype=torch.long)
                for d in range(3):
                    index = index ^ ((c[d] * primes[d]) & 0xffffffff)
            index = (index & 0xffffffff) % size
            acc = acc + w[:, None] * emb[offsets[l] + index].double()
        outs.append(acc)
    return torch.stack(outs, 0)   # [L,B,C]


def test_grid_oracle_vs_independent_torch_and_autograd():
    offsets, pls = O.grid_offsets(num_levels=8, base_resolution=4, log2_hashmap_size=10, desired_resolution=200)
    rng = np.random.default_rng(3)
    emb = torch.from_numpy(rng.uniform(-1, 1, (int(offsets[-1]), 2)).astype(np.float32)).requires_grad_(True)
    x = torch.from_numpy(rng.uniform(0, 1, (128, 3)).astype(np.float32))
    ref = _torch_grid_reference(x, emb, offsets, pls, 4)
    out, _ = O.grid_encode_forward(x.numpy(), emb.detach().numpy(), offsets, float(np.log2(pls)), 4)
    np.testing.assert_allclose(out, ref.detach().numpy(), atol=3e-5)   # fp32 pos = x*scale+0.5 vs fp64
    g = torch.from_numpy(rng.uniform(-1, 1, ref.shape))
    (ref * g).sum().backward()
    ge, _ = O.grid_encode_backward(g.float().numpy(), x.numpy(), emb.detach().numpy(), offsets, float(np.log2(pls)), 4)
    np.testing.assert_allclose(ge, emb.grad.numpy(), atol=1e-4)


def test_grid_oracle_dense_level_is_exact_trilinear_and_jacobian():
    # a table holding a linear field f(v) = a.v + b on a dense level must be reproduced exactly by interpolation
    offsets, pls = O.grid_offsets(num_levels=2, base_resolution=8, log2_hashmap_size=19, desired_resolution=16)
    S = float(np.log2(pls))
    emb = np.zeros((int(offsets[-1]), 2), np.float32)
    a = np.array([0.3, -0.2, 0.5])
    for l in range(2):
        scale, res = O.grid_level_params(l, np.float32(S), 8)
        n = res + 1
        ii = np.arange(n)
        gx, gy, gz = np.meshgrid(ii, ii, ii, indexing="ij")
        idx = (gx + gy * n + gz * n * n).reshape(-1)
        val = (a[0] * gx + a[1] * gy + a[2] * gz).reshape(-1)
        emb[offsets[l] + idx, 0] = val
        emb[offsets[l] + idx, 1] = 1.0
    x = np.random.default_rng(4).uniform(0, 1, (64, 3)).astype(np.float32)
    out, jac = O.grid_encode_forward(x, emb, offsets, S, 8, True)
    for l in range(2):
        scale, _ = O.grid_level_params(l, np.float32(S), 8)
        pos = x.astype(np.float64) * scale + 0.5
        np.testing.assert_allclose(out[l, :, 0], pos @ a, atol=2e-5)
        np.testing.assert_allclose(out[l, :, 1], 1.0, atol=1e-6)
        j = jac.reshape(64, 2, 3, 2)[:, l]
        np.testing.assert_allclose(j[:, :, 0], np.broadcast_to(a * scale, (64, 3)), atol=2e-4)
        np.testing.assert_allclose(j[:, :, 1], 0.0, atol=1e-5)


def test_composite_oracle_vs_reference_run_formula():
    """oracle composite_rays_train fwd+bwd == NeRFRenderer.run's cumprod compositing + torch autograd (reference)."""
    g = golden("ref_composite_vs_run")
    N, T = g["sigmas"].shape
    nears, fars = g["nears"].astype(np.float32), g["fars"].astype(np.float32)
    z = nears[:, None] + (fars - nears)[:, None] * np.linspace(0.0, 1.0, T, dtype=np.float32)[None]
    dl = np.concatenate([z[:, 1:] - z[:, :-1], ((fars - nears) / T)[:, None]], 1).astype(np.float32)
    deltas = np.stack([dl, dl], -1).reshape(-1, 2)
    rays = np.stack([np.arange(N), np.arange(N) * T, np.full(N, T)], -1).astype(np.int32)
    M = N * T + 1
    sig = np.concatenate([g["sigmas"].reshape(-1), [0]]).astype(np.float32)
    rgb = np.concatenate([g["rgbs"].reshape(-1, 3), np.zeros((1, 3))]).astype(np.float32)
    dlt = np.concatenate([deltas, np.zeros((1, 2))]).astype(np.float32)
    ws, depth, image = O.composite_rays_train_forward(sig, rgb, dlt, rays)
    np.testing.assert_allclose(image, g["image"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(ws, g["weights_sum"], rtol=1e-4, atol=2e-6)
    gs, gc = O.composite_rays_train_backward(np.zeros(N, np.float32), g["grad_image"], sig, rgb, dlt, rays, ws, image)
    np.testing.assert_allclose(gs[:-1].reshape(N, T), g["g_sig_img"], rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(gc[:-1].reshape(N, T, 3), g["g_rgb_img"], rtol=1e-4, atol=2e-6)
    gs2, _ = O.composite_rays_train_backward(g["grad_ws"], np.zeros((N, 3), np.float32), sig, rgb, dlt, rays, ws, image)
    np.testing.assert_allclose(gs2[:-1].reshape(N, T), g["g_sig_ws"], rtol=2e-3, atol=2e-5)


def test_near_far_vs_slab_reference():
    rng = np.random.default_rng(5)
    o = rng.uniform(-4, 4, (500, 3)).astype(np.float32)
    d = rng.normal(size=(500, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    bound = 2.0
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    n, f = O.near_far_from_aabb(o, d, aabb, 0.2)
    tmin = (-bound - o.astype(np.float64)) / d
    tmax = (bound - o.astype(np.float64)) / d
    near = np.minimum(tmin, tmax).max(1)
    far = np.maximum(tmin, tmax).min(1)
    hit = far >= near
    assert (n[~hit] == np.finfo(np.float32).max).all() and (f[~hit] == np.finfo(np.float32).max).all()
    np.testing.assert_allclose(n[hit], np.maximum(near[hit], 0.2), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(f[hit], far[hit], rtol=1e-5, atol=1e-6)


def test_ffmlp_oracle_vs_torch_linear_stack():
    rng = np.random.default_rng(6)
    for (i, o, h, k) in ((32, 16, 64, 2), (32, 16, 64, 3), (64, 16, 64, 2)):
        nW = h * (i + h * (k - 1) + o)
        W = torch.from_numpy(rng.uniform(-0.2, 0.2, nW).astype(np.float32)).requires_grad_(True)
        x = torch.from_numpy(rng.uniform(-1, 1, (40, i)).astype(np.float32)).requires_grad_(True)
        hcur = torch.relu(x @ W[: h * i].view(h, i).t())
        off = h * i
        for _ in range(k - 1):
            hcur = torch.relu(hcur @ W[off: off + h * h].view(h, h).t())
            off += h * h
        y = hcur @ W[off:].view(o, h).t()
        g = torch.from_numpy(rng.uniform(-1, 1, (40, o)).astype(np.float32))
        (y * g).sum().backward()
        out, fb = O.ffmlp_forward(x.detach().numpy(), W.detach().numpy(), i, o, h, k)
        np.testing.assert_allclose(out, y.detach().numpy(), rtol=1e-4, atol=1e-5)
        gi, gw, _ = O.ffmlp_backward(g.numpy(), x.detach().numpy(), W.detach().numpy(), fb, i, o, h, k)
        np.testing.assert_allclose(gi, x.grad.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(gw, W.grad.numpy(), rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------ reference-derived pins (minted in-container by
# oracle/make_golden.py from the reference's own statements of the math; VERDICT r1 "missing" #6)
def test_sh_oracle_vs_reference_literal_tables():
    """tests/golden/ref_sh_literals.npz = the 64 + 192 literal polynomials of shencoder.cu:51-121,131-351, parsed from
    the reference source at mint time and evaluated in float64.  The oracle (which generates the basis from
    recurrences) must reproduce every one of them, on and off the unit sphere, for every degree."""
    g = golden("ref_sh_literals")
    d, Y, J = g["d"], g["y"], g["dy_dx"]
    for deg in range(1, 9):
        C2 = deg * deg
        y, j = O.sh_encode_forward(d, deg, True)
        np.testing.assert_allclose(y, Y[:, :C2], rtol=2e-5, atol=3e-6, err_msg=f"degree {deg}")
        np.testing.assert_allclose(j.reshape(len(d), 3, C2), J[:, :, :C2], rtol=2e-5, atol=3e-5,
                                   err_msg=f"degree {deg} jacobian")
    # ordering / sign convention of the first band, spelled out (outputs[1] = -0.4886 y, [2] = +0.4886 z, [3] = -0.4886 x)
    y, _ = O.sh_encode_forward(np.array([[0.25, -0.5, 0.75]], np.float32), 2)
    np.testing.assert_allclose(y[0], [0.28209479, 0.48860251 * 0.5, 0.48860251 * 0.75, -0.48860251 * 0.25], rtol=1e-6)


def test_near_far_oracle_vs_reference_near_far_from_bound():
    """tests/golden/ref_near_far_from_bound.npz = nerf/renderer.py:48-72 near_far_from_bound(type='cube') run on the
    imported reference.  Same slab test as near_far_from_aabb (raymarching.cu:94-158) up to three documented deltas:
    `+1e-15` in the divisor (invisible in fp32 unless a direction component is ~0), a miss gives 1e9 instead of
    FLT_MAX, and min_near is hard-coded to 0.05."""
    g = golden("ref_near_far_from_bound")
    FLT_MAX = np.float32(3.4028234663852886e38)
    for bound in (1, 2, 3):
        o, d = g[f"o_b{bound}"], g[f"d_b{bound}"]
        near_ref, far_ref = g[f"near_b{bound}"], g[f"far_b{bound}"]
        aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
        near, far = O.near_far_from_aabb(o, d, aabb, 0.05)
        miss_ref = far_ref >= 1e9
        assert miss_ref.sum() >= 2 and (~miss_ref).sum() >= 100
        assert np.array_equal(near == FLT_MAX, miss_ref) and np.array_equal(far == FLT_MAX, miss_ref)
        hit = ~miss_ref
        np.testing.assert_allclose(near[hit], near_ref[hit], rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(far[hit], far_ref[hit], rtol=2e-6, atol=1e-6)
