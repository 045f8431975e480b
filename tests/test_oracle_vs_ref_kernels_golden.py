"""The C oracle (and the scalar second statement of the marcher) against outputs of the REFERENCE's own kernels: arrays
minted by oracle/mint_ref_gpu.py from oracle/_ref (the reference's raymarching.cu / shencoder.cu built for gfx950, see
oracle/build_ref.py) on an MI355X, stored as data under tests/golden/ref_kernels_gfx950.npz.  Bit for bit on everything
the marchers emit (per-ray counts, positions, step sizes, real deltas -- with dt_gamma != 0 and with the PCG32 jitter),
1e-5 on compositing and the SH basis.  CPU only."""
import math

import numpy as np
import pytest

from oracle import oracle as O
from util import golden, assert_close

H = 128


@pytest.fixture(scope="module")
def z():
    return golden("ref_kernels_gfx950")


def _ray_order(x, rays):
    rows = [np.arange(o, o + n) for _, o, n in rays if n > 0]
    return x[np.concatenate(rows)]


@pytest.mark.parametrize("bound", [2, 3])
def test_near_far_and_training_march_equal_the_reference_kernels(z, bound):
    k = f"b{bound}_"
    C = 1 + math.ceil(math.log2(bound))
    o, d, bits = z[k + "o"], z[k + "d"], z[k + "bits"]
    N = len(o)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.2)
    assert np.array_equal(nears, z[k + "nears"]) and np.array_equal(fars, z[k + "fars"])
    for tag, dt_gamma, perturb in (("plain", 0.0, 0), ("gamma", 1.0 / 256, 0), ("jitter", 0.0, 1)):
        x, dd, dl, rays, counter = O.march_rays_train(o, d, bits, bound, dt_gamma, 1024, C, H, N * 1024, nears, fars, perturb)
        assert np.array_equal(counter, z[k + tag + "_counter"]), tag
        assert np.array_equal(rays[:, 2], z[k + tag + "_counts"]), tag
        assert np.array_equal(_ray_order(x, rays), z[k + tag + "_xyzs"]), tag
        assert np.array_equal(_ray_order(dl, rays), z[k + tag + "_deltas"]), tag
        assert int(counter[0]) > 20 * N


def test_second_statement_of_the_marcher_equals_the_reference_kernels(z):
    """oracle/march_second.py (scalar float32 Python) on the first rays of the jittered bound-2 case"""
    from oracle import march_second as S
    bound, k = 2, "b2_"
    C = 2
    o, d, bits = z[k + "o"][:24], z[k + "d"][:24], z[k + "bits"]
    nears, fars = z[k + "nears"][:24], z[k + "fars"][:24]
    counts = z[k + "jitter_counts"][:24]
    tot = int(counts.sum())
    x, dd, dl, rays, counter = S.march_rays_train(o, d, bits, bound, 0.0, 1024, C, H, 24 * 1024, nears, fars, 1)
    assert np.array_equal(np.asarray(rays)[:, 2], counts)
    assert np.array_equal(np.asarray(x, np.float32)[:tot], z[k + "jitter_xyzs"][:tot])
    assert np.array_equal(np.asarray(dl, np.float32)[:tot], z[k + "jitter_deltas"][:tot])


@pytest.mark.parametrize("bound", [2, 3])
def test_compositing_equals_the_reference_kernels(z, bound):
    k = f"b{bound}_"
    counts = z[k + "plain_counts"]
    N = len(counts)
    tot = int(counts.sum())
    sig, rgb = z[k + "comp_sig"], z[k + "comp_rgb"]
    m = len(sig)
    dl = np.zeros((m, 2), np.float32); dl[:tot] = z[k + "plain_deltas"]
    off = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int32)
    rays = np.stack([np.arange(N, dtype=np.int32), off, counts], 1).astype(np.int32)
    ws, dp, im = O.composite_rays_train_forward(sig, rgb, dl, rays)
    assert_close(ws, z[k + "comp_ws"], rtol=1e-5, atol=1e-6)
    assert_close(dp, z[k + "comp_depth"], rtol=1e-5, atol=1e-5)
    assert_close(im, z[k + "comp_image"], rtol=1e-5, atol=1e-6)
    gs, gc = O.composite_rays_train_backward(z[k + "comp_g_ws"], z[k + "comp_g_im"], sig, rgb, dl, rays, ws, im)
    assert_close(gs, z[k + "comp_gs"], rtol=1e-4, atol=2e-6)
    assert_close(gc, z[k + "comp_gc"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("bound", [2, 3])
def test_inference_round_equals_the_reference_kernels(z, bound):
    k = f"b{bound}_"
    C = 1 + math.ceil(math.log2(bound))
    o, d, bits, nears, fars = z[k + "o"], z[k + "d"], z[k + "bits"], z[k + "nears"], z[k + "fars"]
    N = len(o)
    alive = np.arange(N, dtype=np.int32)
    for tag, perturb in (("inf", 0), ("infj", 3)):
        x, dd, dl = O.march_rays(N, 8, alive, nears.copy(), o, d, bound, 0.0, 1024, C, H, bits, nears, fars, N * 8, perturb)
        assert np.array_equal(x, z[k + tag + "_xyzs"]) and np.array_equal(dl, z[k + tag + "_deltas"]), tag
    x, dd, dl = O.march_rays(N, 8, alive, nears.copy(), o, d, bound, 0.0, 1024, C, H, bits, nears, fars, N * 8, 0)
    ws = np.zeros(N, np.float32); dp = np.zeros(N, np.float32); im = np.zeros((N, 3), np.float32)
    rt = nears.copy()
    O.composite_rays(N, 8, alive, rt, z[k + "inf_sig"], z[k + "inf_rgb"], dl, ws, dp, im)
    ref_rt = z[k + "inf_rt"]
    assert np.array_equal(rt < 0, ref_rt < 0)
    assert_close(rt, ref_rt, rtol=1e-6, atol=1e-6)
    assert_close(ws, z[k + "inf_ws"], rtol=1e-5, atol=1e-6); assert_close(im, z[k + "inf_image"], rtol=1e-5, atol=1e-6)
    assert_close(dp, z[k + "inf_depth"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("degree", range(1, 9))
def test_sh_basis_and_jacobian_equal_the_reference_kernels(z, degree):
    v = z["sh_dirs"]
    y, j = O.sh_encode_forward(v, degree, True)
    ys, js = max(1.0, float(np.abs(y).max())), max(1.0, float(np.abs(j).max()))
    assert_close(y, z[f"sh{degree}_y"], rtol=2e-6, atol=2e-7 * ys)
    assert_close(j, z[f"sh{degree}_dy_dx"], rtol=2e-6, atol=2e-7 * js)
    gi = O.sh_encode_backward(z[f"sh{degree}_grad"], v, degree, j)
    assert_close(gi, z[f"sh{degree}_gi"], rtol=1e-5, atol=1e-6 * max(1.0, float(np.abs(gi).max())))


# ------------------------------------------------------------------ the reference's grid encoder (gridencoder.cu on gfx950)
GRID_CASES = ["d3c2hash", "d3c1hash", "d3c4hash", "d3c8tiled", "d2c2hash", "d2c4tiled", "baseline_bound2", "baseline_bound3"]


@pytest.mark.parametrize("tag", GRID_CASES)
def test_grid_encoder_equals_the_reference_kernel(tag):
    """tests/golden/ref_gridencoder_gfx950.npz (oracle/mint_ref_grid_gpu.py): what the reference's kernel_grid /
    kernel_grid_backward / kernel_input_backward computed on an MI355X.  Forward and Jacobian: bit for bit, level by
    level.  The reference evaluates exp2f(level * S) with the device's libm (gridencoder.cu:124); where ROCm's value is
    not glibc's the level must be bit-equal under the neighbouring float (O.grid_exp2f_nudged) -- the constant is the
    only difference -- and at most two levels of a table may need that."""
    z = golden("ref_gridencoder_gfx950")
    x, emb, offsets = z[tag + "_x"], z[tag + "_emb"], z[tag + "_offsets"]
    S, Hb, gridtype = float(z[tag + "_S"]), int(z[tag + "_H"]), int(z[tag + "_gridtype"])
    y, jac = z[tag + "_y"], z[tag + "_dy_dx"]
    L, B, C = y.shape
    D = x.shape[1]
    out, dj = O.grid_encode_forward(x, emb, offsets, S, Hb, True, gridtype)
    dj = dj.reshape(B, L, D * C)
    jac = jac.reshape(B, L, D * C)
    nudged = {}
    for l in range(L):
        if np.array_equal(out[l], y[l]) and np.array_equal(dj[:, l], jac[:, l]):
            continue
        for ulps in (1, -1):
            with O.grid_exp2f_nudged(l, ulps):
                o2, j2 = O.grid_encode_forward(x, emb, offsets, S, Hb, True, gridtype)
            if np.array_equal(o2[l], y[l]) and np.array_equal(j2.reshape(B, L, D * C)[:, l], jac[:, l]):
                nudged[l] = ulps
                break
        else:
            raise AssertionError(f"{tag} level {l}: not the reference kernel's output under any exp2f within one ulp")
    assert len(nudged) <= 2, nudged
    assert np.all(y[:, 2] == 0) and np.all(y[:, 3] == 0)            # the out-of-range rows
    # backward on the reference platform's constants (atomics on the GPU side: tolerance)
    import contextlib
    with contextlib.ExitStack() as st:
        for l, ulps in nudged.items():
            st.enter_context(O.grid_exp2f_nudged(l, ulps))
        ge, gi = O.grid_encode_backward(z[tag + "_g"], x, emb, offsets, S, Hb, z[tag + "_dy_dx"], gridtype)
    scale = np.abs(ge).max()
    np.testing.assert_allclose(ge, z[tag + "_grad_emb"], rtol=1e-4, atol=4e-6 * scale)
    np.testing.assert_allclose(gi, z[tag + "_grad_x"], rtol=1e-5, atol=2e-6 * np.abs(gi).max())


def test_grid_encoder_platform_constant_is_named():
    """Which levels of BASELINE's tables the device's exp2f moved: recorded, so that a change of ROCm shows up here."""
    z = golden("ref_gridencoder_gfx950")
    moved = {}
    for tag in ("baseline_bound2", "baseline_bound3"):
        x, emb, offsets = z[tag + "_x"], z[tag + "_emb"], z[tag + "_offsets"]
        out, _ = O.grid_encode_forward(x, emb, offsets, float(z[tag + "_S"]), 16, False, 0)
        moved[tag] = [l for l in range(16) if not np.array_equal(out[l], z[tag + "_y"][l])]
    assert moved == {"baseline_bound2": [11], "baseline_bound3": []}, moved
