"""Parity of the HIP path (through the C ABI: enerf_amd/backends -> libenerf_hip.so) against the CPU oracle on the
same seeded inputs.  Integer / index outputs and marched sample positions must be bit-exact; floating-point
compositing / encodings within the stated tolerance (north_star: 1e-4 rel fp32)."""
import math

import numpy as np
import pytest
import torch

from oracle import oracle as O
from util import synthetic_density_grid, camera_rays, assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"
H = 128


def cu(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


@pytest.fixture(scope="module")
def rm():
    from enerf_amd.backends import _raymarching
    return _raymarching


@pytest.fixture(scope="module")
def scenes():
    out = {}
    for bound in (1, 2, 3):
        grid = synthetic_density_grid(bound, H)
        bits = O.packbits(grid.reshape(-1), 0.01)
        out[bound] = (grid, bits, 1 + math.ceil(math.log2(bound)))
    return out


def _rays(n, seed, bound):
    o, d = camera_rays(n, seed, bound)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    return o, d, aabb


# ------------------------------------------------------------------ small integer kernels: bit-exact
def test_near_far_bit_exact(rm):
    o, d, aabb = _rays(5000, 1, 2)
    d[10] = [1, 0, 0]; d[11] = [0, -1, 0]; d[12] = [0, 0, 1]          # axis-parallel: 1/0 = inf paths
    o[13] = [5, 5, 5]                                                     # outside, pointing away
    n_ref, f_ref = O.near_far_from_aabb(o, d, aabb, 0.2)
    nears = torch.empty(len(o), device=DEV); fars = torch.empty(len(o), device=DEV)
    rm.near_far_from_aabb(cu(o), cu(d), cu(aabb), len(o), 0.2, nears, fars)
    assert np.array_equal(nears.cpu().numpy(), n_ref) and np.array_equal(fars.cpu().numpy(), f_ref)


def test_near_far_vs_reference_near_far_from_bound(rm):
    """HIP near_far_from_aabb against the imported reference's own near_far_from_bound (nerf/renderer.py:48-72; fixture
    minted by oracle/make_golden.py; deltas: +1e-15 divisor, miss = 1e9 vs FLT_MAX, min_near 0.05)."""
    from util import golden
    g = golden("ref_near_far_from_bound")
    FLT_MAX = np.float32(3.4028234663852886e38)
    for bound in (1, 2, 3):
        o, d = g[f"o_b{bound}"], g[f"d_b{bound}"]
        aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
        nears = torch.empty(len(o), device=DEV); fars = torch.empty(len(o), device=DEV)
        rm.near_far_from_aabb(cu(o), cu(d), cu(aabb), len(o), 0.05, nears, fars)
        near, far = nears.cpu().numpy(), fars.cpu().numpy()
        miss = g[f"far_b{bound}"] >= 1e9
        assert np.array_equal(near == FLT_MAX, miss) and np.array_equal(far == FLT_MAX, miss)
        np.testing.assert_allclose(near[~miss], g[f"near_b{bound}"][~miss], rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(far[~miss], g[f"far_b{bound}"][~miss], rtol=2e-6, atol=1e-6)


def test_morton_and_packbits_bit_exact(rm, scenes):
    rng = np.random.default_rng(2)
    c = rng.integers(0, 128, (100003, 3)).astype(np.int32)
    idx = torch.empty(len(c), dtype=torch.int32, device=DEV)
    rm.morton3D(cu(c), len(c), idx)
    assert np.array_equal(idx.cpu().numpy(), O.morton3D(c))
    back = torch.empty(len(c), 3, dtype=torch.int32, device=DEV)
    rm.morton3D_invert(idx, len(c), back)
    assert np.array_equal(back.cpu().numpy(), c)
    grid, bits, C = scenes[3]                                            # full size: 3 x 128^3 cells
    g = grid.copy(); g[0, :8] = 0.01; g[1, 8:16] = -1
    out = torch.empty(g.size // 8, dtype=torch.uint8, device=DEV)
    rm.packbits(cu(g), g.size // 8, 0.01, out)
    assert np.array_equal(out.cpu().numpy(), O.packbits(g.reshape(-1), 0.01))
    assert np.array_equal(out.cpu().numpy(), np.packbits(g.reshape(-1) > 0.01, bitorder="little"))


def test_polar_from_ray(rm):
    o, d, _ = _rays(1000, 3, 1)
    ref = O.polar_from_ray(o * 0.3, d, 4.0)
    out = torch.empty(1000, 2, device=DEV)
    rm.polar_from_ray(cu(o * 0.3), cu(d), 4.0, 1000, out)
    assert_close(out, ref, rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------ march_rays_train: rays/counter/samples bit-exact
@pytest.fixture(params=["wave-per-ray", "thread-per-ray"])
def march_route(request):
    """Both fixed-step training marchers on every march_rays_train test: the wave-per-ray lattice marcher with its chunk
    log (what small batches take) and the one-thread-per-ray walk with its run log (what batches of 65 536+ rays take),
    forced through enerf_debug_march_thread_min_rays whatever the test's ray count."""
    from enerf_amd import _lib
    prev = _lib.lib().enerf_debug_march_thread_min_rays(1 if request.param == "thread-per-ray" else 0x7fffffff)
    yield request.param
    _lib.lib().enerf_debug_march_thread_min_rays(prev)


def _gpu_march_train(rm, o, d, bits, bound, dt_gamma, C, M, nears, fars, perturb, max_steps=1024):
    N = len(o)
    xyzs = torch.zeros(M, 3, device=DEV); dirs = torch.zeros(M, 3, device=DEV); deltas = torch.zeros(M, 2, device=DEV)
    rays = torch.empty(N, 3, dtype=torch.int32, device=DEV)
    counter = torch.zeros(2, dtype=torch.int32, device=DEV)
    rm.march_rays_train(cu(o), cu(d), cu(bits), bound, dt_gamma, max_steps, N, C, H, M, cu(nears), cu(fars), xyzs, dirs,
                        deltas, rays, counter, perturb)
    torch.cuda.synchronize()
    return [x.cpu().numpy() for x in (xyzs, dirs, deltas, rays, counter)]


@pytest.mark.parametrize("bound,perturb,dt_gamma,N", [(1, 0, 0.0, 300), (2, 1, 0.0, 1000), (3, 1, 0.0, 4096),
                                                      (3, 0, 1.0 / 128, 500), (2, 1, 1.0 / 256, 257)])
def test_march_rays_train_bit_exact(rm, scenes, bound, perturb, dt_gamma, N, march_route):
    grid, bits, C = scenes[bound]
    o, d, aabb = _rays(N, 10 + bound, bound)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.2)
    M = N * 1024
    ref = O.march_rays_train(o, d, bits, bound, dt_gamma, 1024, C, H, M, nears, fars, perturb)
    got = _gpu_march_train(rm, o, d, bits, bound, dt_gamma, C, M, nears, fars, perturb)
    assert np.array_equal(got[4], ref[4]), (got[4], ref[4])             # counter
    assert np.array_equal(got[3], ref[3])                               # rays (index, offset, num_steps)
    tot = int(ref[4][0])
    assert tot > 1000
    for a, b, name in zip(got[:3], ref[:3], ("xyzs", "dirs", "deltas")):
        assert np.array_equal(a[:tot], b[:tot]), name
        assert not a[tot:].any()


def test_march_rays_train_overflow_drop_rule(rm, scenes, march_route):
    bound = 2
    grid, bits, C = scenes[bound]
    o, d, aabb = _rays(512, 21, bound)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.2)
    tot = int(O.march_rays_train(o, d, bits, bound, 0.0, 1024, C, H, 512 * 1024, nears, fars, 1)[4][0])
    M = tot // 3
    M += 128 - M % 128
    ref = O.march_rays_train(o, d, bits, bound, 0.0, 1024, C, H, M, nears, fars, 1)
    got = _gpu_march_train(rm, o, d, bits, bound, 0.0, C, M, nears, fars, 1)
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)
    assert got[4][0] == tot


@pytest.mark.parametrize("overflow", [False, True])
@pytest.mark.parametrize("dt_gamma", [0.0, 1.0 / 128])
def test_march_rays_train_ex_zero_fills_unwritten_rows(rm, scenes, overflow, dt_gamma, march_route):
    """march_rays_train_ex(zero_unwritten=1) on NaN-filled buffers == march_rays_train on zero-filled ones, bit for bit:
    the budget tail and a dropped ray's clipped reservation are the rows no ray writes."""
    bound = 2
    grid, bits, C = scenes[bound]
    o, d, aabb = _rays(700, 23, bound)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.2)
    tot = int(O.march_rays_train(o, d, bits, bound, dt_gamma, 1024, C, H, 700 * 1024, nears, fars, 1)[4][0])
    M = tot // 2 if overflow else tot + 3000
    M += 128 - M % 128
    ref = _gpu_march_train(rm, o, d, bits, bound, dt_gamma, C, M, nears, fars, 1)
    N = len(o)
    xyzs = torch.full((M, 3), float("nan"), device=DEV); dirs = torch.full((M, 3), float("nan"), device=DEV)
    deltas = torch.full((M, 2), float("nan"), device=DEV)
    rays = torch.empty(N, 3, dtype=torch.int32, device=DEV)
    counter = torch.zeros(2, dtype=torch.int32, device=DEV)
    rm.march_rays_train_ex(cu(o), cu(d), cu(bits), bound, dt_gamma, 1024, N, C, H, M, cu(nears), cu(fars), xyzs, dirs,
                           deltas, rays, counter, 1, True)
    got = [x.cpu().numpy() for x in (xyzs, dirs, deltas, rays, counter)]
    for a, b, name in zip(got, ref, ("xyzs", "dirs", "deltas", "rays", "counter")):
        assert np.array_equal(a, b), name
    assert (got[4][0] > M) == overflow


@pytest.mark.parametrize("overflow", [False, True])
def test_march_rays_train_background_mode_is_bit_identical(rm, scenes, overflow, march_route):
    """Flag bit 1 of march_rays_train_ex (batch prepared ahead on a side stream: one marching wavefront per SIMD,
    rays in turn) == the oracle, bit for bit, with more rays than the launch has wavefronts."""
    bound = 2
    grid, bits, C = scenes[bound]
    o, d, aabb = _rays(3000, 29, bound)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.2)
    N = len(o)
    for perturb in (0, 1):
        tot = int(O.march_rays_train(o, d, bits, bound, 0.0, 1024, C, H, N * 1024, nears, fars, perturb)[4][0])
        M = tot // 2 if overflow else tot + 1000
        M += 128 - M % 128
        ref = O.march_rays_train(o, d, bits, bound, 0.0, 1024, C, H, M, nears, fars, perturb)
        xyzs = torch.full((M, 3), float("nan"), device=DEV); dirs = torch.full((M, 3), float("nan"), device=DEV)
        deltas = torch.full((M, 2), float("nan"), device=DEV)
        rays = torch.empty(N, 3, dtype=torch.int32, device=DEV)
        counter = torch.zeros(2, dtype=torch.int32, device=DEV)
        rm.march_rays_train_ex(cu(o), cu(d), cu(bits), bound, 0.0, 1024, N, C, H, M, cu(nears), cu(fars), xyzs, dirs,
                               deltas, rays, counter, perturb, 3)
        got = [x.cpu().numpy() for x in (xyzs, dirs, deltas, rays, counter)]
        for a, b, name in zip(got, ref, ("xyzs", "dirs", "deltas", "rays", "counter")):
            assert np.array_equal(a, b), (name, perturb)


@pytest.mark.parametrize("dt_gamma,flags", [(0.0, 0), (0.0, 3), (1.0 / 128, 0)])
def test_march_with_near_far_inside_the_count_pass(rm, scenes, dt_gamma, flags, march_route):
    """enerf_march_fuse_near_far: the armed march computes near / far in its count pass (or, on the routes that do not
    take the wave-per-ray marcher, runs the near_far kernel itself) and writes them into the arrays it is given --
    nears / fars == near_far_from_aabb, rays / counter / samples == the oracle's, bit for bit; the arming lasts one call."""
    bound = 3
    grid, bits, C = scenes[bound]
    o, d, aabb = _rays(2500, 41, bound)
    d[3] = [0, 0, 1]; o[4] = [7, 7, 7]                                  # axis-parallel; a ray that misses the box
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.2)
    N = len(o)
    M = int(O.march_rays_train(o, d, bits, bound, dt_gamma, 1024, C, H, N * 1024, nears, fars, 1)[4][0]) + 500
    M += 128 - M % 128
    ref = O.march_rays_train(o, d, bits, bound, dt_gamma, 1024, C, H, M, nears, fars, 1)
    g_n = torch.full((N,), float("nan"), device=DEV); g_f = torch.full((N,), float("nan"), device=DEV)
    xyzs = torch.full((M, 3), float("nan"), device=DEV); dirs = torch.full((M, 3), float("nan"), device=DEV)
    deltas = torch.full((M, 2), float("nan"), device=DEV)
    rays = torch.empty(N, 3, dtype=torch.int32, device=DEV)
    counter = torch.zeros(2, dtype=torch.int32, device=DEV)
    aabb_d = cu(aabb)                                                   # (the pointer is read by the march, not by the arming)
    rm.march_fuse_near_far(aabb_d, 0.2)
    rm.march_rays_train_ex(cu(o), cu(d), cu(bits), bound, dt_gamma, 1024, N, C, H, M, g_n, g_f, xyzs, dirs, deltas, rays,
                           counter, 1, 1 | flags)
    assert np.array_equal(g_n.cpu().numpy(), nears) and np.array_equal(g_f.cpu().numpy(), fars)
    got = [x.cpu().numpy() for x in (xyzs, dirs, deltas, rays, counter)]
    for a, b, name in zip(got, ref, ("xyzs", "dirs", "deltas", "rays", "counter")):
        assert np.array_equal(a, b), name
    # not armed any more: the next call reads the arrays it is given
    counter.zero_()
    bad = torch.full((N,), 3.4028234663852886e38, device=DEV)
    rm.march_rays_train_ex(cu(o), cu(d), cu(bits), bound, dt_gamma, 1024, N, C, H, M, bad, bad.clone(), xyzs, dirs, deltas,
                           rays, counter, 1, 1 | flags)
    assert int(counter[0]) == 0


def test_march_count_mirrored_into_pinned_host_memory(rm, scenes):
    """enerf_march_mirror_count: the armed count pass writes (samples reserved, rays marched) into pinned host memory too --
    a host watching the two words it pre-set to -1 reads the device counter's values; the arming lasts one call."""
    bound = 2
    grid, bits, C = scenes[bound]
    o, d, aabb = _rays(1500, 7, bound)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.2)
    N = len(o)
    host = torch.empty(2, dtype=torch.int32, pin_memory=True)
    rays = torch.empty(N, 3, dtype=torch.int32, device=DEV)
    counter = torch.zeros(2, dtype=torch.int32, device=DEV)
    host.fill_(-1)
    rm.march_mirror_count(host)
    rm.march_rays_train_count(cu(o), cu(d), cu(bits), bound, 0.0, 1024, N, C, H, cu(nears), cu(fars), rays, counter, 1, 0)
    torch.cuda.synchronize()
    assert host.tolist() == counter.cpu().tolist() and host[1].item() == N and host[0].item() > 0
    host.fill_(-1)
    rm.march_rays_train_count(cu(o), cu(d), cu(bits), bound, 0.0, 1024, N, C, H, cu(nears), cu(fars), rays, counter, 1, 0)
    torch.cuda.synchronize()
    assert host.tolist() == [-1, -1]


@pytest.mark.parametrize("dt_gamma", [0.0, 1.0 / 128])
def test_march_rays_train_count_then_write_equals_worst_case_buffers_cropped(rm, scenes, dt_gamma, march_route):
    """While no sample budget exists the reference allocates N * max_steps zero rows, marches, reads the count back and
    crops to the count rounded up past the next multiple of 128 (raymarching.py:195-228).  march_rays_train_count +
    march_rays_train_write into NaN-filled buffers of exactly that cropped size: same rows, rays and counter as the
    oracle run with the worst-case buffers, bit for bit -- including the corner where every ray takes max_steps samples,
    so that the total equals N * max_steps and the reference's `>=` rule drops the last ray."""
    bound = 2
    grid, bits, C = scenes[bound]
    cases = [(bits, 1024, 2500, 1), (bits, 1024, 2500, 0)]
    if dt_gamma == 0.0:
        sat = np.full(C * H ** 3 // 8, 0xff, np.uint8)
        cases.append((sat, 8, 64, 0))                                  # every ray: 8 of 8 steps -> total == N * max_steps
    for grid_bits, max_steps, N, perturb in cases:
        o, d, aabb = _rays(N, 37, bound)
        if max_steps == 8:
            o[:] = np.array([0.1, 0.2, -0.3], np.float32)              # inside the cube, far from its faces
        nears, fars = O.near_far_from_aabb(o, d, aabb, 0.2)
        worst = N * max_steps
        ref = O.march_rays_train(o, d, grid_bits, bound, dt_gamma, max_steps, C, H, worst, nears, fars, perturb)
        tot = int(ref[4][0])
        m = min(tot + 128 - tot % 128, worst)
        rays = torch.empty(N, 3, dtype=torch.int32, device=DEV)
        counter = torch.zeros(2, dtype=torch.int32, device=DEV)
        args = (cu(o), cu(d), cu(grid_bits), bound, dt_gamma, max_steps, N, C, H)
        nf = (cu(nears), cu(fars))
        rm.march_rays_train_count(*args, *nf, rays, counter, perturb, 0)
        assert np.array_equal(counter.cpu().numpy(), ref[4]) and np.array_equal(rays.cpu().numpy(), ref[3])
        xyzs = torch.full((m, 3), float("nan"), device=DEV); dirs = torch.full((m, 3), float("nan"), device=DEV)
        deltas = torch.full((m, 2), float("nan"), device=DEV)
        rm.march_rays_train_write(*args, m, *nf, xyzs, dirs, deltas, rays, counter, perturb, 1)
        for a, b, name in zip((xyzs, dirs, deltas), ref[:3], ("xyzs", "dirs", "deltas")):
            assert np.array_equal(a.cpu().numpy(), b[:m]), (name, max_steps, perturb)
        if max_steps == 8:
            assert tot == worst and m == worst and not ref[0][(N - 1) * 8:].any()      # the last ray was dropped


def test_march_count_occupied_box_test_changes_nothing(rm, scenes, march_route):
    """The count pass first tests each ray against the bounding box of the occupied cells and marches only to the box's
    far side (raymarching.hip: clip_to_occupied).  Same rays / counter / samples with the test switched off, and both
    equal the oracle, on a grid whose occupied region is a small off-centre block (most rays miss it, some graze it,
    some are axis-parallel) and on an empty grid."""
    bound = 3
    C = 1 + math.ceil(math.log2(bound))
    grid = np.zeros((C, H ** 3), np.float32)
    ax = np.arange(H, dtype=np.int32)
    cc = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    blk = (np.abs(cc[:, 0] - 80) < 9) & (np.abs(cc[:, 1] - 60) < 6) & (np.abs(cc[:, 2] - 70) < 12)
    idx = O.morton3D(cc[blk]).astype(np.int64)
    grid[0, idx] = 1.0
    grid[1, idx[::3]] = 1.0
    for g in (grid, np.zeros_like(grid)):
        bits = O.packbits(g.reshape(-1), 0.01)
        o, d, aabb = _rays(3000, 91, bound)
        d[5] = [1, 0, 0]; o[5] = [-2.5, -0.05, 0.1]                    # axis-parallel, through the block
        d[6] = [0, 0, 1]; o[6] = [2.0, 2.0, -2.9]                      # axis-parallel, misses it
        nears, fars = O.near_far_from_aabb(o, d, aabb, 0.2)
        M = len(o) * 1024
        ref = O.march_rays_train(o, d, bits, bound, 0.0, 1024, C, H, M, nears, fars, 1)
        outs = []
        N = len(o)
        gbits = cu(bits)
        rm.occupied_box_update(gbits, C, H, bound)
        for flags in (4 | 1, 1):                                        # with and without the box test
            xyzs = torch.full((M, 3), float("nan"), device=DEV); dirs = torch.full((M, 3), float("nan"), device=DEV)
            deltas = torch.full((M, 2), float("nan"), device=DEV)
            rays = torch.empty(N, 3, dtype=torch.int32, device=DEV)
            counter = torch.zeros(2, dtype=torch.int32, device=DEV)
            rm.march_rays_train_ex(cu(o), cu(d), gbits, bound, 0.0, 1024, N, C, H, M, cu(nears), cu(fars), xyzs, dirs,
                                   deltas, rays, counter, 1, flags)
            outs.append([x.cpu().numpy() for x in (xyzs, dirs, deltas, rays, counter)])
        # a box computed for another bitfield pointer is ignored, not trusted
        other = gbits.clone()
        rays2 = torch.empty(N, 3, dtype=torch.int32, device=DEV); counter2 = torch.zeros(2, dtype=torch.int32, device=DEV)
        rm.march_rays_train_count(cu(o), cu(d), other, bound, 0.0, 1024, N, C, H, cu(nears), cu(fars), rays2, counter2, 1, 4)
        assert np.array_equal(rays2.cpu().numpy(), ref[3])
        for got in outs:
            for a, b, name in zip(got, ref, ("xyzs", "dirs", "deltas", "rays", "counter")):
                assert np.array_equal(a, b), name
        if g is grid:
            hit = (ref[3][:, 2] > 0).mean()
            assert 0.02 < hit < 0.6 and ref[3][5, 2] > 0 and ref[3][6, 2] == 0


def test_march_rays_train_saturated_grid_and_chunk_log_overflow(rm, march_route):
    """Every cell occupied: rays emit a sample at every lattice point until max_steps (1024) -- the maximum the path
    can produce per ray, and more emitting 64-point chunks than the count pass's per-ray log holds for the longest
    rays (the write pass then re-marches those rays).  Bit-exact against the oracle, perturbed and not."""
    bound = 2
    C = 1 + math.ceil(math.log2(bound))
    bits = np.full(C * H ** 3 // 8, 0xff, np.uint8)
    o, d, aabb = _rays(96, 33, bound)
    o[:8] = np.array([-1.9, -1.9, -1.9], np.float32) + np.linspace(0, 0.05, 8, dtype=np.float32)[:, None]
    d[:8] = np.array([1, 1, 1], np.float32) / np.sqrt(3, dtype=np.float32)          # along the diagonal: > 1024 points
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.2)
    for perturb in (0, 1):
        M = 96 * 1024
        ref = O.march_rays_train(o, d, bits, bound, 0.0, 1024, C, H, M, nears, fars, perturb)
        got = _gpu_march_train(rm, o, d, bits, bound, 0.0, C, M, nears, fars, perturb)
        assert ref[3][:8, 2].max() == 1024                     # saturated at max_steps
        for a, b, name in zip(got, ref, ("xyzs", "dirs", "deltas", "rays", "counter")):
            assert np.array_equal(a, b), name
    # max_steps = 8192 with every other cell occupied (even x): the diagonal rays emit ~3600 samples from ~110
    # chunks, more than the 64 the log holds -> re-march fallback of the write pass
    bits = np.full(C * H ** 3 // 8, 0x55, np.uint8)
    M = 96 * 8192
    ref = O.march_rays_train(o, d, bits, bound, 0.0, 8192, C, H, M, nears, fars, 1)
    got = _gpu_march_train(rm, o, d, bits, bound, 0.0, C, M, nears, fars, 1, max_steps=8192)
    assert ref[3][:8, 2].max() > 3000
    for a, b, name in zip(got, ref, ("xyzs", "dirs", "deltas", "rays", "counter")):
        assert np.array_equal(a, b), name


def test_empty_inputs_are_noops():
    """N == 0 / B == 0 / n_alive == 0 through every entry point: status 0, nothing read or written."""
    from enerf_amd import _lib
    lib = _lib.lib()
    s = _lib.stream_handle()
    z = None
    assert lib.enerf_near_far_from_aabb(z, z, z, 0, 0.2, z, z, s) == 0
    assert lib.enerf_polar_from_ray(z, z, 1.0, 0, z, s) == 0
    assert lib.enerf_morton3D(z, 0, z, s) == 0
    assert lib.enerf_morton3D_invert(z, 0, z, s) == 0
    assert lib.enerf_packbits(z, 0, 0.01, z, s) == 0
    assert lib.enerf_march_rays_train(z, z, z, 1.0, 0.0, 1024, 0, 1, 128, 0, z, z, z, z, z, z, z, 0, s) == 0
    assert lib.enerf_composite_rays_train_forward(z, z, z, z, 0, 0, z, z, z, s) == 0
    assert lib.enerf_composite_rays_train_backward(z, z, z, z, z, z, z, z, 0, 0, z, z, s) == 0
    assert lib.enerf_march_rays(0, 8, z, z, z, z, 1.0, 0.0, 1024, 1, 128, z, z, z, z, z, z, 0, s) == 0
    assert lib.enerf_composite_rays(0, 8, z, z, z, z, z, z, z, z, s) == 0
    assert lib.enerf_compact_rays(0, z, z, z, z, z, s) == 0
    assert lib.enerf_grid_encode_forward(z, z, z, z, 0, 3, 2, 16, 0.5, 16, 0, z, 0, 0, 0, 0.0, 1.0, s) == 0
    assert lib.enerf_grid_encode_backward(z, z, z, z, z, 0, 3, 2, 16, 0.5, 16, 0, z, z, 0, 0, 0, 0.0, 1.0, s) == 0
    assert lib.enerf_sh_encode_forward(z, z, 0, 3, 4, 0, z, 0, s) == 0
    assert lib.enerf_sh_encode_backward(z, z, 0, 3, 4, z, z, 0, s) == 0
    assert lib.enerf_ffmlp_forward(z, z, 0, 32, 16, 64, 2, 0, 6, z, z, 2, s) == 0
    assert lib.enerf_ffmlp_inference(z, z, 0, 32, 16, 64, 2, 0, 6, z, z, 2, s) == 0
    assert lib.enerf_ffmlp_backward(z, z, z, z, 0, 32, 16, 64, 2, 0, 6, 1, z, z, z, 2, s) == 0
    assert lib.enerf_mlp32_forward(z, z, 0, 32, 16, 1, 0, 6, z, z, 0, 0, z, s) == 0
    assert lib.enerf_mlp32_backward(z, z, z, z, 0, 32, 16, 1, 0, z, z, z, 0, 0, z, 0, z, z, 0, s) == 0
    torch.cuda.synchronize()
    # and through the host-side mirrors
    from enerf_amd.network import NeRFNetwork
    m = NeRFNetwork(encoding="hashgrid", bound=1, cuda_ray=True, out_dim_color=3).to(DEV)
    sg, c = m(torch.empty(0, 3, device=DEV), torch.empty(0, 3, device=DEV))
    assert sg.shape == (0,) and c.shape == (0, 3)


# ------------------------------------------------------------------ composite_rays_train: 1e-4 rel
@pytest.mark.parametrize("bound,N", [(1, 200), (3, 4096)])
def test_composite_rays_train_fwd_bwd(rm, scenes, bound, N):
    grid, bits, C = scenes[bound]
    o, d, aabb = _rays(N, 30 + bound, bound)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.2)
    xyzs, dirs, deltas, rays, counter = O.march_rays_train(o, d, bits, bound, 0.0, 1024, C, H, N * 1024, nears, fars, 1)
    tot = int(counter[0]); M = tot + 128 - tot % 128
    rng = np.random.default_rng(4)
    sig = (rng.random(M) * 25).astype(np.float32)
    sig[rng.random(M) < 0.1] = 0
    rgb = rng.random((M, 3)).astype(np.float32)
    dl = deltas[:M]
    ws, depth, image = O.composite_rays_train_forward(sig, rgb, dl, rays)
    g_ws = torch.empty(N, device=DEV); g_d = torch.empty(N, device=DEV); g_im = torch.empty(N, 3, device=DEV)
    rm.composite_rays_train_forward(cu(sig), cu(rgb), cu(dl), cu(rays), M, N, g_ws, g_d, g_im)
    assert_close(g_ws, ws, rtol=1e-4, atol=1e-6)
    assert_close(g_im, image, rtol=1e-4, atol=1e-6)
    assert_close(g_d, depth, rtol=1e-4, atol=1e-5)
    gws = rng.normal(size=N).astype(np.float32); gim = rng.normal(size=(N, 3)).astype(np.float32)
    gs_ref, gc_ref = O.composite_rays_train_backward(gws, gim, sig, rgb, dl, rays, ws, image)
    gs = torch.zeros(M, device=DEV); gc = torch.zeros(M, 3, device=DEV)
    rm.composite_rays_train_backward(cu(gws), cu(gim), cu(sig), cu(rgb), cu(dl), cu(rays), g_ws, g_im, M, N, gs, gc)
    assert_close(gc, gc_ref, rtol=1e-4, atol=1e-6)
    # grad_sigma involves the cancellation (final - running): compare at the scale of the per-ray gradient
    scale = np.abs(gs_ref).max()
    assert np.abs(gs.cpu().numpy() - gs_ref).max() < 2e-5 * max(scale, 1.0)


def test_composite_rays_train_dropped_and_empty_rays(rm):
    # rays = (index, offset, count): empty ray, overflowing ray (offset+count >= M), permuted indices
    M = 64
    rays = np.array([[2, 0, 10], [0, 10, 0], [1, 10, 54], [3, 10, 53]], np.int32)
    rng = np.random.default_rng(5)
    sig = rng.random(M).astype(np.float32) * 5; rgb = rng.random((M, 3)).astype(np.float32)
    dl = np.full((M, 2), 0.01, np.float32)
    ws, depth, image = O.composite_rays_train_forward(sig, rgb, dl, rays)
    a = torch.full((4,), 7.0, device=DEV); b = torch.full((4,), 7.0, device=DEV); c = torch.full((4, 3), 7.0, device=DEV)
    rm.composite_rays_train_forward(cu(sig), cu(rgb), cu(dl), cu(rays), M, 4, a, b, c)
    assert_close(a, ws, rtol=1e-5, atol=1e-7); assert_close(c, image, rtol=1e-5, atol=1e-7)
    assert ws[1] == 0 and ws[0] == 0 and ws[2] > 0       # ray index 1 overflows (10+54 >= 64), index 0 is empty


@pytest.mark.parametrize("bg_kind,overflow", [("scalar", True), ("rgb", False), ("per_ray", True), ("scalar", False)])
def test_composite_blend_forward_and_mse_backward(rm, scenes, bg_kind, overflow):
    """The fused forms (include/enerf_hip.h: ..._forward_blend / ..._backward_mse) against the plain kernels plus the
    elementwise steps they absorb: image + (1 - ws) * bg;  grad_image = (out - target) * scale,  grad_ws = -(g . bg);
    zero gradients wherever no ray reaches, on NaN-filled outputs.  Includes an empty and a dropped ray."""
    bound, N = 2, 1500
    grid, bits, C = scenes[bound]
    o, d, aabb = _rays(N, 41, bound)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.2)
    tot = int(O.march_rays_train(o, d, bits, bound, 0.0, 1024, C, H, N * 1024, nears, fars, 1)[4][0])
    M = int(tot * 0.9) if overflow else tot + 2000            # overflow: the last rays are dropped
    M += 128 - M % 128
    xyzs, dirs, deltas, rays, counter = O.march_rays_train(o, d, bits, bound, 0.0, 1024, C, H, M, nears, fars, 1)
    assert (counter[0] > M) == overflow and (rays[:, 2] == 0).any()
    rng = np.random.default_rng(8)
    sig = cu((rng.random(M) * 25).astype(np.float32)); rgb = cu(rng.random((M, 3)).astype(np.float32))
    dl = cu(deltas[:M]); rays_t = cu(rays); cnt = cu(counter)
    bg = {"scalar": 0.75, "rgb": cu(rng.random(3).astype(np.float32)),
          "per_ray": cu(rng.random((N, 3)).astype(np.float32))}[bg_kind]
    ws0 = torch.empty(N, device=DEV); dp0 = torch.empty(N, device=DEV); im0 = torch.empty(N, 3, device=DEV)
    rm.composite_rays_train_forward(sig, rgb, dl, rays_t, M, N, ws0, dp0, im0)
    for depth in (torch.empty(N, device=DEV), None):
        ws1 = torch.empty(N, device=DEV); im1 = torch.empty(N, 3, device=DEV); out = torch.empty(N, 3, device=DEV)
        rm.composite_rays_train_forward_blend(sig, rgb, dl, rays_t, M, N, ws1, depth, im1, bg, out)
        assert torch.equal(ws1, ws0) and torch.equal(im1, im0)
        if depth is not None:
            assert torch.equal(depth, dp0)
        assert torch.equal(out, im0 + (1 - ws0).unsqueeze(-1) * bg)
    target = cu(rng.random((N, 3)).astype(np.float32))
    scale = 2.0 / (3 * N) * 1.7
    g_im = (out - target) * scale
    g_ws = -(g_im * bg).sum(-1)
    gs0 = torch.zeros(M, device=DEV); gc0 = torch.zeros(M, 3, device=DEV)
    rm.composite_rays_train_backward(g_ws.contiguous(), g_im.contiguous(), sig, rgb, dl, rays_t, ws0, im0, M, N, gs0, gc0)
    gs1 = torch.full((M,), float("nan"), device=DEV); gc1 = torch.full((M, 3), float("nan"), device=DEV)
    loss = torch.full((1,), 2.0, device=DEV)
    rm.composite_rays_train_backward_mse(out, target, scale, bg, cnt, sig, rgb, dl, rays_t, ws0, im0, M, N, gs1, gc1, loss)
    want = 2.0 + float(torch.nn.functional.mse_loss(out, target))             # the loss value is ADDED to the scalar
    assert abs(float(loss) - want) < 1e-5 * want
    assert torch.isfinite(gs1).all() and torch.isfinite(gc1).all()
    assert float((gc1 - gc0).abs().max()) <= 1e-6 * float(gc0.abs().max())
    assert float((gs1 - gs0).abs().max()) <= 2e-5 * float(gs0.abs().max())
    covered = torch.zeros(M, dtype=torch.bool, device=DEV)
    for i, off, n in rays.tolist():
        if n and off + n < M:
            covered[off:off + n] = True
    assert not gs1[~covered].any() and not gc1[~covered].any() and (~covered).sum() > (0 if overflow else 2000)
    # both halves in one launch (..._fwd_bwd_mse): images and gradients bit-identical to forward_blend + backward_mse
    ws2 = torch.full((N,), float("nan"), device=DEV); im2 = torch.full((N, 3), float("nan"), device=DEV)
    out2 = torch.full((N, 3), float("nan"), device=DEV)
    gs2 = torch.full((M,), float("nan"), device=DEV); gc2 = torch.full((M, 3), float("nan"), device=DEV)
    loss2 = torch.full((1,), 2.0, device=DEV)
    rm.composite_rays_train_fwd_bwd_mse(sig, rgb, dl, rays_t, M, N, ws2, im2, bg, out2, target, scale, cnt, gs2, gc2, loss2)
    assert torch.equal(ws2, ws0) and torch.equal(im2, im0) and torch.equal(out2, out)
    assert torch.equal(gs2, gs1) and torch.equal(gc2, gc1)
    assert abs(float(loss2) - float(loss)) <= 1e-6 * float(loss)       # (per-workgroup float atomics: order varies)
    gs3 = torch.full((M,), float("nan"), device=DEV); gc3 = torch.full((M, 3), float("nan"), device=DEV)
    rm.composite_rays_train_fwd_bwd_mse(sig, rgb, dl, rays_t, M, N, ws2, im2, bg, out2, target, scale, cnt, gs3, gc3)
    assert torch.equal(gs3, gs1) and torch.equal(gc3, gc1)                    # without the loss value


# ------------------------------------------------------------------ inference trio
def test_inference_loop_march_composite_compact(rm, scenes):
    bound = 2
    grid, bits, C = scenes[bound]
    N = 3000
    o, d, aabb = _rays(N, 40, bound)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.2)
    rng = np.random.default_rng(6)
    # oracle state
    ws = np.zeros(N, np.float32); dp = np.zeros(N, np.float32); im = np.zeros((N, 3), np.float32)
    alive = np.arange(N, dtype=np.int32); rt = nears.copy()
    # gpu state
    g_ws = torch.zeros(N, device=DEV); g_dp = torch.zeros(N, device=DEV); g_im = torch.zeros(N, 3, device=DEV)
    g_alive = torch.zeros(2, N, dtype=torch.int32, device=DEV); g_rt = torch.zeros(2, N, device=DEV)
    g_alive[0] = torch.arange(N, dtype=torch.int32, device=DEV); g_rt[0] = cu(nears)
    cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
    co, cd, cb, cn, cf = cu(o), cu(d), cu(bits), cu(nears), cu(fars)
    n_alive, i, step = N, 0, 0
    for it in range(200):
        if it > 0:
            new_alive, new_t, n_new = O.compact_rays(n_alive, alive, rt)
            cnt.zero_()
            rm.compact_rays(n_alive, g_alive[i % 2], g_alive[(i + 1) % 2], g_rt[i % 2], g_rt[(i + 1) % 2], cnt)
            assert int(cnt.item()) == n_new
            assert np.array_equal(g_alive[i % 2][:n_new].cpu().numpy(), new_alive[:n_new])     # stable order
            alive, rt, n_alive = new_alive[:n_new].copy(), new_t[:n_new].copy(), n_new
            # keep both sides' t identical so the marcher inputs stay bit-equal
            g_rt[i % 2][:n_new] = cu(rt)
        if n_alive <= 0:
            break
        n_step = max(min(N // n_alive, 8), 1)
        Mi = n_alive * n_step; Mi += 128 - Mi % 128
        x, dd, dl = O.march_rays(n_alive, n_step, alive, rt, o, d, bound, 0.0, 1024, C, H, bits, nears, fars, Mi, 0)
        gx = torch.zeros(Mi, 3, device=DEV); gd = torch.zeros(Mi, 3, device=DEV); gl = torch.zeros(Mi, 2, device=DEV)
        rm.march_rays(n_alive, n_step, g_alive[i % 2], g_rt[i % 2], co, cd, bound, 0.0, 1024, C, H, cb, cn, cf, gx, gd,
                      gl, 0)
        assert np.array_equal(gx.cpu().numpy(), x) and np.array_equal(gl.cpu().numpy(), dl)
        assert np.array_equal(gd.cpu().numpy(), dd)
        sig = (rng.random(Mi) * 30).astype(np.float32); rgb = rng.random((Mi, 3)).astype(np.float32)
        rt_o = rt.copy()
        O.composite_rays(n_alive, n_step, alive, rt_o, sig, rgb, dl, ws, dp, im)
        rm.composite_rays(n_alive, n_step, g_alive[i % 2], g_rt[i % 2], cu(sig), cu(rgb), gl, g_ws, g_dp, g_im)
        got_t = g_rt[i % 2][:n_alive].cpu().numpy()
        assert np.array_equal(got_t < 0, rt_o < 0)                       # same rays terminate
        np.testing.assert_allclose(got_t, rt_o, rtol=1e-6, atol=1e-6)
        rt = rt_o
        step += n_step; i += 1
    assert it > 5
    assert_close(g_ws, ws, rtol=1e-4, atol=1e-6)
    assert_close(g_im, im, rtol=1e-4, atol=1e-6)
    assert_close(g_dp, dp, rtol=1e-4, atol=1e-5)


def test_inference_march_perturb_seed_bit_exact(rm, scenes):
    bound = 3
    grid, bits, C = scenes[bound]
    N = 700
    o, d, aabb = _rays(N, 41, bound)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.2)
    alive = np.random.default_rng(7).permutation(N).astype(np.int32)[:500]
    rt = nears[alive]
    M = 500 * 4
    x, dd, dl = O.march_rays(500, 4, alive, rt, o, d, bound, 1.0 / 256, 1024, C, H, bits, nears, fars, M, 5)
    gx = torch.zeros(M, 3, device=DEV); gd = torch.zeros(M, 3, device=DEV); gl = torch.zeros(M, 2, device=DEV)
    rm.march_rays(500, 4, cu(alive), cu(rt), cu(o), cu(d), bound, 1.0 / 256, 1024, C, H, cu(bits), cu(nears), cu(fars),
                  gx, gd, gl, 5)
    assert np.array_equal(gx.cpu().numpy(), x) and np.array_equal(gl.cpu().numpy(), dl)


@pytest.mark.parametrize("n_alive,n_step,dt_gamma", [(3000, 8, 0.0), (70000, 8, 0.0), (900, 40, 0.0), (1500, 6, 1.0 / 256)])
def test_march_rays_ex_zero_fills_what_it_does_not_write(rm, scenes, n_alive, n_step, dt_gamma):
    """march_rays_ex on NaN-filled buffers == march_rays on zero-filled ones, bit for bit (thread-per-ray, wave-per-ray
    and the dt_gamma != 0 marcher): slots a ray does not fill and the alignment rows come out zero."""
    bound = 2
    grid, bits, C = scenes[bound]
    N = max(n_alive, 4000)
    o, d, aabb = _rays(N, 43, bound)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.2)
    rng = np.random.default_rng(9)
    alive = rng.permutation(N).astype(np.int32)[:n_alive]
    rt = (nears[alive] + rng.random(n_alive).astype(np.float32) * 2.0).astype(np.float32)     # some rays are nearly done
    M = n_alive * n_step
    M += 128 - M % 128
    args = (n_alive, n_step, cu(alive), cu(rt), cu(o), cu(d), bound, dt_gamma, 1024, C, H, cu(bits), cu(nears), cu(fars))
    z = [torch.zeros(M, k, device=DEV) for k in (3, 3, 2)]
    rm.march_rays(*args, *z, 0)
    e = [torch.full((M, k), float("nan"), device=DEV) for k in (3, 3, 2)]
    rm.march_rays_ex(*args, *e, 0)
    for a, b, name in zip(e, z, ("xyzs", "dirs", "deltas")):
        assert torch.equal(a, b), name
    filled = (z[2][:n_alive * n_step, 0] != 0).view(n_alive, n_step).sum(1)
    assert int((filled < n_step).sum()) > 0 and int((filled == n_step).sum()) > 0     # both kinds of ray occur


# ------------------------------------------------------------------ grid encoder
def _table(offsets, C, seed):
    rng = np.random.default_rng(seed)
    return rng.uniform(-1, 1, (int(offsets[-1]), C)).astype(np.float32)


@pytest.mark.parametrize("D,C,gridtype", [(3, 2, 0), (3, 1, 0), (3, 4, 0), (3, 8, 1), (2, 2, 0)])
def test_grid_encode_forward_backward_small(D, C, gridtype):
    from enerf_amd.backends import _gridencoder as ge
    offsets, pls = O.grid_offsets(input_dim=D, num_levels=8, level_dim=C, base_resolution=4, log2_hashmap_size=10,
                                  desired_resolution=160)
    S = float(np.log2(pls)); Hb = 4; L = 8; B = 777
    emb = _table(offsets, C, 50 + C)
    rng = np.random.default_rng(51)
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    x[0] = 0.0; x[1] = 1.0; x[2, 0] = 1.2; x[3, 1] = -0.1           # corners + out-of-range
    ref_out, ref_jac = O.grid_encode_forward(x, emb, offsets, S, Hb, True, gridtype)
    for layout in (0, 1):
        out = torch.empty((L, B, C) if layout == 0 else (B, L * C), device=DEV)
        jac = torch.empty(B, L * D * C, device=DEV)
        ge.grid_encode_forward(cu(x), cu(emb), cu(offsets), out, B, D, C, L, S, Hb, True, jac, gridtype, layout=layout)
        got = out.cpu().numpy() if layout == 0 else out.cpu().numpy().reshape(B, L, C).transpose(1, 0, 2)
        np.testing.assert_allclose(got, ref_out, rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(jac.cpu().numpy(), ref_jac, rtol=1e-4, atol=2e-4)
        g = rng.normal(size=(L, B, C)).astype(np.float32)
        ge_ref, gi_ref = O.grid_encode_backward(g, x, emb, offsets, S, Hb, ref_jac, gridtype)
        gemb = torch.zeros(emb.shape, device=DEV); gin = torch.zeros(B, D, device=DEV)
        gg = cu(g) if layout == 0 else cu(np.ascontiguousarray(g.transpose(1, 0, 2).reshape(B, L * C)))
        ge.grid_encode_backward(gg, cu(x), cu(emb), cu(offsets), gemb, B, D, C, L, S, Hb, True, jac, gin, gridtype,
                                layout=layout)
        np.testing.assert_allclose(gemb.cpu().numpy(), ge_ref, rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(gin.cpu().numpy(), gi_ref, rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("D,C,gridtype", [(3, 2, 0), (3, 1, 0), (3, 4, 1), (2, 2, 0), (3, 8, 0)])
def test_grid_backward_binned_path(D, C, gridtype):
    """The binned backward (record lists per table tile, summed in LDS), forced on a small table so that every branch
    runs: several tiles per level, replica lists on the coarse levels, ray-ordered samples (in-wave run aggregation),
    all three gradient layouts, accumulation into a non-zero gradient buffer."""
    from enerf_amd import _lib
    from enerf_amd.backends import _gridencoder as ge
    offsets, pls = O.grid_offsets(input_dim=D, num_levels=10, level_dim=C, base_resolution=4, log2_hashmap_size=16,
                                  desired_resolution=512)
    S = float(np.log2(pls)); Hb = 4; L = 10; B = 20011
    emb = _table(offsets, C, 55 + C)
    rng = np.random.default_rng(56)
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    t = np.arange(6000, dtype=np.float32)[:, None] % 60          # 100 "rays" of 60 closely spaced samples
    o = np.repeat(rng.uniform(0.2, 0.6, (100, D)), 60, 0); d = np.repeat(rng.uniform(-1, 1, (100, D)), 60, 0)
    x[:6000] = (o + d * t * 0.002).astype(np.float32)
    x[7] = 1.5
    g = rng.normal(size=(L, B, C)).astype(np.float32)
    ge_ref, _ = O.grid_encode_backward(g, x, emb, offsets, S, Hb, None, gridtype)
    Bp = (B + 31) // 32 * 32
    dummy = torch.empty(1, device=DEV)
    lib = _lib.lib()
    try:
        for layout in (0, 1, 2):
            if layout == 0:
                gg = cu(g)
            elif layout == 1:
                gg = cu(np.ascontiguousarray(g.transpose(1, 0, 2).reshape(B, L * C)))
            else:
                gp = np.zeros((L, Bp, C), np.float32); gp[:, :B] = g
                gg = cu(gp)
            res = []
            # binned path with one list per tile / with replica lists on the small levels; global-atomic kernel
            for min_batch, min_tiles in ((0, 1), (0, 8), (0xffffffff, 8)):
                lib.enerf_debug_grid_bwd_binned(min_batch, min_tiles)
                gemb = torch.full(emb.shape, 0.5, device=DEV)
                ge.grid_encode_backward(gg, cu(x), cu(emb), cu(offsets), gemb, B, D, C, L, S, Hb, False, dummy, dummy,
                                        gridtype, layout=layout)
                res.append(gemb.cpu().numpy() - 0.5)
            scale = np.abs(ge_ref).max()
            for r in res:
                np.testing.assert_allclose(r, ge_ref, rtol=1e-4, atol=2e-5 * scale)
    finally:
        lib.enerf_debug_grid_bwd_binned(16384, 8)


def test_grid_encode_half_table():
    from enerf_amd.backends import _gridencoder as ge
    offsets, pls = O.grid_offsets(num_levels=8, base_resolution=4, log2_hashmap_size=10, desired_resolution=160)
    S = float(np.log2(pls)); L = 8; B = 500; C = 2
    emb = _table(offsets, C, 60).astype(np.float16)
    x = np.random.default_rng(61).uniform(0, 1, (B, 3)).astype(np.float32)
    ref, _ = O.grid_encode_forward(x, emb.astype(np.float32), offsets, S, 4)
    out = torch.empty(L, B, C, device=DEV, dtype=torch.half)
    dummy = torch.empty(1, device=DEV, dtype=torch.half)
    ge.grid_encode_forward(cu(x), cu(emb), cu(offsets), out, B, 3, C, L, S, 4, False, dummy, 0)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, atol=2e-3)
    g = np.random.default_rng(62).normal(size=(L, B, C)).astype(np.float16)
    ge_ref, _ = O.grid_encode_backward(g.astype(np.float32), x, emb.astype(np.float32), offsets, S, 4)
    gemb = torch.zeros(emb.shape, device=DEV, dtype=torch.half)
    ge.grid_encode_backward(cu(g), cu(x), cu(emb), cu(offsets), gemb, B, 3, C, L, S, 4, False, dummy, dummy, 0)
    np.testing.assert_allclose(gemb.float().cpu().numpy(), ge_ref, atol=3e-2, rtol=2e-2)


@pytest.mark.parametrize("bound", [2, 3])
def test_grid_encode_full_size_tables(bound):
    """BASELINE table sizes (6.3M / 6.5M rows, L=16, C=2, T=2^19): direct comparison on 100k points plus the
    size-independent properties: linearity in the table and sum(grad_embeddings) == sum(grad) (weights sum to 1)."""
    from enerf_amd.backends import _gridencoder as ge
    offsets, pls = O.grid_offsets(desired_resolution=2048 * bound)
    S = float(np.log2(pls)); L = 16; C = 2; B = 100000
    rng = np.random.default_rng(70 + bound)
    emb1 = rng.uniform(-1, 1, (int(offsets[-1]), C)).astype(np.float32)
    emb2 = rng.uniform(-1, 1, (int(offsets[-1]), C)).astype(np.float32)
    x = rng.uniform(0, 1, (B, 3)).astype(np.float32)
    ref, _ = O.grid_encode_forward(x, emb1, offsets, S, 16)
    cx, co = cu(x), cu(offsets)
    dummy = torch.empty(1, device=DEV)

    def enc(e):
        out = torch.empty(L, B, C, device=DEV)
        ge.grid_encode_forward(cx, e, co, out, B, 3, C, L, S, 16, False, dummy, 0)
        return out
    e1, e2 = cu(emb1), cu(emb2)
    y1 = enc(e1)
    np.testing.assert_allclose(y1.cpu().numpy(), ref, rtol=1e-5, atol=3e-6)
    assert_close(enc(e1 + 2 * e2), y1 + 2 * enc(e2), rtol=1e-4, atol=1e-5)
    g = rng.normal(size=(L, B, C)).astype(np.float32)
    gemb = torch.zeros_like(e1)
    ge.grid_encode_backward(cu(g), cx, e1, co, gemb, B, 3, C, L, S, 16, False, dummy, dummy, 0)
    for l in range(L):
        a, b = int(offsets[l]), int(offsets[l + 1])
        np.testing.assert_allclose(gemb[a:b].sum(0).cpu().numpy(), g[l].sum(0), rtol=1e-3, atol=2e-2)
    ge_ref, _ = O.grid_encode_backward(g, x, emb1, offsets, S, 16)
    np.testing.assert_allclose(gemb.cpu().numpy(), ge_ref, rtol=1e-4, atol=1e-4)


def test_grid_encode_2d_vs_torch():
    from enerf_amd.backends import _gridencoder as ge
    offsets, pls = O.grid_offsets(input_dim=2, num_levels=4, base_resolution=16, log2_hashmap_size=19,
                                  desired_resolution=2048)
    S = float(np.log2(pls)); L = 4; C = 2; B = 1000
    rng = np.random.default_rng(80)
    emb = rng.uniform(-1, 1, (int(offsets[-1]), C)).astype(np.float32)
    x = rng.uniform(0, 1, (B, 2)).astype(np.float32)
    out = torch.empty(L, B, C, device=DEV); dummy = torch.empty(1, device=DEV)
    ge.grid_encode_forward(cu(x), cu(emb), cu(offsets), out, B, 2, C, L, S, 16, False, dummy, 0)
    got = out.cpu().numpy()
    for l in range(L):
        scale, res = O.grid_level_params(l, np.float32(S), 16)
        size = int(offsets[l + 1] - offsets[l])
        pos = x.astype(np.float64) * scale + 0.5
        pg = np.floor(pos).astype(np.int64); fr = pos - pg
        acc = np.zeros((B, C))
        for idx in range(4):
            cxs = pg[:, 0] + (idx & 1); cys = pg[:, 1] + ((idx >> 1) & 1)
            w = (fr[:, 0] if idx & 1 else 1 - fr[:, 0]) * (fr[:, 1] if idx & 2 else 1 - fr[:, 1])
            stride, index = 1, np.zeros(B, np.int64)
            for cc in (cxs, cys):
                if stride <= size:
                    index = index + cc * stride
                    stride *= res + 1
            if stride > size:
                index = (cxs * 1) ^ ((cys * 2654435761) & 0xffffffff)
            index = (index & 0xffffffff) % size
            acc += w[:, None] * emb[offsets[l] + index]
        np.testing.assert_allclose(got[l], acc, atol=max(3e-5, scale * 1.2e-7))   # fp32 pos = x*scale+0.5
    ref, _ = O.grid_encode_forward(x, emb, offsets, S, 16)
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=2e-6)


# ------------------------------------------------------------------ SH encoder
@pytest.mark.parametrize("deg", [1, 2, 3, 4, 5, 6, 7, 8])
def test_sh_encode(deg):
    from enerf_amd.backends import _shencoder as sh
    rng = np.random.default_rng(90 + deg)
    B = 1001
    v = rng.normal(size=(B, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    v[5] *= 0.5
    ref, jref = O.sh_encode_forward(v, deg, True)
    out = torch.empty(B, deg * deg, device=DEV); jac = torch.empty(B, 3 * deg * deg, device=DEV)
    sh.sh_encode_forward(cu(v), out, B, 3, deg, True, jac)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-4, atol=2e-6 * deg)
    np.testing.assert_allclose(jac.cpu().numpy(), jref, rtol=1e-4, atol=2e-5 * deg * deg)
    out2 = torch.empty(B, deg * deg, device=DEV)
    sh.sh_encode_forward(cu(v), out2, B, 3, deg, False, torch.empty(1, device=DEV))
    assert torch.equal(out, out2)
    g = rng.normal(size=(B, deg * deg)).astype(np.float32)
    gi_ref = O.sh_encode_backward(g, v, deg, jref)
    gi = torch.zeros(B, 3, device=DEV)
    sh.sh_encode_backward(cu(g), cu(v), B, 3, deg, jac, gi)
    np.testing.assert_allclose(gi.cpu().numpy(), gi_ref, rtol=1e-3, atol=1e-4 * deg * deg)


@pytest.mark.parametrize("deg", [1, 2, 3, 4, 5, 6, 7, 8])
def test_sh_encode_vs_reference_literal_tables(deg):
    """HIP sh_encode against the reference's literal polynomial tables (shencoder.cu:51-121,131-351) evaluated in
    float64 at mint time (tests/golden/ref_sh_literals.npz), on and off the unit sphere."""
    from enerf_amd.backends import _shencoder as sh
    from util import golden
    g = golden("ref_sh_literals")
    d, Y, J = g["d"], g["y"], g["dy_dx"]
    B, C2 = len(d), deg * deg
    out = torch.empty(B, C2, device=DEV); jac = torch.empty(B, 3 * C2, device=DEV)
    sh.sh_encode_forward(cu(d), out, B, 3, deg, True, jac)
    np.testing.assert_allclose(out.cpu().numpy(), Y[:, :C2], rtol=1e-4, atol=3e-6)
    np.testing.assert_allclose(jac.cpu().numpy().reshape(B, 3, C2), J[:, :, :C2], rtol=1e-4, atol=3e-5)


def test_sh_encode_half():
    from enerf_amd.backends import _shencoder as sh
    v = np.random.default_rng(99).normal(size=(300, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    vh = v.astype(np.float16)
    ref, _ = O.sh_encode_forward(vh.astype(np.float32), 4)
    out = torch.empty(300, 16, device=DEV, dtype=torch.half)
    sh.sh_encode_forward(cu(vh), out, 300, 3, 4, False, torch.empty(1, device=DEV, dtype=torch.half))
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, atol=2e-3)


# ------------------------------------------------------------------ end to end: renderer on the HIP path vs the
# same renderer on the CPU oracle backend (same weights, same rays)
def _cpu_model_and_outputs(bound, o, d, bits, train, monkeypatch):
    import enerf_amd.raymarching as rmod, enerf_amd.gridencoder as gmod, enerf_amd.shencoder as smod
    from oracle import backend as ob
    from enerf_amd.network import NeRFNetwork
    with monkeypatch.context() as mp:
        mp.setattr(rmod, "_backend", ob.raymarching_backend); mp.setattr(rmod, "_DEVICE", "cpu")
        mp.setattr(gmod, "_backend", ob.gridencoder_backend); mp.setattr(smod, "_backend", ob.shencoder_backend)
        torch.manual_seed(0)
        model = NeRFNetwork(encoding="hashgrid", bound=bound, cuda_ray=True, out_dim_color=3)
        g = torch.Generator().manual_seed(5)
        model.encoder.embeddings.data.copy_(torch.rand(model.encoder.embeddings.shape, generator=g) * 2 - 1)
        model.density_bitfield.copy_(torch.from_numpy(bits))
        state = {k: v.clone() for k, v in model.state_dict().items()}
        model.train(train)
        ro, rd = torch.from_numpy(o)[None], torch.from_numpy(d)[None]
        if train:
            out = model.render(ro, rd, staged=False, bg_color=torch.full((3,), 0.3), perturb=True, force_all_rays=True)
            ((out["image"] ** 2).sum() + out["depth"].sum()).backward()
            grads = {n: p.grad.clone() for n, p in model.named_parameters()}
            return state, out, grads, model.step_counter.clone()
        with torch.no_grad():
            out = model.render(ro, rd, staged=False, bg_color=None, perturb=False)
        return state, out, None, None


@pytest.mark.parametrize("train", [True, False])
def test_render_end_to_end_vs_cpu_oracle(scenes, monkeypatch, train, mlp32_mode):
    from enerf_amd.network import NeRFNetwork
    bound = 2
    grid, bits, C = scenes[bound]
    o, d, _ = _rays(192, 55, bound)
    state, ref, ref_grads, ref_counter = _cpu_model_and_outputs(bound, o, d, bits, train, monkeypatch)
    model = NeRFNetwork(encoding="hashgrid", bound=bound, cuda_ray=True, out_dim_color=3)
    model.load_state_dict(state)
    model.to(DEV).train(train)
    ro, rd = cu(o)[None], cu(d)[None]
    if train:
        out = model.render(ro, rd, staged=False, bg_color=torch.full((3,), 0.3, device=DEV), perturb=True,
                           force_all_rays=True)
        ((out["image"] ** 2).sum() + out["depth"].sum()).backward()
        assert torch.equal(model.step_counter.cpu(), ref_counter)            # sample counts bit-exact
        assert_close(out["image"], ref["image"], rtol=1e-4, atol=2e-5)
        assert_close(out["depth"], ref["depth"], rtol=1e-4, atol=2e-5)
        for n, p in model.named_parameters():
            r = ref_grads[n]
            top = float(r.abs().max())
            err = (p.grad.cpu() - r).abs()
            if mlp32_mode == "fp32":
                assert float(err.max()) < 2e-4 * top + 1e-7, n
            else:
                # ~7 k samples x 192 hidden units at ~1e-5 forward error: a few sit on the other side of their ReLU than
                # on the oracle route, and each moves its own sample's terms (a handful of entries) by a whole term
                assert float(err.max()) < 2e-3 * top + 1e-7, n
                assert float((err > 2e-4 * top + 1e-7).float().mean()) < 0.02, n
    else:
        with torch.no_grad():
            out = model.render(ro, rd, staged=False, bg_color=None, perturb=False)
        assert_close(out["image"], ref["image"], rtol=1e-4, atol=2e-5)
        assert_close(out["depth"], ref["depth"], rtol=1e-4, atol=2e-5)


def test_inference_batch_mult_gives_identical_image(scenes):
    """The K-times-larger inference batches (enerf_amd extension) only change the chunking: image bit-identical."""
    from enerf_amd.network import NeRFNetwork
    bound = 2
    grid, bits, C = scenes[bound]
    torch.manual_seed(1)
    m = NeRFNetwork(encoding="hashgrid", bound=bound, cuda_ray=True, out_dim_color=3).to(DEV).eval()
    m.encoder.embeddings.data.uniform_(-1, 1)
    m.density_bitfield.copy_(torch.from_numpy(bits))
    o, d, _ = _rays(5000, 77, bound)
    ro, rd = cu(o)[None], cu(d)[None]
    with torch.no_grad():
        a = m.render(ro, rd, staged=False, bg_color=None, perturb=False)
        m.infer_batch_mult = 8
        b = m.render(ro, rd, staged=False, bg_color=None, perturb=False)
    assert torch.equal(a["image"], b["image"]) and torch.equal(a["depth"], b["depth"])


@pytest.mark.gpu
def test_march_count_fresh_counter_flag_equals_a_zeroed_counter():
    """Flag bit 3 of the count / _ex entry points: the counter is taken as (0, 0) whatever it holds -- same rays, same
    counter as zeroing it first (the reference wrapper's counter.zero_(), raymarching.py:198), for the one-workgroup
    scan and the tiled one (N > 16384)."""
    from enerf_amd.backends import _raymarching as rb
    from enerf_amd import raymarching, scene
    bound, C, H = 2, 2, 128
    bits = raymarching.packbits(scene.density_grid(bound, "cuda"), 0.01)
    for n in (4096, 20000):
        g = torch.Generator(device="cuda").manual_seed(n)
        (ro, rd), _ = scene.training_batch(0, n, "cuda", generator=g)
        aabb = torch.tensor([-bound] * 3 + [bound] * 3, dtype=torch.float32, device="cuda")
        nears, fars = raymarching.near_far_from_aabb(ro, rd, aabb, 0.2)
        out = []
        for fresh in (False, True):
            rays = torch.full((n, 3), -7, dtype=torch.int32, device="cuda")
            counter = torch.zeros(2, dtype=torch.int32, device="cuda") if not fresh else \
                torch.tensor([123456, 77], dtype=torch.int32, device="cuda")
            rb.march_rays_train_count(ro, rd, bits, bound, 0.0, 1024, n, C, H, nears, fars, rays, counter, 1,
                                      8 if fresh else 0)
            out.append((rays.clone(), counter.clone()))
        assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
        assert int(out[0][1][1]) == n and int(out[0][1][0]) > 0


@pytest.mark.gpu
def test_marcher_workspaces_are_ordered_between_streams():
    """The marcher's chunk log / scan / box workspaces are one set per device.  A training stage issued ahead on a side
    stream (fused_render.prefetch_march) and a march of other rays launched right behind it on the current stream (an
    evaluation render between two training steps) must not run on the same log at once: the library orders the second
    stream after the first (csrc/runtime.hip: workspace_family_enter).  200 rounds of exactly that, each compared bit
    for bit with the stage marched on its own.  (Unordered, a corrupted log is replayed into out-of-bounds rows: the
    memory access fault a 500-training run hit.)"""
    from enerf_amd import fused_render, raymarching, scene
    from enerf_amd.backends import _raymarching as rb
    from enerf_amd.network import NeRFNetwork
    torch.manual_seed(0)
    model = NeRFNetwork(encoding="hashgrid", bound=2, cuda_ray=True, out_dim_color=3).cuda()
    model.density_bitfield.copy_(raymarching.packbits(scene.density_grid(2, "cuda"), 0.01))
    model.mean_count = 140000
    g = torch.Generator(device="cuda").manual_seed(7)
    (ro, rd), _ = scene.training_batch(0, 4096, "cuda", generator=g)
    (ro2, rd2), _ = scene.training_batch(1, 16384, "cuda", generator=g)
    ro, rd, ro2, rd2 = (t.contiguous().view(-1, 3) for t in (ro, rd, ro2, rd2))
    aabb = model.aabb_train
    C, H = int(model.cascade), int(model.grid_size)

    def stage_alone():
        model._premarched = None
        fused_render.prefetch_march(model, ro, rd, perturb=True)
        pre = fused_render._take_premarched(model, ro, rd, True, 0, 1024)
        torch.cuda.synchronize()
        return {k: pre[k].clone() for k in ("rays", "xyzs", "dirs", "deltas")}, pre["counter"].clone()

    want, want_counter = stage_alone()
    side = torch.cuda.Stream()
    nears2, fars2 = raymarching.near_far_from_aabb(ro2, rd2, aabb, 0.2)
    for _ in range(200):
        model._premarched = None
        fused_render.prefetch_march(model, ro, rd, perturb=True, stream=side)
        rays2 = torch.empty(16384, 3, dtype=torch.int32, device="cuda")
        counter2 = torch.zeros(2, dtype=torch.int32, device="cuda")
        rb.march_rays_train_count(ro2, rd2, model.density_bitfield, model.bound, 0.0, 1024, 16384, C, H, nears2, fars2,
                                  rays2, counter2, 1, 8)
        pre = fused_render._take_premarched(model, ro, rd, True, 0, 1024)
        torch.cuda.synchronize()
        assert torch.equal(pre["counter"], want_counter)
        for k in want:
            assert torch.equal(pre[k], want[k]), k


def test_prof_hooks_time_one_launch_in_n_and_count_their_points():
    """enerf_prof_sample_every(n): launch k of a family is timed when k % n == 0 (counted from the last reset);
    enerf_prof_read_units returns the points of the TIMED launches only -- the pair bench.py's roofline is made of."""
    from enerf_amd import _lib
    from enerf_amd.backends import _gridencoder as gb
    from enerf_amd.gridencoder import GridEncoder
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=15,
                      desired_resolution=512).to(DEV)
    sizes = [1000, 2000, 3000, 4000, 5000, 6000, 7000]
    try:
        _lib.prof.reset()
        _lib.prof.sample_every(3)
        _lib.prof.enable(True, only=("grid_fwd",))
        for n in sizes:
            with torch.no_grad():
                enc(torch.rand(n, 3, device=DEV) * 2 - 1, bound=1)
        torch.cuda.synchronize()
        ms, timed = _lib.prof.read("grid_fwd")
        units, seen = _lib.prof.read_units("grid_fwd")
    finally:
        _lib.prof.enable(False)
        _lib.prof.sample_every(1)
        _lib.prof.reset()
    assert seen == len(sizes) and timed == 3                       # launches 0, 3, 6
    assert units == sizes[0] + sizes[3] + sizes[6]
    assert 0.0 < ms < 50.0
