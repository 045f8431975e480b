"""Device-side update_extra_state (csrc/density_update.hip) against the numpy restatement in oracle/density_update.py
and, end to end, against the op-by-op torch route of enerf_amd/renderer.py."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import density_update as OD
from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _lib():
    from enerf_amd import _lib as L
    return L


def _cells(grid, C, H, bound, N, seed):
    L = _lib()
    P = C * H ** 3 if grid is None else C * 2 * N
    idx = torch.empty(P, dtype=torch.int32, device=DEV)
    xyz = torch.full((P, 3), float("nan"), device=DEV)
    L.check(L.lib().enerf_density_grid_cells(None if grid is None else grid.data_ptr(), C, H, float(bound), N,
                                             ctypes.c_uint64(seed), idx.data_ptr(), xyz.data_ptr(), L.stream_handle()),
            "cells")
    torch.cuda.synchronize()
    return idx.cpu().numpy().reshape(C, -1), xyz.cpu().numpy().reshape(C, -1, 3)


def _check_positions(idx, xyz, bound, H):
    for cas in range(idx.shape[0]):
        span, half = OD.cascade_geometry(cas, bound, H)
        centre = OD.cell_centres(idx[cas], cas, bound, H)
        off = xyz[cas] - centre
        assert np.abs(off).max() <= half * (1 + 1e-5) + 1e-6
        assert np.abs(off).max() > 0.98 * half                       # the jitter fills the cell ...
        assert abs(off.mean()) < 0.02 * half                         # ... symmetrically
        assert np.abs(np.corrcoef(off[:, 0], off[:, 1])[0, 1]) < 0.02


@pytest.mark.parametrize("bound,H", [(3, 32), (1, 64), (2, 128)])
def test_full_sweep_visits_every_cell_once_in_x_fastest_order(bound, H):
    C = 1 + int(np.ceil(np.log2(bound)))
    idx, xyz = _cells(None, C, H, bound, 0, 1234)
    x, y, z = np.meshgrid(np.arange(H), np.arange(H), np.arange(H), indexing="ij")       # ij: z fastest -> transpose
    coords = np.stack([x.transpose(2, 1, 0).ravel(), y.transpose(2, 1, 0).ravel(), z.transpose(2, 1, 0).ravel()], -1)
    want = O.morton3D(coords.astype(np.int32))
    for cas in range(C):
        assert np.array_equal(idx[cas], want)
        assert np.array_equal(np.sort(idx[cas]), np.arange(H ** 3))
    _check_positions(idx, xyz, bound, H)
    idx2, xyz2 = _cells(None, C, H, bound, 0, 1235)
    assert np.array_equal(idx, idx2) and not np.array_equal(xyz, xyz2)       # another seed: same cells, other jitter
    idx3, xyz3 = _cells(None, C, H, bound, 0, 1234)
    assert np.array_equal(xyz, xyz3)                                          # same seed: reproducible


def test_partial_update_draws_uniform_and_occupied_cells_sorted():
    C, H, bound = 3, 64, 3
    H3 = H ** 3
    N = H3 // 4
    rng = np.random.default_rng(3)
    grid = np.zeros((C, H3), np.float32)
    occ = [rng.choice(H3, size=k, replace=False) for k in (40, 5000, 0)]     # cascade 2 has no occupied cell
    for cas in range(C):
        grid[cas, occ[cas]] = rng.random(len(occ[cas])).astype(np.float32) + 0.1
    grid[0, rng.choice(H3, 1000)] = -1.0                                      # untrained cells are not "occupied"
    grid[0, occ[0]] = 0.5
    idx, xyz = _cells(torch.from_numpy(grid).to(DEV), C, H, bound, N, 99)
    assert idx.shape == (C, 2 * N)
    for cas in range(C):
        assert np.all(np.diff(idx[cas]) >= 0) and idx[cas].min() >= 0 and idx[cas].max() < H3
        in_occ = np.isin(idx[cas], occ[cas]).sum()
        expect = N + N * len(occ[cas]) / H3 if len(occ[cas]) else 0
        assert abs(in_occ - expect) <= 5 * np.sqrt(N * max(len(occ[cas]), 1) / H3) + 1, (cas, in_occ, expect)
        if len(occ[cas]):
            hits = np.bincount(np.searchsorted(np.sort(occ[cas]), idx[cas][np.isin(idx[cas], occ[cas])]),
                               minlength=len(occ[cas]))
            lam = N / len(occ[cas]) + 2 * N / H3                             # every occupied cell gets its share:
            assert abs(hits.mean() - lam) < 0.02 * lam                       # Poisson(lam) counts
            assert 0.8 < hits.var() / lam < 1.25 if len(occ[cas]) > 1000 else hits.min() > 0.8 * lam
        # the uniform half covers the grid: about 1 - exp(-1/4) of all cells are hit at least once
        # (a cascade without occupied cells draws all 2N uniformly: 1 - exp(-1/2))
        frac = len(np.unique(idx[cas])) / H3
        want = 1 - np.exp(-0.25 if len(occ[cas]) else -0.5)
        assert abs(frac - want) < 0.03
    _check_positions(idx, xyz, bound, H)


@pytest.mark.parametrize("C,H", [(3, 32), (1, 64)])
def test_grid_update_matches_oracle(C, H):
    L = _lib()
    H3 = H ** 3
    rng = np.random.default_rng(11)
    grid = (rng.random((C, H3)) ** 4 * 0.05).astype(np.float32)
    grid[rng.random((C, H3)) < 0.3] = 0.0
    grid[rng.random((C, H3)) < 0.05] = -1.0                                   # mark_untrained_grid cells
    n = H3 // 2
    indices = np.stack([rng.choice(H3, size=n, replace=False) for _ in range(C)]).astype(np.int32)
    sigmas = (rng.random((C, n)) ** 3 * 40).astype(np.float32)
    sigmas[rng.random((C, n)) < 0.01] = np.nan                                # a diverged network output is ignored
    counter = rng.integers(1000, 500000, size=(16, 2)).astype(np.int32)
    scale, decay, thresh = 0.003383 * 1.5, 0.95, 0.01
    want_grid, want_mean, want_bits = OD.apply_update(grid, indices, np.nan_to_num(sigmas, nan=-1.0), scale, decay, thresh)
    g = torch.from_numpy(grid).to(DEV)
    bits = torch.zeros(C * H3 // 8, dtype=torch.uint8, device=DEV)
    stats = torch.zeros(2, dtype=torch.float64, device=DEV)
    d_idx, d_sig, d_cnt = (torch.from_numpy(a).to(DEV) for a in (indices, sigmas, counter))
    for total_step in (16, 5, 0):
        g.copy_(torch.from_numpy(grid))
        L.check(L.lib().enerf_density_grid_update(d_idx.data_ptr(), d_sig.data_ptr(), n, C, H, scale, decay, thresh,
                                                  g.data_ptr(), bits.data_ptr(), d_cnt.data_ptr(), total_step,
                                                  stats.data_ptr(), L.stream_handle()), "update")
        torch.cuda.synchronize()
        mean, counted = stats.tolist()
        assert counted == int(counter[:total_step, 0].sum())
        if total_step:
            assert int(counted / total_step) == OD.mean_count(counter, total_step)
    assert np.array_equal(g.cpu().numpy(), want_grid)                         # max / multiply: bit-exact
    assert abs(mean - want_mean) <= 1e-6 * want_mean
    got_bits = bits.cpu().numpy()
    # a cell within rounding of the threshold may fall either way (fp32 mean vs the oracle's fp64): none do here
    near = np.abs(np.clip(want_grid, 0, None) - min(want_mean, thresh)) < 1e-7
    assert not near.any() or np.mean(got_bits != want_bits) < 1e-4
    if not near.any():
        assert np.array_equal(got_bits, want_bits)


def test_duplicate_cells_take_one_of_their_values():
    L = _lib()
    C, H = 1, 32
    H3 = H ** 3
    indices = torch.tensor([5, 5, 5, 9], dtype=torch.int32, device=DEV)
    sigmas = torch.tensor([1.0, 2.0, 3.0, 4.0], device=DEV)
    g = torch.zeros(C, H3, device=DEV)
    bits = torch.zeros(H3 // 8, dtype=torch.uint8, device=DEV)
    stats = torch.zeros(2, dtype=torch.float64, device=DEV)
    L.check(L.lib().enerf_density_grid_update(indices.data_ptr(), sigmas.data_ptr(), 4, C, H, 1.0, 0.95, 0.01,
                                              g.data_ptr(), bits.data_ptr(), None, 0, stats.data_ptr(),
                                              L.stream_handle()), "update")
    assert float(g[0, 5]) in (1.0, 2.0, 3.0) and float(g[0, 9]) == 4.0 and int((g != 0).sum()) == 2


@pytest.mark.parametrize("partial", [False, True])
def test_update_extra_state_native_vs_torch_route(monkeypatch, partial):
    """Same model, same counters: the device-side update and the op-by-op torch route agree on what is deterministic
    (mean_count, bookkeeping) and, up to the different jitter / sampling draws, on the grid they produce."""
    from enerf_amd import density_update, scene
    from enerf_amd.network import NeRFNetwork
    res = []
    for native in (True, False):
        torch.manual_seed(0)
        model = NeRFNetwork(encoding="hashgrid", bound=2, cuda_ray=True, out_dim_color=3).to(DEV)
        model.encoder.embeddings.data.uniform_(-1.0, 1.0)
        monkeypatch.setattr(density_update, "ENABLED", native)
        if partial:
            scene.install_occupancy(model)
            model.iter_density = 16
        model.local_step = 7
        model.step_counter[:, 0] = torch.arange(16, dtype=torch.int32, device=DEV) * 1000 + 123456
        before = model.density_grid.clone()
        model.update_extra_state()
        res.append((model.density_grid.clone(), model.density_bitfield.clone(), model.mean_density, model.mean_count,
                    model.local_step, model.iter_density, before))
    (g1, b1, m1, c1, l1, i1, before), (g0, b0, m0, c0, l0, i0, _) = res
    assert c1 == c0 == int((123456 * 7 + 1000 * 21) / 7) and l1 == l0 == 0 and i1 == i0
    assert abs(m1 - m0) <= 0.05 * m0
    occ1, occ0 = (g1 > min(m1, 0.01)), (g0 > min(m0, 0.01))
    assert abs(int(occ1.sum()) - int(occ0.sum())) <= 0.05 * int(occ0.sum()) + 100
    # random-init hash features make the density at a jittered point of a cell noise-like: the two routes (different
    # jitter) agree on the distribution of the new grid, cell by cell only where the old grid dominates (partial)
    q = torch.tensor([0.1, 0.25, 0.5, 0.75, 0.9, 0.99], device=DEV)
    for cas in range(g1.shape[0]):
        q1, q0 = torch.quantile(g1[cas][::7], q), torch.quantile(g0[cas][::7], q)
        assert bool(((q1 - q0).abs() <= 0.05 * q0.abs() + 1e-5).all()), (cas, q1, q0)
    if partial:
        assert float((occ1 != occ0).float().mean()) < 0.05
    bits = torch.tensor(O.packbits(g1.cpu().numpy().reshape(-1), np.float32(min(m1, 0.01))), device=DEV)
    assert float((bits != b1).float().mean()) < 1e-5
    if partial:
        touched = (g1 != before * 1.0) | (g1 != before)
        assert 0.1 < float((g1 != before).float().mean()) < 0.8      # about half the cells are re-evaluated


def test_mark_untrained_grid_native_vs_oracle_and_torch_route(monkeypatch):
    """One launch per call against the numpy restatement and the reference-shaped 5-level torch loop: cells on a frustum
    boundary may fall either way (summation order of the 3x3 rotation), everything else must agree."""
    from enerf_amd import density_update, scene
    from enerf_amd.network import NeRFNetwork
    poses = np.stack([scene.pose(i).numpy() for i in range(0, 32, 5)]).astype(np.float32)
    intrinsic = (320.0, 320.0, 320.0, 240.0)
    grids = []
    for native in (True, False):
        model = NeRFNetwork(encoding="hashgrid", bound=2, cuda_ray=True, out_dim_color=3).to(DEV)
        model.density_grid.fill_(0.25)
        monkeypatch.setattr(density_update, "ENABLED", native)
        model.mark_untrained_grid(poses, intrinsic)
        grids.append(model.density_grid.cpu().numpy())
    want = OD.untrained_cells(poses, intrinsic, 2, 2, 128)
    for g in grids:
        assert set(np.unique(g)) == {-1.0, 0.25}
    frac = want.mean()
    assert 0.02 < frac < 0.98                                   # the cameras see part of the volume, not all of it
    assert np.mean((grids[0] == -1) != want) < 2e-4
    assert np.mean((grids[0] == -1) != (grids[1] == -1)) < 2e-4


@pytest.mark.gpu
def test_sweep_points_generated_inside_the_grid_kernel_equal_the_written_ones():
    """enerf_grid_encode_forward_sweep (query points generated in the kernel, csrc/sweep_points.h) against
    enerf_density_grid_cells + enerf_grid_encode_forward on the same seed: bit-identical features, and the whole update
    (density grid, bitfield, budget) identical with density_update.SWEEP_IN_KERNEL on and off."""
    import ctypes
    import numpy as np
    from enerf_amd import _lib as L, density_update, fused_network
    from enerf_amd.network import NeRFNetwork
    from enerf_amd.backends import _gridencoder as gb
    torch.manual_seed(3)
    model = NeRFNetwork(encoding="hashgrid", bound=2, cuda_ray=True, out_dim_color=3).cuda()
    enc = model.encoder
    C, H = int(model.cascade), int(model.grid_size)
    P, seed = C * H ** 3, 0x1234567890ABCDEF
    idx = torch.empty(P, dtype=torch.int32, device="cuda")
    xyz = torch.empty(P, 3, dtype=torch.float32, device="cuda")
    L.check(L.lib().enerf_density_grid_cells(None, C, H, float(model.bound), H ** 3 // 4, ctypes.c_uint64(seed),
                                             idx.data_ptr(), xyz.data_ptr(), L.stream_handle()), "cells")
    S = float(np.log2(enc.per_level_scale))
    aff = (float(model.bound), float(np.float32(1.0) / np.float32(2 * model.bound)))
    Pp = (P + 31) // 32 * 32
    a = torch.zeros(16, Pp, 2, device="cuda")
    gb.grid_encode_forward(xyz, enc.embeddings.detach(), enc.offsets, a, P, 3, 2, 16, S, enc.base_resolution, False, a,
                           enc.gridtype_id, layout=2, affine=aff)
    b = torch.zeros(16, Pp, 2, device="cuda")
    L.check(L.lib().enerf_grid_encode_forward_sweep(enc.embeddings.detach().data_ptr(), enc.offsets.data_ptr(),
                                                    b.data_ptr(), C, H, float(model.bound), ctypes.c_uint64(seed), 2, 16, S,
                                                    int(enc.base_resolution), int(enc.gridtype_id), 2, aff[0], aff[1],
                                                    L.stream_handle()), "sweep")
    assert torch.equal(a, b)
    states = []
    for in_kernel in (False, True):
        torch.manual_seed(3)
        m = NeRFNetwork(encoding="hashgrid", bound=2, cuda_ray=True, out_dim_color=3).cuda()
        m.local_step = 5
        m.step_counter[:5, 0] = torch.tensor([100, 200, 300, 400, 500], dtype=torch.int32)
        density_update.SWEEP_IN_KERNEL = in_kernel
        try:
            m.update_extra_state()
            m.update_extra_state()
        finally:
            density_update.SWEEP_IN_KERNEL = True
        states.append((m.density_grid.clone(), m.density_bitfield.clone(), m.mean_density, m.mean_count))
    assert torch.equal(states[0][0], states[1][0]) and torch.equal(states[0][1], states[1][1])
    assert states[0][2:] == states[1][2:]
