"""march_rays_train / march_rays have no second statement anywhere in the reference, so the oracle is checked through
the self-consistency properties SURVEY.md 8c(7) lists: every emitted sample sits in an occupied cell and on its ray,
deltas telescope, counts respect max_steps / M, and the inference marcher iterated to exhaustion reproduces the
training marcher's sample sequence when perturb = 0."""
import math

import numpy as np
import pytest

from oracle import oracle as O
from util import synthetic_density_grid, camera_rays

H = 128


def _scene(bound):
    grid = synthetic_density_grid(bound, H)
    bits = O.packbits(grid.reshape(-1), 0.01)
    return grid, bits, 1 + math.ceil(math.log2(bound))


def _cell_bit(bits, xyz, dt, bound, C):
    mx = np.abs(xyz).max(1)
    _, e = np.frexp(mx)
    lvl_pos = np.clip(e, 0, C - 1)
    _, e2 = np.frexp(dt * H * 0.5)
    lvl = np.maximum(lvl_pos, np.clip(e2, 0, C - 1))
    mb = np.minimum(2.0 ** lvl, bound).astype(np.float32)
    n = np.clip((0.5 * (xyz / mb[:, None] + 1) * H), 0, H - 1).astype(np.int32)
    idx = lvl.astype(np.int64) * H ** 3 + O.morton3D(n).astype(np.int64)
    return (bits[idx // 8] >> (idx % 8)) & 1


@pytest.mark.parametrize("bound,perturb,dt_gamma", [(1, 0, 0.0), (2, 1, 0.0), (3, 1, 0.0), (3, 0, 1.0 / 128)])
def test_train_march_properties(bound, perturb, dt_gamma):
    grid, bits, C = _scene(bound)
    occ = np.unpackbits(bits, bitorder="little").mean()
    assert 0.001 < occ < 0.2
    o, d = camera_rays(96, 7 + bound, bound)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.2)
    N, max_steps = len(o), 1024
    M = N * max_steps
    xyzs, dirs, deltas, rays, counter = O.march_rays_train(o, d, bits, bound, dt_gamma, max_steps, C, H, M, nears, fars,
                                                           perturb)
    steps = rays[:, 2]
    assert counter[1] == N and counter[0] == steps.sum() and counter[0] > 200
    assert (rays[:, 0] == np.arange(N)).all()
    assert (rays[:, 1] == np.concatenate([[0], np.cumsum(steps)[:-1]])).all()
    assert steps.max() <= max_steps
    assert (steps[nears == np.finfo(np.float32).max] == 0).all()
    tot = counter[0]
    assert not xyzs[tot:].any() and not deltas[tot:].any()
    # every emitted sample is in an occupied cell
    assert _cell_bit(bits, xyzs[:tot], deltas[:tot, 0], bound, C).all()
    dt_min = np.float32(2 * np.float32(1.7320508075688772) / max_steps)
    for n in range(N):
        s, k = rays[n, 1], rays[n, 2]
        if k == 0:
            continue
        p, dl = xyzs[s:s + k], deltas[s:s + k]
        assert (dirs[s:s + k] == d[n]).all()
        # on the ray (inside the clamp box): cross(p - o, d) ~ 0
        inside = (np.abs(p) < bound).all(1)
        cr = np.cross(p[inside] - o[n], d[n])
        assert np.abs(cr).max() < 5e-5 if len(cr) else True
        # deltas telescope: sum(real deltas) = t_after_last - t0, t0 in [near, near + dt_min)
        t_last = np.dot(p[-1] - o[n], d[n]) + dl[-1, 0]
        t0 = t_last - dl[:, 1].astype(np.float64).sum()
        assert nears[n] - 1e-3 <= t0 <= nears[n] + (dt_min if perturb else 0) + 1e-3
        assert (dl[:, 0] >= dt_min * 0.999).all() and (dl[:, 1] >= dl[:, 0] * 0.999).all()
        assert t_last - dl[-1, 0] < fars[n]


def test_overflow_drops_rays_and_keeps_counter():
    bound = 2
    grid, bits, C = _scene(bound)
    o, d = camera_rays(64, 3, bound)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.2)
    full = O.march_rays_train(o, d, bits, bound, 0.0, 1024, C, H, 64 * 1024, nears, fars, 0)
    tot = int(full[4][0])
    M = tot // 2
    xyzs, dirs, deltas, rays, counter = O.march_rays_train(o, d, bits, bound, 0.0, 1024, C, H, M, nears, fars, 0)
    assert counter[0] == tot                       # counts requested samples incl. dropped rays
    assert (rays == full[3]).all()
    kept = (rays[:, 2] > 0) & (rays[:, 1] + rays[:, 2] < M)     # note '>=' drop rule
    assert kept.any() and (~kept & (rays[:, 2] > 0)).any()
    for n in range(64):
        s, k = rays[n, 1], rays[n, 2]
        if kept[n]:
            assert (xyzs[s:s + k] == full[0][s:s + k]).all()
        elif k and s < M:
            assert not xyzs[s:min(s + k, M)].any()


@pytest.mark.parametrize("bound", [1, 3])
def test_inference_march_reproduces_training_samples(bound):
    grid, bits, C = _scene(bound)
    o, d = camera_rays(48, 11, bound)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.2)
    N = len(o)
    xyzs, _, deltas, rays, _ = O.march_rays_train(o, d, bits, bound, 0.0, 1024, C, H, N * 1024, nears, fars, 0)
    alive = np.arange(N, dtype=np.int32)
    rt = nears.copy()
    got = [[] for _ in range(N)]
    n_step = 5
    for _ in range(400):
        n_alive = len(alive)
        if n_alive == 0:
            break
        x, dd, dl = O.march_rays(n_alive, n_step, alive, rt, o, d, bound, 0.0, 1024, C, H, bits, nears, fars,
                                 n_alive * n_step, 0)
        x = x.reshape(n_alive, n_step, 3)
        dl = dl.reshape(n_alive, n_step, 2)
        new_alive, new_t = [], []
        for j, r in enumerate(alive):
            k = int((dl[j, :, 0] > 0).sum())
            got[r].extend(x[j, :k])
            if k == n_step:       # ray continues: new t = t + sum(real deltas)  (composite_rays does this)
                new_alive.append(r)
                new_t.append(np.float32(rt[j] + np.float32(0)))
                t = np.float32(rt[j])
                for s in range(k):
                    t = np.float32(t + dl[j, s, 1])
                new_t[-1] = t
        alive, rt = np.array(new_alive, np.int32), np.array(new_t, np.float32)
    for n in range(N):
        s, k = rays[n, 1], rays[n, 2]
        a = np.array(got[n]).reshape(-1, 3)
        assert len(a) == k, (n, len(a), k)
        if k:
            np.testing.assert_allclose(a, xyzs[s:s + k], atol=2e-5)
