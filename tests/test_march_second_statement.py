"""The C oracle's march_rays_train against an independent pure-Python / float32 restatement written from the kernel's
description (oracle/march_second.py): bit-exact positions, step sizes, real delta-t, ray table and counters.  This is
the pin the marcher gets in place of golden vectors (the reference has none, and its CUDA cannot be built here)."""
import math

import numpy as np
import pytest

from oracle import march_second as S
from oracle import oracle as O
from util import camera_rays, synthetic_density_grid

H = 128


@pytest.mark.parametrize("bound,perturb,dt_gamma,n", [(1, 0, 0.0, 24), (2, 1, 0.0, 24), (3, 1, 1.0 / 128, 20),
                                                      (2, 0, 1.0 / 256, 16)])
def test_c_oracle_matches_the_second_statement(bound, perturb, dt_gamma, n):
    grid = synthetic_density_grid(bound, H)
    bits = O.packbits(grid.reshape(-1), 0.01)
    C = 1 + math.ceil(math.log2(bound))
    o, d = camera_rays(n, 100 + bound, bound)
    d[0] = (0.0, 0.0, -1.0) if o[0][2] > 0 else (0.0, 0.0, 1.0)           # an axis-parallel ray: 1/d = +-inf
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.2)
    M = n * 1024
    ref = O.march_rays_train(o, d, bits, bound, dt_gamma, 1024, C, H, M, nears, fars, perturb)
    got = S.march_rays_train(o, d, bits, bound, dt_gamma, 1024, C, H, M, nears, fars, perturb)
    assert int(ref[4][0]) > 200                                            # the rays do meet the scene
    for a, b, name in zip(got, ref, ("xyzs", "dirs", "deltas", "rays", "counter")):
        assert np.array_equal(np.asarray(a), np.asarray(b)), name


def test_second_statement_overflow_rule():
    bound = 2
    grid = synthetic_density_grid(bound, H)
    bits = O.packbits(grid.reshape(-1), 0.01)
    o, d = camera_rays(12, 7, bound)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.2)
    tot = int(O.march_rays_train(o, d, bits, bound, 0.0, 1024, 2, H, 12 * 1024, nears, fars, 1)[4][0])
    M = tot // 2
    ref = O.march_rays_train(o, d, bits, bound, 0.0, 1024, 2, H, M, nears, fars, 1)
    got = S.march_rays_train(o, d, bits, bound, 0.0, 1024, 2, H, M, nears, fars, 1)
    for a, b in zip(got, ref):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    assert int(got[4][0]) == tot                                           # dropped rays still count


@pytest.mark.parametrize("perturb,dt_gamma", [(0, 0.0), (5, 0.0), (3, 1.0 / 256)])
def test_inference_march_matches_the_second_statement(perturb, dt_gamma):
    bound = 2
    grid = synthetic_density_grid(bound, H)
    bits = O.packbits(grid.reshape(-1), 0.01)
    o, d = camera_rays(40, 19, bound)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.2)
    rng = np.random.default_rng(2)
    alive = rng.permutation(40).astype(np.int32)[:24]
    rt = (nears[alive] + rng.random(24).astype(np.float32) * 1.5).astype(np.float32)
    n_step = 6
    M = 24 * n_step
    ref = O.march_rays(24, n_step, alive, rt, o, d, bound, dt_gamma, 1024, 2, H, bits, nears, fars, M, perturb)
    got = S.march_rays(24, n_step, alive, rt, o, d, bits, bound, dt_gamma, 1024, 2, H, fars, M, perturb)
    assert (ref[2][:, 0] != 0).sum() > 40
    for a, b, name in zip(got, ref, ("xyzs", "dirs", "deltas")):
        assert np.array_equal(np.asarray(a), np.asarray(b)), name
