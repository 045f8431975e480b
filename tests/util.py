import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def det_fill_(params, seed, lo=-1.0, hi=1.0):
    """Same deterministic parameter fill as oracle/make_golden.py:det_fill_."""
    g = torch.Generator().manual_seed(seed)
    for p in params:
        p.data.copy_(torch.rand(p.shape, generator=g) * (hi - lo) + lo)


def t(a):
    return torch.from_numpy(np.asarray(a))


def assert_close(a, b, rtol=1e-5, atol=1e-6, msg=""):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=msg)


# ---- synthetic scene shared by tests, smoke() and bench.py: shell + 8 blobs -> occupancy bitfield
def synthetic_density_grid(bound, H=128):
    """density at the cell centres of every cascade, in the renderer's morton order: [cascade, H^3] float32."""
    import math
    from oracle import oracle as O
    cascade = 1 + math.ceil(math.log2(bound))
    ax = np.arange(H, dtype=np.int32)
    coords = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    idx = O.morton3D(coords).astype(np.int64)
    grid = np.zeros((cascade, H ** 3), np.float32)
    blobs = np.array([[sx, sy, sz] for sx in (-0.4, 0.4) for sy in (-0.4, 0.4) for sz in (-0.4, 0.4)], np.float32)
    for cas in range(cascade):
        b = min(2 ** cas, bound)
        hgs = b / H
        xyz = (2 * coords.astype(np.float32) / (H - 1) - 1) * (b - hgs)
        r = np.linalg.norm(xyz, axis=1)
        dens = (np.abs(r - 0.6) < 0.05).astype(np.float32) * 1.0
        for c in blobs:
            dens += np.exp(-np.sum((xyz - c) ** 2, axis=1) / (2 * 0.06 ** 2)).astype(np.float32)
        grid[cas, idx] = dens
    return grid


def camera_rays(n, seed, bound, radius=1.5):
    """n rays from cameras on a circle of `radius` looking roughly at the origin (plus a few misses)."""
    g = np.random.default_rng(seed)
    ang = g.uniform(0, 2 * np.pi, n)
    o = np.stack([radius * np.cos(ang), np.full(n, 0.3), radius * np.sin(ang)], -1)
    tgt = g.uniform(-0.8, 0.8, (n, 3))
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    if n >= 8:
        d[0] = [0.0, 1.0, 0.0]            # axis-parallel
        d[1] = -o[1] / np.linalg.norm(o[1])
        d[2] = [0.0, 0.0, 1.0] if o[2][2] > 0 else [0.0, 0.0, -1.0]   # pointing away: may miss
    return o.astype(np.float32), d.astype(np.float32)
