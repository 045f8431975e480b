"""Deterministic synthetic scene replacing the (absent) E-NeRF datasets: SURVEY.md 8(d).

Analytic density = sphere shell | |x| - 0.6 | < 0.05  union  8 Gaussian blobs at (+-0.4)^3, sampled at the cell
centres of every cascade -> density_grid [cascade, 128^3] (morton order) -> packbits(thresh 0.01).
Cameras: pinhole 640x480 (fx = fy = 320, cx = 320, cy = 240) on a circle of radius 1.5 at height 0.3 looking at the
origin.  Everything is computed with torch + the `raymarching` wrappers, i.e. on the device the backend runs on.
"""
import math

import numpy as np
import torch

from . import raymarching

W, H = 640, 480
INTRINSICS = (320.0, 320.0, 320.0, 240.0)


def analytic_density(xyz):
    r = xyz.norm(dim=-1)
    dens = ((r - 0.6).abs() < 0.05).float()
    for sx in (-0.4, 0.4):
        for sy in (-0.4, 0.4):
            for sz in (-0.4, 0.4):
                c = torch.tensor([sx, sy, sz], dtype=xyz.dtype, device=xyz.device)
                dens = dens + torch.exp(-((xyz - c) ** 2).sum(-1) / (2 * 0.06 ** 2))
    return dens


def analytic_color(xyz):
    return 0.5 + 0.5 * torch.sin(xyz * torch.tensor([3.0, 5.0, 7.0], device=xyz.device))


def density_grid(bound, device, grid_size=128):
    """[cascade, grid_size^3] float32 in the renderer's morton order."""
    cascade = 1 + math.ceil(math.log2(bound))
    ax = torch.arange(grid_size, dtype=torch.int32, device=device)
    xx, yy, zz = torch.meshgrid(ax, ax, ax, indexing="ij")
    coords = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], dim=-1).contiguous()
    indices = raymarching.morton3D(coords).long()
    grid = torch.zeros(cascade, grid_size ** 3, dtype=torch.float32, device=device)
    unit = 2 * coords.float() / (grid_size - 1) - 1
    for cas in range(cascade):
        b = min(2 ** cas, bound)
        hgs = b / grid_size
        grid[cas, indices] = analytic_density(unit * (b - hgs))
    return grid


def install_occupancy(model, thresh=0.01):
    """Give a cuda_ray model the synthetic occupancy (density_grid + bitfield); returns copies for later restore."""
    dev = model.density_grid.device
    g = density_grid(model.bound, dev, model.grid_size)
    model.density_grid.copy_(g)
    model.density_bitfield = raymarching.packbits(model.density_grid, thresh, model.density_bitfield)
    return model.density_grid.clone(), model.density_bitfield.clone()


def pose(k, n_poses=32, radius=1.5, height=0.3):
    """cam2world [4,4] (x right, y down, z forward), camera k of n on the circle, looking at the origin."""
    ang = 2 * math.pi * k / n_poses
    eye = np.array([radius * math.cos(ang), height, radius * math.sin(ang)])
    fwd = -eye / np.linalg.norm(eye)
    up = np.array([0.0, 1.0, 0.0])
    right = np.cross(up, fwd)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    m = np.eye(4, dtype=np.float32)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, down, fwd, eye
    return torch.from_numpy(m)


def pixel_rays(c2w, inds, device):
    """rays through pixel indices `inds` (row-major over HxW) of camera c2w: [1,N,3] origins and unit directions."""
    fx, fy, cx, cy = INTRINSICS
    c2w = c2w.to(device)
    i = (inds % W).float()
    j = (inds // W).float()
    dirs = torch.stack([(i - cx) / fx, (j - cy) / fy, torch.ones_like(i)], dim=-1)
    dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    rays_d = dirs @ c2w[:3, :3].t()
    rays_o = c2w[:3, 3].expand_as(rays_d)
    return rays_o[None].contiguous(), rays_d[None].contiguous()


def training_batch(step, n_rays, device, generator=None, rank=0, delta_deg=0.0):
    """Random pixels of one pose per step (nerf/utils.py:138); `delta_deg` rotates the pose for event pairs."""
    g = generator
    inds = torch.randint(0, H * W, (n_rays,), device=device, generator=g)
    k = (step * 7 + rank * 3) % 32
    c2w = pose(k + delta_deg / (360.0 / 32))
    return pixel_rays(c2w, inds, device), inds
