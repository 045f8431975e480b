"""CPU restatement (numpy, TEST INFRASTRUCTURE ONLY) of the deterministic parts of NeRFRenderer.update_extra_state
(nerf/renderer.py:472-560 of the reference):

  cell_centres       :496-503, :528-533   xyzs = 2 * coords / (grid_size - 1) - 1, scaled by (bound_c - half_grid_size);
                                          the jitter added on top is uniform in [-half_grid_size, half_grid_size]
  apply_update       :538-553             tmp_grid scatter, EMA max with decay, mean of clamp(grid, 0), packbits against
                                          min(mean, density_thresh)
  mean_count         :555-558             int(sum(step_counter[:total_step, 0]) / total_step)
  untrained_cells    :408-469             mark_untrained_grid: cells no camera frustum (widened by two half-cells) sees

Parity status: these functions transcribe the method's tensor expressions; they are not themselves run against the
reference.  What IS (round 4): the reference's own NeRFRenderer.update_extra_state / mark_untrained_grid, executed on CPU
over this oracle (oracle/make_golden.py gold_cuda_ray -> tests/golden/ref_cuda_ray.npz), pin enerf_amd's plain-tensor
route (density_update.update_torch / mark_untrained_torch) bit for bit on the same torch random stream
(tests/test_host_cuda_ray_vs_reference.py); the device-side passes are compared with that route and with these functions
on the GPU (tests/test_gpu_density_update.py): which cells a device-side update draws comes from its own counter-based
generator and is checked in distribution.
"""
import numpy as np

from . import oracle as O


def cascade_geometry(cas, bound, grid_size):
    b = min(2 ** cas, bound)                                  # :498
    half = b / grid_size                                      # :499
    return b - half, half


def cell_centres(indices, cas, bound, grid_size):
    """Unjittered query positions of Morton cells `indices` of cascade `cas` -> [n,3] float32."""
    coords = O.morton3D_invert(np.asarray(indices, np.int32)).astype(np.float32)
    xyzs = np.float32(2) * coords / np.float32(grid_size - 1) - np.float32(1)      # :496
    span, _ = cascade_geometry(cas, bound, grid_size)
    return (xyzs * np.float32(span)).astype(np.float32)                            # :501


def apply_update(density_grid, indices, sigmas, sigma_scale, decay, density_thresh):
    """density_grid [C,H^3] f32, indices [C,n] (unique per cascade, or the last writer wins), sigmas [C,n]
    -> (new grid, mean, bitfield)."""
    grid = np.array(density_grid, np.float32, copy=True)
    tmp = -np.ones_like(grid)                                                      # :480
    for cas in range(grid.shape[0]):
        tmp[cas, np.asarray(indices[cas], np.int64)] = np.asarray(sigmas[cas], np.float32) * np.float32(sigma_scale)
    valid = (grid >= 0) & (tmp >= 0)                                               # :541
    grid[valid] = np.maximum(grid[valid] * np.float32(decay), tmp[valid])          # :542
    mean = float(np.mean(np.clip(grid, 0, None), dtype=np.float64))                # :543
    thresh = min(mean, density_thresh)                                             # :547
    return grid, mean, O.packbits(grid.reshape(-1), np.float32(thresh))            # :548


def mean_count(step_counter, local_step):
    total_step = min(16, local_step)                                               # :555
    if total_step <= 0:
        return None
    return int(int(np.asarray(step_counter)[:total_step, 0].sum()) / total_step)   # :557


def untrained_cells(poses, intrinsic, cascade, bound, grid_size):
    """-> bool [cascade, H^3] (Morton order): True where mark_untrained_grid writes -1 (:408-469)."""
    poses = np.asarray(poses, np.float32)
    fx, fy, cx, cy = intrinsic
    H3 = grid_size ** 3
    idx = np.arange(H3, dtype=np.int32)
    out = np.zeros((cascade, H3), bool)
    for cas in range(cascade):
        span, half = cascade_geometry(cas, bound, grid_size)
        world = cell_centres(idx, cas, bound, grid_size)                                      # :440-446
        count = np.zeros(H3, np.int64)
        for pose in poses:
            cam = (world - pose[:3, 3].astype(np.float32)) @ pose[:3, :3].astype(np.float32)   # :454-455
            mask = (cam[:, 2] > 0) & (np.abs(cam[:, 0]) < np.float32(cx / fx) * cam[:, 2] + np.float32(half * 2)) \
                & (np.abs(cam[:, 1]) < np.float32(cy / fy) * cam[:, 2] + np.float32(half * 2))  # :458-461
            count += mask
        out[cas] = count == 0                                                                  # :468
    return out
