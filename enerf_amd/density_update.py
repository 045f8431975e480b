"""NeRFRenderer.update_extra_state on the device (nerf/renderer.py:472-560 of the reference): csrc/density_update.hip picks
the cells and applies the results, the density network runs in between as one batch over all cascades.  One read-back
(mean density + step-counter sum, 16 bytes) instead of the reference's nonzero / .item() / .item() per update."""
import ctypes

import torch

from . import _lib as L
from . import fused_network
from . import raymarching as _rm

ENABLED = True
SWEEP_IN_KERNEL = True     # full sweeps: query points generated inside the grid kernel (False: written out first)
_GOLDEN = 0x9E3779B97F4A7C15
_CHUNK = 1 << 21          # density() batch for networks outside the fused fp32 path


def supported(model):
    g = getattr(model, "density_grid", None)
    H = int(getattr(model, "grid_size", 0))
    return (ENABLED and model.cuda_ray and g is not None and g.is_cuda and g.dtype == torch.float32
            and g.is_contiguous() and _rm._DEVICE == "cuda" and 16 <= H <= 512 and H & (H - 1) == 0
            and 1 <= int(model.cascade) <= 8 and model.density_bitfield.is_contiguous())


def _sigmas(model, xyzs):
    if fused_network.supported(model, xyzs, xyzs):
        return fused_network.density_sigma(model, xyzs)               # sigma only: no geo_feat written
    out = torch.empty(xyzs.shape[0], dtype=torch.float32, device=xyzs.device)
    for a in range(0, xyzs.shape[0], _CHUNK):
        out[a:a + _CHUNK] = model.density(xyzs[a:a + _CHUNK])["sigma"].reshape(-1).detach().float()
    return out


@torch.no_grad()
def update(model, decay=0.95, split=False):
    """One update_extra_state: density grid EMA, bitfield, mean_density, mean_count, counters reset."""
    resolve_pending(model)                       # (an earlier update's mean_density, long since on the host)
    lib = L.lib()
    stream = L.stream_handle()
    dev = model.density_grid.device
    C, H = int(model.cascade), int(model.grid_size)
    full = model.iter_density < 16                                    # renderer.py:484
    N = H ** 3 // 4                                                   # renderer.py:515
    P = C * H ** 3 if full else C * 2 * N
    seed = ((torch.initial_seed() + 1) * _GOLDEN + int(model.iter_density) * 0xD1B54A32D192ED03) & (2 ** 64 - 1)
    probe = model.density_grid.new_empty(1, 3)
    if full and SWEEP_IN_KERNEL and fused_network.supported(model, probe, probe):
        # full sweep through the fused fp32 network: the grid kernel generates the query points itself (the same points
        # enerf_density_grid_cells would write: csrc/sweep_points.h), the update takes them in the sweep's own order
        indices = None
        sigmas = fused_network.density_sigma_sweep(model, C, H, seed)
    else:
        indices = fused_network._density_scratch(model, "indices", (P,), torch.int32, dev)
        xyzs = fused_network._density_scratch(model, "xyzs", (P, 3), torch.float32, dev)
        L.check(lib.enerf_density_grid_cells(None if full else model.density_grid.data_ptr(), C, H, float(model.bound), N,
                                             ctypes.c_uint64(seed), indices.data_ptr(), xyzs.data_ptr(), stream),
                "density_grid_cells")
        sigmas = _sigmas(model, xyzs).contiguous()
    stats = torch.empty(2, dtype=torch.float64, device=dev)
    total_step = min(16, int(model.local_step))
    model.local_step = 0                       # renders queued from here on take slots 0.. of the restarted ring
    L.check(lib.enerf_density_grid_update(None if indices is None else indices.data_ptr(), sigmas.data_ptr(), P // C, C, H,
                                          float(model.density_scale * 0.003383), float(decay),
                                          float(model.density_thresh), model.density_grid.data_ptr(),
                                          model.density_bitfield.data_ptr(), model.step_counter.data_ptr(), total_step,
                                          stats.data_ptr(), stream), "density_grid_update")
    _rm.BITFIELD_EPOCH[0] += 1                                        # the bitfield was rewritten behind torch's back
    model.iter_density += 1
    if not split:
        _finish(model, stats.tolist(), total_step)                    # the update's only host synchronisation
        return None
    # split: the 16 bytes travel to pinned memory behind the update's kernels; the caller queues whatever does not need
    # them (the step's near/far and march count pass read the new bitfield, not the budget) and then calls update_end
    host = torch.empty(2, dtype=torch.float64, pin_memory=True)
    host.copy_(stats, non_blocking=True)
    done = torch.cuda.Event()
    done.record()
    return (done, host, stats, total_step)


def _finish(model, values, total_step):
    mean, counted = values
    model.mean_density = mean
    if total_step > 0:
        model.mean_count = int(counted / total_step)


def update_begin(model, decay=0.95):
    """update() up to its read-back: everything is queued, nothing is waited for.  -> handle for update_end."""
    return update(model, decay, split=True)


def update_end(model, handle, early=None):
    """Wait for the update's 16 bytes and set mean_density / mean_count.  `early` = (mean_count or None, total_step) from
    fused_render.early_mean_count, taken before the update was queued: the budget is set from it and NOTHING is waited
    for -- mean_density (read on the host by checkpoints only) is filled in when the copy has landed (resolve_pending)."""
    done, host, _stats, total_step = handle
    if early is not None and early[1] == total_step:
        if early[0] is not None:
            model.mean_count = early[0]
        model._pending_density_stats = (done, host, _stats, total_step, early[0])
        return
    done.synchronize()
    _finish(model, host.tolist(), total_step)


def resolve_pending(model, wait=True):
    """mean_density of the last update whose read-back was left in flight (update_end(early=...))."""
    pend = getattr(model, "_pending_density_stats", None)
    if pend is None:
        return
    done, host, _stats, total_step, early_count = pend
    if not wait and not done.query():
        return
    done.synchronize()
    mean, counted = host.tolist()
    model.mean_density = mean
    model._pending_density_stats = None
    if total_step > 0 and early_count is not None and int(counted / total_step) != early_count:
        # the update kernel's own sum of the ring is the reference value: it must be what was read early (nothing may
        # march between the early read and the update -- fused_render.early_mean_count's conditions)
        raise RuntimeError(f"update_extra_state: the early sample budget {early_count} is not the update's own "
                           f"{int(counted / total_step)} (counted {counted} over {total_step} renders)")


@torch.no_grad()
def mark_untrained(model, poses, intrinsic):
    """NeRFRenderer.mark_untrained_grid (nerf/renderer.py:408-469) as one launch: cells no camera sees get -1."""
    poses = torch.as_tensor(poses, dtype=torch.float32).to(model.density_grid.device).contiguous()
    fx, fy, cx, cy = (float(v) for v in intrinsic)
    L.check(L.lib().enerf_mark_untrained_grid(poses.data_ptr(), poses.shape[0], poses.shape[1] * poses.shape[2], fx, fy,
                                              cx, cy, int(model.cascade), int(model.grid_size), float(model.bound),
                                              model.density_grid.data_ptr(), L.stream_handle()),
            "mark_untrained_grid")


# ---------------------------------------------------------------------------------------------------------------------
# The same two maintenance passes as plain tensor programs over the `raymarching` wrappers.  They serve whatever the
# device-side passes above do not (a backend without them, e.g. the host-side test backend; ENABLED = False) and are the
# comparator of tests/test_gpu_density_update.py.  Semantics: SURVEY.md 3.4 and Appendix A ("Renderer/network glue").

def _all_cells(H, dev):
    """Integer coordinates of every grid cell in the reference's order (nerf/renderer.py:489-496: meshgrid 'ij' over x, y,
    z, flattened -- z fastest) and their Morton indices.  The order decides which draw of torch's random stream jitters
    which cell: with it, this route reproduces the reference's update_extra_state bit for bit on the same stream
    (tests/test_host_cuda_ray_vs_reference.py)."""
    ax = torch.arange(H, dtype=torch.int32, device=dev)
    xx, yy, zz = torch.meshgrid(ax, ax, ax, indexing="ij")
    coords = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], dim=-1).contiguous()
    return coords, _rm.morton3D(coords).long()


def _jittered_density(model, coords, cas):
    """density * density_scale * dt_min at one uniformly drawn point inside each of the given cells of cascade `cas`."""
    H = model.grid_size
    extent = min(2 ** cas, model.bound)
    half_cell = extent / H
    pts = (2 * coords.float() / (H - 1) - 1) * (extent - half_cell)
    pts += (torch.rand_like(pts) * 2 - 1) * half_cell                 # (in place, as renderer.py:507 has it)
    return _sigmas(model, pts) * (model.density_scale * 0.003383)


@torch.no_grad()
def update_torch(model, decay=0.95):
    """update_extra_state op by op, in the reference's order of cells, cascades and random draws (renderer.py:472-560; H =
    128 is one block of its S = 128 split).  First 16 calls: every cell of every cascade is re-evaluated; afterwards, per
    cascade, H^3/4 uniformly drawn cells followed by H^3/4 draws (with replacement) from the cells whose density is
    positive.  Then: grid <- max(grid * decay, new) where both are valid (>= 0; -1 marks untrained cells), mean of the
    clamped grid, bitfield at min(mean, density_thresh), sample budget from the step counters."""
    H, C = model.grid_size, model.cascade
    dev = model.density_grid.device
    fresh = torch.full_like(model.density_grid, -1.0)
    if model.iter_density < 16:
        coords, cells = _all_cells(H, dev)
        for cas in range(C):
            fresh[cas, cells] = _jittered_density(model, coords, cas)
    else:
        n = H ** 3 // 4
        for cas in range(C):
            coords = torch.randint(0, H, (n, 3), device=dev)
            uniform = _rm.morton3D(coords).long()
            occupied = torch.nonzero(model.density_grid[cas] > 0).squeeze(-1)
            drawn = occupied[torch.randint(0, occupied.shape[0], [n], dtype=torch.long, device=dev)]
            cells = torch.cat([uniform, drawn], dim=0)
            coords = torch.cat([coords, _rm.morton3D_invert(drawn)], dim=0)
            fresh[cas, cells] = _jittered_density(model, coords, cas)

    both = (model.density_grid >= 0) & (fresh >= 0)
    model.density_grid[both] = torch.maximum(model.density_grid[both] * decay, fresh[both])
    model.mean_density = torch.mean(model.density_grid.clamp(min=0)).item()
    model.iter_density += 1
    model.density_bitfield = _rm.packbits(model.density_grid, min(model.mean_density, model.density_thresh),
                                          model.density_bitfield)
    windows = min(16, model.local_step)
    if windows > 0:
        model.mean_count = int(model.step_counter[:windows, 0].sum().item() / windows)
    model.local_step = 0


@torch.no_grad()
def mark_untrained_torch(model, poses, intrinsic, cam_chunk=64, cell_chunk=1 << 18):
    """mark_untrained_grid op by op: a cell of a cascade stays trainable when at least one camera has its centre in
    front of it (z > 0) and inside the image frustum widened by one cell (|x| < cx/fx * z + 2 * half_cell, same in y);
    every other cell gets density -1."""
    H, C = model.grid_size, model.cascade
    dev = model.density_grid.device
    poses = torch.as_tensor(poses, dtype=torch.float32).to(dev)
    fx, fy, cx, cy = intrinsic
    coords, cells = _all_cells(H, dev)
    unit = 2 * coords.float() / (H - 1) - 1
    seen = torch.zeros_like(model.density_grid, dtype=torch.bool)
    for cas in range(C):
        extent = min(2 ** cas, model.bound)
        half_cell = extent / H
        for a in range(0, cells.shape[0], cell_chunk):
            world = unit[a:a + cell_chunk] * (extent - half_cell)                              # [P,3]
            hit = torch.zeros(world.shape[0], dtype=torch.bool, device=dev)
            for b in range(0, poses.shape[0], cam_chunk):
                R, t = poses[b:b + cam_chunk, :3, :3], poses[b:b + cam_chunk, :3, 3]
                cam = (world.unsqueeze(0) - t.unsqueeze(1)) @ R                                # [B,P,3] camera frame
                x, y, z = cam.unbind(-1)
                inside = (z > 0) & (x.abs() < cx / fx * z + 2 * half_cell) & (y.abs() < cy / fy * z + 2 * half_cell)
                hit |= inside.any(0)
            seen[cas, cells[a:a + cell_chunk]] = hit
    model.density_grid[~seen] = -1
