"""NeRFRenderer.update_extra_state on the device (nerf/renderer.py:472-560 of the reference): csrc/density_update.hip picks
the cells and applies the results, the density network runs in between as one batch over all cascades.  One read-back
(mean density + step-counter sum, 16 bytes) instead of the reference's nonzero / .item() / .item() per update."""
import ctypes

import torch

from . import _lib as L
from . import fused_network
from . import raymarching as _rm

ENABLED = True
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
def update(model, decay=0.95):
    """One update_extra_state: density grid EMA, bitfield, mean_density, mean_count, counters reset."""
    lib = L.lib()
    stream = L.stream_handle()
    dev = model.density_grid.device
    C, H = int(model.cascade), int(model.grid_size)
    full = model.iter_density < 16                                    # renderer.py:484
    N = H ** 3 // 4                                                   # renderer.py:515
    P = C * H ** 3 if full else C * 2 * N
    indices = torch.empty(P, dtype=torch.int32, device=dev)
    xyzs = torch.empty(P, 3, dtype=torch.float32, device=dev)
    seed = ((torch.initial_seed() + 1) * _GOLDEN + int(model.iter_density) * 0xD1B54A32D192ED03) & (2 ** 64 - 1)
    L.check(lib.enerf_density_grid_cells(None if full else model.density_grid.data_ptr(), C, H, float(model.bound), N,
                                         ctypes.c_uint64(seed), indices.data_ptr(), xyzs.data_ptr(), stream),
            "density_grid_cells")
    sigmas = _sigmas(model, xyzs).contiguous()
    stats = torch.empty(2, dtype=torch.float64, device=dev)
    total_step = min(16, int(model.local_step))
    L.check(lib.enerf_density_grid_update(indices.data_ptr(), sigmas.data_ptr(), P // C, C, H,
                                          float(model.density_scale * 0.003383), float(decay),
                                          float(model.density_thresh), model.density_grid.data_ptr(),
                                          model.density_bitfield.data_ptr(), model.step_counter.data_ptr(), total_step,
                                          stats.data_ptr(), stream), "density_grid_update")
    mean, counted = stats.tolist()                                    # the update's only host synchronisation
    model.mean_density = mean
    model.iter_density += 1
    if total_step > 0:
        model.mean_count = int(counted / total_step)
    model.local_step = 0


@torch.no_grad()
def mark_untrained(model, poses, intrinsic):
    """NeRFRenderer.mark_untrained_grid (nerf/renderer.py:408-469) as one launch: cells no camera sees get -1."""
    poses = torch.as_tensor(poses, dtype=torch.float32).to(model.density_grid.device).contiguous()
    fx, fy, cx, cy = (float(v) for v in intrinsic)
    L.check(L.lib().enerf_mark_untrained_grid(poses.data_ptr(), poses.shape[0], poses.shape[1] * poses.shape[2], fx, fy,
                                              cx, cy, int(model.cascade), int(model.grid_size), float(model.bound),
                                              model.density_grid.data_ptr(), L.stream_handle()),
            "mark_untrained_grid")
