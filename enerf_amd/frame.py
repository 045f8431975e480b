"""Inference rendering of a ray batch (the evaluation branch of NeRFRenderer.run_cuda).

Two schedules, same image:

`render_rounds`   the reference's loop (nerf/renderer.py:364-401): alive rays take `n_step` sample slots per round
                  (march_rays), the network runs over all slots, composite_rays accumulates and marks finished rays,
                  compact_rays drops them, and the host reads the survivor count back to size the next round.
                  Needed when the marcher's jitter is on (it is re-drawn per round) or the step grows with t.

`render_frame`    MI355X-first: with 288 GB of HBM there is no reason to ration sample slots.  Every ray's samples are
                  marched contiguously in one pass (the training marcher: count -> scan -> write, offsets from a
                  deterministic scan, one 8-byte read-back to size the buffers), the network runs over real samples
                  only -- the rounds evaluate ~40 % padding slots on a 640x480 frame -- in a few large chunks, and one
                  compositing pass applies the rounds' arithmetic and termination rule per ray
                  (enerf_composite_rays_frame).  6 + 2 x chunks launches and one host synchronisation per frame instead
                  of ~9 launches and one synchronisation per round.  What it gives up: samples behind the point where a
                  ray's transmittance drops below 1e-5 are evaluated and then ignored (the rounds stop at most n_step - 1
                  samples late).  Bit-identical image as long as no ray reaches the loop's 1024-step cap.
"""
import torch

from . import raymarching
from .backends import _raymarching as _rb

FRAME_ENABLED = True
CHUNK = 1 << 22          # samples per network launch in render_frame (activations of a chunk stay cache-sized)


def frame_supported(model, rays_o, perturb, dt_gamma, bg_color):
    if not (FRAME_ENABLED and raymarching._DEVICE == "cuda" and rays_o.is_cuda and rays_o.dtype == torch.float32
            and not perturb and dt_gamma == 0 and model.bg_radius <= 0 and not torch.is_grad_enabled()):
        return False
    if isinstance(bg_color, torch.Tensor):
        return (bg_color.is_cuda and bg_color.dtype == torch.float32 and bg_color.is_contiguous()
                and bg_color.numel() in (1, 3, 3 * rays_o.shape[0]))
    return True


def render_frame(model, rays_o, rays_d, bg_color=1, max_steps=1024, trace=None):
    """rays_o, rays_d [N,3] fp32 on the device -> (depth [N], image [N,3]).  `trace(dict)` (tests) receives every
    intermediate buffer."""
    N, dev = rays_o.shape[0], rays_o.device
    f32 = dict(dtype=torch.float32, device=dev)
    nears, fars = torch.empty(N, **f32), torch.empty(N, **f32)
    _rb.near_far_from_aabb(rays_o, rays_d, model.aabb_infer, N, model.min_near, nears, fars)
    rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    geom = (rays_o, rays_d, model.density_bitfield, model.bound, 0.0, max_steps, N, model.cascade, model.grid_size)
    from .fused_render import occupied_box_flag
    _rb.march_rays_train_count(*geom, nears, fars, rays, counter, 0, occupied_box_flag(model))
    total = int(counter[0].item())                      # the frame's one synchronisation: sizes the sample buffers
    M = total + 128 - total % 128
    xyzs, dirs, deltas = torch.empty(M, 3, **f32), torch.empty(M, 3, **f32), torch.empty(M, 2, **f32)
    _rb.march_rays_train_write(*geom, M, nears, fars, xyzs, dirs, deltas, rays, counter, 0, 1)
    sigmas, rgbs = torch.empty(M, **f32), torch.empty(M, 3, **f32)
    scale = float(model.density_scale)
    into = getattr(model, "forward_into", None)
    for a in range(0, M, CHUNK):
        if into is not None:                            # the network writes its slice of the frame's arrays itself
            into(xyzs[a:a + CHUNK], dirs[a:a + CHUNK], sigmas[a:a + CHUNK], rgbs[a:a + CHUNK])
        else:
            s, c = model(xyzs[a:a + CHUNK], dirs[a:a + CHUNK])
            sigmas[a:a + CHUNK] = s
            rgbs[a:a + CHUNK] = c
    if scale != 1.0:
        sigmas *= scale
    weights_sum, depth, image = torch.empty(N, **f32), torch.empty(N, **f32), torch.empty(N, 3, **f32)
    used = torch.zeros(1, dtype=torch.int32, device=dev) if trace is not None else None
    _rb.composite_rays_frame(sigmas, rgbs, deltas, rays, N, M, nears, fars, bg_color, weights_sum, depth, image, used)
    _rb.STATS["infer_samples"] += total
    _rb.STATS["infer_calls"] += 1
    if trace is not None:
        trace(dict(nears=nears, fars=fars, rays=rays, counter=counter, xyzs=xyzs, dirs=dirs, deltas=deltas,
                   sigmas=sigmas, rgbs=rgbs, weights_sum=weights_sum, used=used, M=M))
    return depth, image


def render_rounds(model, rays_o, rays_d, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024):
    """The reference's round schedule.  `model.infer_batch_mult` = K widens a round to K x N slots (up to 8 K per ray):
    fewer rounds, hence fewer count read-backs; the per-ray sample sequence does not depend on the slotting."""
    N, dev = rays_o.shape[0], rays_o.device
    nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, model.aabb_infer, model.min_near)
    acc = dict(dtype=torch.float32, device=dev)
    weights_sum, depth, image = torch.zeros(N, **acc), torch.zeros(N, **acc), torch.zeros(N, 3, **acc)
    alive = torch.zeros(2, N, dtype=torch.int32, device=dev)        # ping-pong: compaction reads one row, writes the other
    t_now = torch.zeros(2, N, **acc)
    torch.arange(N, out=alive[0])
    t_now[0] = nears
    survivors = torch.zeros(1, dtype=torch.int32, device=dev)
    widen = max(int(getattr(model, "infer_batch_mult", 1)), 1)
    n_alive, taken, cur = N, 0, 0
    # every round marches the same bitfield: the library's box of the occupied cells (refreshed here if the bitfield
    # changed) is vouched for once, and each round's walk stops at its far side
    use_box = raymarching._DEVICE == "cuda" and rays_o.is_cuda
    if use_box:
        from . import _lib as L
        from .fused_render import occupied_box_flag
        use_box = bool(occupied_box_flag(model))
        if use_box:
            L.lib().enerf_march_rays_use_box(1)
    try:
        return _rounds_loop(model, rays_o, rays_d, bg_color, perturb, dt_gamma, max_steps, nears, fars, weights_sum, depth,
                            image, alive, t_now, survivors, widen, N)
    finally:
        if use_box:
            L.lib().enerf_march_rays_use_box(0)


def _rounds_loop(model, rays_o, rays_d, bg_color, perturb, dt_gamma, max_steps, nears, fars, weights_sum, depth, image,
                 alive, t_now, survivors, widen, N):
    n_alive, taken, cur = N, 0, 0
    while taken < 1024 and n_alive > 0:
        n_step = max(min(widen * N // n_alive, 8 * widen), 1)
        xyzs, dirs, deltas = raymarching.march_rays(n_alive, n_step, alive[cur], t_now[cur], rays_o, rays_d, model.bound,
                                                    model.density_bitfield, model.cascade, model.grid_size, nears, fars,
                                                    128, perturb, dt_gamma, max_steps)
        sigmas, rgbs = model(xyzs, dirs)
        raymarching.composite_rays(n_alive, n_step, alive[cur], t_now[cur], model.density_scale * sigmas, rgbs, deltas,
                                   weights_sum, depth, image)
        taken += n_step
        if taken >= 1024:
            break
        survivors.zero_()
        raymarching.compact_rays(n_alive, alive[1 - cur], alive[cur], t_now[1 - cur], t_now[cur], survivors)
        n_alive = int(survivors.item())
        cur = 1 - cur
    image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
    depth = torch.clamp(depth - nears, min=0) / (fars - nears)
    return depth, image
