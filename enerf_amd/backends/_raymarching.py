"""`_raymarching`: same 11 functions as raymarching/src/bindings.cpp:5-20 of the reference, same positional
arguments, tensors pre-allocated by the caller; every call forwards to libenerf_hip.so on torch's current stream."""
from .. import _lib as L

# inference samples marched since the last reset (bench.py: render Msamples/s)
STATS = {"infer_samples": 0, "infer_calls": 0}


def _f32(t, name):
    import torch
    L.check_cuda(t, name)
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be a float32 tensor (raymarching wrappers cast_inputs=torch.float32)")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")
    return t.data_ptr()


def _i32(t, name):
    L.check_cuda(t, name)
    L.check_int(t, name)
    L.check_contiguous(t, name)
    return t.data_ptr()


def _u8(t, name):
    import torch
    L.check_cuda(t, name)
    if t.dtype != torch.uint8:
        raise RuntimeError(f"{name} must be a uint8 tensor")
    L.check_contiguous(t, name)
    return t.data_ptr()


def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
    L.check(L.lib().enerf_near_far_from_aabb(_f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), _f32(aabb, "aabb"),
                                             int(N), float(min_near), _f32(nears, "nears"), _f32(fars, "fars"),
                                             L.stream_handle()), "near_far_from_aabb")


def polar_from_ray(rays_o, rays_d, radius, N, coords):
    L.check(L.lib().enerf_polar_from_ray(_f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), float(radius), int(N),
                                         _f32(coords, "coords"), L.stream_handle()), "polar_from_ray")


def morton3D(coords, N, indices):
    L.check(L.lib().enerf_morton3D(_i32(coords, "coords"), int(N), _i32(indices, "indices"), L.stream_handle()),
            "morton3D")


def morton3D_invert(indices, N, coords):
    L.check(L.lib().enerf_morton3D_invert(_i32(indices, "indices"), int(N), _i32(coords, "coords"),
                                          L.stream_handle()), "morton3D_invert")


def packbits(grid, N, density_thresh, bitfield):
    L.check(L.lib().enerf_packbits(_f32(grid, "grid"), int(N), float(density_thresh), _u8(bitfield, "bitfield"),
                                   L.stream_handle()), "packbits")


def march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas,
                     rays, counter, perturb):
    L.check(L.lib().enerf_march_rays_train(_f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), _u8(grid, "grid"),
                                           float(bound), float(dt_gamma), int(max_steps), int(N), int(C), int(H),
                                           int(M), _f32(nears, "nears"), _f32(fars, "fars"), _f32(xyzs, "xyzs"),
                                           _f32(dirs, "dirs"), _f32(deltas, "deltas"), _i32(rays, "rays"),
                                           _i32(counter, "counter"), int(perturb), L.stream_handle()),
            "march_rays_train")


def march_rays_train_ex(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas,
                        rays, counter, perturb, zero_unwritten):
    """march_rays_train into possibly uninitialised xyzs / dirs / deltas (include/enerf_hip.h)."""
    L.check(L.lib().enerf_march_rays_train_ex(_f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), _u8(grid, "grid"),
                                              float(bound), float(dt_gamma), int(max_steps), int(N), int(C), int(H),
                                              int(M), _f32(nears, "nears"), _f32(fars, "fars"), _f32(xyzs, "xyzs"),
                                              _f32(dirs, "dirs"), _f32(deltas, "deltas"), _i32(rays, "rays"),
                                              _i32(counter, "counter"), int(perturb), int(zero_unwritten),
                                              L.stream_handle()), "march_rays_train_ex")


def march_fuse_near_far(aabb, min_near):
    """Arm the next march_rays_train(_ex / _count) call to compute near / far itself and write them into the nears / fars
    arrays it is given (include/enerf_hip.h)."""
    L.check(L.lib().enerf_march_fuse_near_far(_f32(aabb, "aabb"), float(min_near)), "march_fuse_near_far")


def march_mirror_count(host_counter):
    """Arm the next march_rays_train(_ex / _count) call to write its two counter words into `host_counter` as well: a
    pinned int32 host tensor of 2 elements (include/enerf_hip.h)."""
    import torch
    if host_counter.is_cuda or not host_counter.is_pinned() or host_counter.dtype != torch.int32 or host_counter.numel() < 2:
        raise ValueError("march_mirror_count: a pinned int32 host tensor of two elements")
    L.check(L.lib().enerf_march_mirror_count(host_counter.data_ptr()), "march_mirror_count")


def occupied_box_update(grid, C, H, bound):
    """(Re)compute the library's bounding box of the occupied cells of `grid` (include/enerf_hip.h)."""
    L.check(L.lib().enerf_occupied_box_update(_u8(grid, "grid"), int(C), int(H), float(bound), L.stream_handle()),
            "occupied_box_update")


def march_rays_train_count(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, nears, fars, rays, counter, perturb,
                           flags=0):
    """Count + scan half of march_rays_train: fills rays / counter, writes no samples (include/enerf_hip.h)."""
    L.check(L.lib().enerf_march_rays_train_count(_f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), _u8(grid, "grid"),
                                                 float(bound), float(dt_gamma), int(max_steps), int(N), int(C), int(H),
                                                 _f32(nears, "nears"), _f32(fars, "fars"), _i32(rays, "rays"),
                                                 _i32(counter, "counter"), int(perturb), int(flags),
                                                 L.stream_handle()), "march_rays_train_count")


def march_rays_train_write(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas,
                           rays, counter, perturb, zero_unwritten):
    """Write half of march_rays_train for the batch last counted (include/enerf_hip.h)."""
    L.check(L.lib().enerf_march_rays_train_write(_f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), _u8(grid, "grid"),
                                                 float(bound), float(dt_gamma), int(max_steps), int(N), int(C), int(H),
                                                 int(M), _f32(nears, "nears"), _f32(fars, "fars"), _f32(xyzs, "xyzs"),
                                                 _f32(dirs, "dirs"), _f32(deltas, "deltas"), _i32(rays, "rays"),
                                                 _i32(counter, "counter"), int(perturb), int(zero_unwritten),
                                                 L.stream_handle()), "march_rays_train_write")


def composite_rays_frame(sigmas, rgbs, deltas, rays, N, M, nears, fars, bg_color, weights_sum, depth, image,
                         used_samples=None):
    """Whole-frame inference compositing over contiguously marched samples (include/enerf_hip.h)."""
    bg, stride, scalar = _background(bg_color, N)
    L.check(L.lib().enerf_composite_rays_frame(
        _f32(sigmas, "sigmas"), _f32(rgbs, "rgbs"), _f32(deltas, "deltas"), _i32(rays, "rays"), int(N), int(M),
        _f32(nears, "nears"), _f32(fars, "fars"), bg, stride, scalar, _f32(weights_sum, "weights_sum"),
        _f32(depth, "depth"), _f32(image, "image"), None if used_samples is None else used_samples.data_ptr(),
        L.stream_handle()), "composite_rays_frame")


def _background(bg_color, N):
    """-> (pointer, stride, scalar) of the C ABI's background triple."""
    import torch
    if isinstance(bg_color, torch.Tensor):
        if bg_color.numel() == 1:
            return None, 0, float(bg_color)
        bg = _f32(bg_color, "bg_color")
        if bg_color.numel() == 3:
            return bg, 0, 0.0
        if bg_color.numel() == 3 * N:
            return bg, 3, 0.0
        raise ValueError(f"bg_color: expected 1, 3 or {3 * N} elements, got {bg_color.numel()}")
    return None, 0, float(bg_color)


def composite_rays_train_forward_blend(sigmas, rgbs, deltas, rays, M, N, weights_sum, depth, image, bg_color,
                                       out_image):
    bg, stride, scalar = _background(bg_color, N)
    L.check(L.lib().enerf_composite_rays_train_forward_blend(
        _f32(sigmas, "sigmas"), _f32(rgbs, "rgbs"), _f32(deltas, "deltas"), _i32(rays, "rays"), int(M), int(N),
        _f32(weights_sum, "weights_sum"), None if depth is None else _f32(depth, "depth"), _f32(image, "image"), bg,
        stride, scalar, _f32(out_image, "out_image"), L.stream_handle()), "composite_rays_train_forward_blend")


def composite_rays_train_backward_mse(out_image, target, grad_scale, bg_color, counter, sigmas, rgbs, deltas, rays,
                                      weights_sum, image, M, N, grad_sigmas, grad_rgbs, loss=None):
    bg, stride, scalar = _background(bg_color, N)
    L.check(L.lib().enerf_composite_rays_train_backward_mse(
        _f32(out_image, "out_image"), None if target is None else _f32(target, "target"), float(grad_scale), bg,
        stride, scalar,
        _i32(counter, "counter"), _f32(sigmas, "sigmas"), _f32(rgbs, "rgbs"), _f32(deltas, "deltas"),
        _i32(rays, "rays"), _f32(weights_sum, "weights_sum"), _f32(image, "image"), int(M), int(N),
        _f32(grad_sigmas, "grad_sigmas"), _f32(grad_rgbs, "grad_rgbs"),
        None if loss is None else _f32(loss, "loss"), L.stream_handle()), "composite_rays_train_backward_mse")


def composite_rays_train_fwd_bwd_mse(sigmas, rgbs, deltas, rays, M, N, weights_sum, image, bg_color, out_image, target,
                                     grad_scale, counter, grad_sigmas, grad_rgbs, loss=None):
    """forward_blend + backward_mse(target) in one launch (include/enerf_hip.h)."""
    bg, stride, scalar = _background(bg_color, N)
    L.check(L.lib().enerf_composite_rays_train_fwd_bwd_mse(
        _f32(sigmas, "sigmas"), _f32(rgbs, "rgbs"), _f32(deltas, "deltas"), _i32(rays, "rays"), int(M), int(N),
        _f32(weights_sum, "weights_sum"), _f32(image, "image"), bg, stride, scalar, _f32(out_image, "out_image"),
        _f32(target, "target"), float(grad_scale), _i32(counter, "counter"), _f32(grad_sigmas, "grad_sigmas"),
        _f32(grad_rgbs, "grad_rgbs"), None if loss is None else _f32(loss, "loss"), L.stream_handle()),
        "composite_rays_train_fwd_bwd_mse")


def composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, weights_sum, depth, image):
    L.check(L.lib().enerf_composite_rays_train_forward(_f32(sigmas, "sigmas"), _f32(rgbs, "rgbs"),
                                                       _f32(deltas, "deltas"), _i32(rays, "rays"), int(M), int(N),
                                                       _f32(weights_sum, "weights_sum"), _f32(depth, "depth"),
                                                       _f32(image, "image"), L.stream_handle()),
            "composite_rays_train_forward")


def composite_rays_train_backward(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N,
                                  grad_sigmas, grad_rgbs):
    L.check(L.lib().enerf_composite_rays_train_backward(
        _f32(grad_weights_sum, "grad_weights_sum"), _f32(grad_image, "grad_image"), _f32(sigmas, "sigmas"),
        _f32(rgbs, "rgbs"), _f32(deltas, "deltas"), _i32(rays, "rays"), _f32(weights_sum, "weights_sum"),
        _f32(image, "image"), int(M), int(N), _f32(grad_sigmas, "grad_sigmas"), _f32(grad_rgbs, "grad_rgbs"),
        L.stream_handle()), "composite_rays_train_backward")


def march_rays_ex(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears,
                  fars, xyzs, dirs, deltas, perturb):
    """march_rays into uninitialised buffers: unfilled slots and alignment rows are zeroed by the kernel."""
    STATS["infer_samples"] += int(n_alive) * int(n_step)
    STATS["infer_calls"] += 1
    L.check(L.lib().enerf_march_rays_ex(int(n_alive), int(n_step), _i32(rays_alive, "rays_alive"),
                                        _f32(rays_t, "rays_t"), _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"),
                                        float(bound), float(dt_gamma), int(max_steps), int(C), int(H), _u8(grid, "grid"),
                                        _f32(nears, "nears"), _f32(fars, "fars"), _f32(xyzs, "xyzs"),
                                        _f32(dirs, "dirs"), _f32(deltas, "deltas"), int(perturb), int(xyzs.shape[0]),
                                        L.stream_handle()), "march_rays_ex")


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears,
               fars, xyzs, dirs, deltas, perturb):
    STATS["infer_samples"] += int(n_alive) * int(n_step)
    STATS["infer_calls"] += 1
    L.check(L.lib().enerf_march_rays(int(n_alive), int(n_step), _i32(rays_alive, "rays_alive"),
                                     _f32(rays_t, "rays_t"), _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"),
                                     float(bound), float(dt_gamma), int(max_steps), int(C), int(H), _u8(grid, "grid"),
                                     _f32(nears, "nears"), _f32(fars, "fars"), _f32(xyzs, "xyzs"),
                                     _f32(dirs, "dirs"), _f32(deltas, "deltas"), int(perturb), L.stream_handle()),
            "march_rays")


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
    L.check(L.lib().enerf_composite_rays(int(n_alive), int(n_step), _i32(rays_alive, "rays_alive"),
                                         _f32(rays_t, "rays_t"), _f32(sigmas, "sigmas"), _f32(rgbs, "rgbs"),
                                         _f32(deltas, "deltas"), _f32(weights_sum, "weights_sum"),
                                         _f32(depth, "depth"), _f32(image, "image"), L.stream_handle()),
            "composite_rays")


def compact_rays(n_alive, rays_alive, rays_alive_old, rays_t, rays_t_old, alive_counter):
    L.check(L.lib().enerf_compact_rays(int(n_alive), _i32(rays_alive, "rays_alive"),
                                       _i32(rays_alive_old, "rays_alive_old"), _f32(rays_t, "rays_t"),
                                       _f32(rays_t_old, "rays_t_old"), _i32(alive_counter, "alive_counter"),
                                       L.stream_handle()), "compact_rays")
