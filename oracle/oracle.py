"""ctypes/numpy driver for oracle/liboracle.so (the CPU restatement in enerf_oracle.c).

TEST INFRASTRUCTURE ONLY -- nothing in enerf_amd/ may import this module.
Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
"""
import contextlib
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

_c = ctypes
_f32p = _c.POINTER(_c.c_float)
_i32p = _c.POINTER(_c.c_int32)
_u8p = _c.POINTER(_c.c_uint8)
_u32p = _c.POINTER(_c.c_uint32)
_u32, _f32, _int, _u64 = _c.c_uint32, _c.c_float, _c.c_int, _c.c_uint64

_SIGS = {
    "orc_set_threads": [_int],
    "orc_pcg32_stream": [_u64, _u64, _u32, _u32p, _f32p],
    "orc_near_far_from_aabb": [_f32p, _f32p, _f32p, _u32, _f32, _f32p, _f32p],
    "orc_polar_from_ray": [_f32p, _f32p, _f32, _u32, _f32p],
    "orc_morton3D": [_i32p, _u32, _i32p],
    "orc_morton3D_invert": [_i32p, _u32, _i32p],
    "orc_packbits": [_f32p, _u32, _f32, _u8p],
    "orc_march_rays_train": [_f32p, _f32p, _u8p, _f32, _f32, _u32, _u32, _u32, _u32, _u32,
                             _f32p, _f32p, _f32p, _f32p, _f32p, _i32p, _i32p, _u32],
    "orc_composite_rays_train_forward": [_f32p, _f32p, _f32p, _i32p, _u32, _u32, _f32p, _f32p, _f32p],
    "orc_composite_rays_train_backward": [_f32p, _f32p, _f32p, _f32p, _f32p, _i32p, _f32p, _f32p,
                                          _u32, _u32, _f32p, _f32p],
    "orc_march_rays": [_u32, _u32, _i32p, _f32p, _f32p, _f32p, _f32, _f32, _u32, _u32, _u32, _u8p,
                       _f32p, _f32p, _f32p, _f32p, _f32p, _u32],
    "orc_composite_rays": [_u32, _u32, _i32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p],
    "orc_compact_rays": [_u32, _i32p, _i32p, _f32p, _f32p, _i32p],
    "orc_grid_level_params": [_u32, _f32, _u32, _f32p, _u32p],
    "orc_grid_scale_nudge": [_u32, _int],
    "orc_grid_encode_forward": [_f32p, _f32p, _i32p, _f32p, _u32, _u32, _u32, _u32, _f32, _u32, _int, _f32p, _u32],
    "orc_grid_encode_backward": [_f32p, _f32p, _f32p, _i32p, _f32p, _u32, _u32, _u32, _u32, _f32, _u32,
                                 _int, _f32p, _f32p, _u32],
    "orc_sh_encode_forward": [_f32p, _f32p, _u32, _u32, _u32, _int, _f32p],
    "orc_sh_encode_backward": [_f32p, _f32p, _u32, _u32, _u32, _f32p, _f32p],
    "orc_ffmlp_forward": [_f32p, _f32p, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _f32p, _f32p, _int],
    "orc_ffmlp_backward": [_f32p, _f32p, _f32p, _f32p, _u32, _u32, _u32, _u32, _u32, _u32, _int,
                           _f32p, _f32p, _f32p, _int],
}

_lib = None


def build(force=False):
    """gcc-build liboracle.so next to this file (idempotent)."""
    src = os.path.join(_HERE, "enerf_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        for name, sig in _SIGS.items():
            fn = getattr(_lib, name)
            fn.argtypes = sig
            fn.restype = None
    return _lib


def set_threads(n):
    lib().orc_set_threads(int(n))


def _p(a, ct):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be C-contiguous"
    return a.ctypes.data_as(ct)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# ---------------------------------------------------------------- raymarching
def pcg32_stream(seed, seq, n):
    u = np.zeros(n, np.uint32)
    f = np.zeros(n, np.float32)
    lib().orc_pcg32_stream(seed, seq, n, _p(u, _u32p), _p(f, _f32p))
    return u, f


def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    rays_o, rays_d, aabb = f32(rays_o).reshape(-1, 3), f32(rays_d).reshape(-1, 3), f32(aabb)
    N = rays_o.shape[0]
    nears, fars = np.empty(N, np.float32), np.empty(N, np.float32)
    lib().orc_near_far_from_aabb(_p(rays_o, _f32p), _p(rays_d, _f32p), _p(aabb, _f32p), N, min_near,
                                 _p(nears, _f32p), _p(fars, _f32p))
    return nears, fars


def polar_from_ray(rays_o, rays_d, radius):
    rays_o, rays_d = f32(rays_o).reshape(-1, 3), f32(rays_d).reshape(-1, 3)
    N = rays_o.shape[0]
    coords = np.empty((N, 2), np.float32)
    lib().orc_polar_from_ray(_p(rays_o, _f32p), _p(rays_d, _f32p), radius, N, _p(coords, _f32p))
    return coords


def morton3D(coords):
    coords = i32(coords).reshape(-1, 3)
    out = np.empty(coords.shape[0], np.int32)
    lib().orc_morton3D(_p(coords, _i32p), coords.shape[0], _p(out, _i32p))
    return out


def morton3D_invert(indices):
    indices = i32(indices).reshape(-1)
    out = np.empty((indices.shape[0], 3), np.int32)
    lib().orc_morton3D_invert(_p(indices, _i32p), indices.shape[0], _p(out, _i32p))
    return out


def packbits(grid, thresh):
    grid = f32(grid)
    N = grid.size // 8
    out = np.empty(N, np.uint8)
    lib().orc_packbits(_p(grid, _f32p), N, thresh, _p(out, _u8p))
    return out


def march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, C, H, M, nears, fars, perturb,
                     counter=None):
    rays_o, rays_d = f32(rays_o).reshape(-1, 3), f32(rays_d).reshape(-1, 3)
    N = rays_o.shape[0]
    grid = np.ascontiguousarray(grid, np.uint8)
    nears, fars = f32(nears), f32(fars)
    xyzs = np.zeros((M, 3), np.float32)
    dirs = np.zeros((M, 3), np.float32)
    deltas = np.zeros((M, 2), np.float32)
    rays = np.empty((N, 3), np.int32)
    if counter is None:
        counter = np.zeros(2, np.int32)
    lib().orc_march_rays_train(_p(rays_o, _f32p), _p(rays_d, _f32p), _p(grid, _u8p), bound, dt_gamma, max_steps,
                               N, C, H, M, _p(nears, _f32p), _p(fars, _f32p), _p(xyzs, _f32p), _p(dirs, _f32p),
                               _p(deltas, _f32p), _p(rays, _i32p), _p(counter, _i32p), int(perturb))
    return xyzs, dirs, deltas, rays, counter


def composite_rays_train_forward(sigmas, rgbs, deltas, rays):
    sigmas, rgbs, deltas, rays = f32(sigmas), f32(rgbs), f32(deltas), i32(rays)
    M, N = sigmas.shape[0], rays.shape[0]
    ws, depth, image = np.empty(N, np.float32), np.empty(N, np.float32), np.empty((N, 3), np.float32)
    lib().orc_composite_rays_train_forward(_p(sigmas, _f32p), _p(rgbs, _f32p), _p(deltas, _f32p), _p(rays, _i32p),
                                           M, N, _p(ws, _f32p), _p(depth, _f32p), _p(image, _f32p))
    return ws, depth, image


def composite_rays_train_backward(grad_ws, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image):
    sigmas, rgbs, deltas, rays = f32(sigmas), f32(rgbs), f32(deltas), i32(rays)
    grad_ws, grad_image, weights_sum, image = f32(grad_ws), f32(grad_image), f32(weights_sum), f32(image)
    M, N = sigmas.shape[0], rays.shape[0]
    gs, gc = np.zeros(M, np.float32), np.zeros((M, 3), np.float32)
    lib().orc_composite_rays_train_backward(_p(grad_ws, _f32p), _p(grad_image, _f32p), _p(sigmas, _f32p),
                                            _p(rgbs, _f32p), _p(deltas, _f32p), _p(rays, _i32p),
                                            _p(weights_sum, _f32p), _p(image, _f32p), M, N, _p(gs, _f32p), _p(gc, _f32p))
    return gs, gc


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid,
               nears, fars, M, perturb):
    rays_o, rays_d = f32(rays_o).reshape(-1, 3), f32(rays_d).reshape(-1, 3)
    rays_alive, rays_t = i32(rays_alive), f32(rays_t)
    grid = np.ascontiguousarray(grid, np.uint8)
    nears, fars = f32(nears), f32(fars)
    xyzs = np.zeros((M, 3), np.float32)
    dirs = np.zeros((M, 3), np.float32)
    deltas = np.zeros((M, 2), np.float32)
    lib().orc_march_rays(n_alive, n_step, _p(rays_alive, _i32p), _p(rays_t, _f32p), _p(rays_o, _f32p),
                         _p(rays_d, _f32p), bound, dt_gamma, max_steps, C, H, _p(grid, _u8p), _p(nears, _f32p),
                         _p(fars, _f32p), _p(xyzs, _f32p), _p(dirs, _f32p), _p(deltas, _f32p), int(perturb))
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
    """In place on rays_t, weights_sum, depth, image (must be float32 C-contiguous numpy arrays)."""
    for a in (rays_t, weights_sum, depth, image):
        assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    rays_alive, sigmas, rgbs, deltas = i32(rays_alive), f32(sigmas), f32(rgbs), f32(deltas)
    lib().orc_composite_rays(n_alive, n_step, _p(rays_alive, _i32p), _p(rays_t, _f32p), _p(sigmas, _f32p),
                             _p(rgbs, _f32p), _p(deltas, _f32p), _p(weights_sum, _f32p), _p(depth, _f32p),
                             _p(image, _f32p))


def compact_rays(n_alive, rays_alive_old, rays_t_old):
    rays_alive_old, rays_t_old = i32(rays_alive_old), f32(rays_t_old)
    rays_alive = np.zeros_like(rays_alive_old)
    rays_t = np.zeros_like(rays_t_old)
    counter = np.zeros(1, np.int32)
    lib().orc_compact_rays(n_alive, _p(rays_alive, _i32p), _p(rays_alive_old, _i32p), _p(rays_t, _f32p),
                           _p(rays_t_old, _f32p), _p(counter, _i32p))
    return rays_alive, rays_t, int(counter[0])


# ---------------------------------------------------------------- gridencoder
def grid_level_params(level, S, H):
    sc = np.zeros(1, np.float32)
    res = np.zeros(1, np.uint32)
    lib().orc_grid_level_params(level, S, H, _p(sc, _f32p), _p(res, _u32p))
    return float(sc[0]), int(res[0])


@contextlib.contextmanager
def grid_exp2f_nudged(level, ulps):
    """Inside: level `level` uses exp2f(level * S) moved by `ulps` (+-1) ulp -- a platform libm whose exp2f is not glibc's
    (see orc_grid_level_params).  For tests against the reference's kernels only."""
    lib().orc_grid_scale_nudge(level, ulps)
    try:
        yield
    finally:
        lib().orc_grid_scale_nudge(level, 0)


def grid_offsets(input_dim=3, num_levels=16, level_dim=2, per_level_scale=2.0, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None):
    """Offset table as gridencoder/grid.py:96-123 computes it (float64 numpy)."""
    if desired_resolution is not None:
        per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
    offsets, offset = [], 0
    max_params = 2 ** log2_hashmap_size
    for i in range(num_levels):
        resolution = int(np.ceil(base_resolution * per_level_scale ** i))
        params_in_level = min(max_params, (resolution + 1) ** input_dim)
        params_in_level = int(np.ceil(params_in_level / 8) * 8)
        offsets.append(offset)
        offset += params_in_level
    offsets.append(offset)
    return np.array(offsets, dtype=np.int32), float(per_level_scale)


def grid_encode_forward(inputs, embeddings, offsets, S, H, calc_grad_inputs=False, gridtype=0):
    """Returns outputs [L,B,C] (the backend layout) and dy_dx [B, L*D*C] or None."""
    inputs, embeddings, offsets = f32(inputs), f32(embeddings), i32(offsets)
    B, D = inputs.shape
    C = embeddings.shape[1]
    L = offsets.shape[0] - 1
    outputs = np.empty((L, B, C), np.float32)
    dy_dx = np.empty((B, L * D * C), np.float32) if calc_grad_inputs else None
    lib().orc_grid_encode_forward(_p(inputs, _f32p), _p(embeddings, _f32p), _p(offsets, _i32p), _p(outputs, _f32p),
                                  B, D, C, L, S, H, int(calc_grad_inputs), _p(dy_dx, _f32p), gridtype)
    return outputs, dy_dx


def grid_encode_backward(grad, inputs, embeddings, offsets, S, H, dy_dx=None, gridtype=0):
    """grad [L,B,C] -> grad_embeddings [rows,C], grad_inputs [B,D] or None."""
    grad, inputs, embeddings, offsets = f32(grad), f32(inputs), f32(embeddings), i32(offsets)
    B, D = inputs.shape
    C = embeddings.shape[1]
    L = offsets.shape[0] - 1
    ge = np.zeros_like(embeddings)
    gi = np.zeros((B, D), np.float32) if dy_dx is not None else None
    lib().orc_grid_encode_backward(_p(grad, _f32p), _p(inputs, _f32p), _p(embeddings, _f32p), _p(offsets, _i32p),
                                   _p(ge, _f32p), B, D, C, L, S, H, int(dy_dx is not None),
                                   _p(f32(dy_dx) if dy_dx is not None else None, _f32p), _p(gi, _f32p), gridtype)
    return ge, gi


# ---------------------------------------------------------------- shencoder
def sh_encode_forward(inputs, degree, calc_grad_inputs=False):
    inputs = f32(inputs)
    B, D = inputs.shape
    out = np.empty((B, degree * degree), np.float32)
    dy_dx = np.empty((B, D * degree * degree), np.float32) if calc_grad_inputs else None
    lib().orc_sh_encode_forward(_p(inputs, _f32p), _p(out, _f32p), B, D, degree, int(calc_grad_inputs), _p(dy_dx, _f32p))
    return out, dy_dx


def sh_encode_backward(grad, inputs, degree, dy_dx):
    grad, inputs, dy_dx = f32(grad), f32(inputs), f32(dy_dx)
    B, D = inputs.shape
    gi = np.zeros((B, D), np.float32)
    lib().orc_sh_encode_backward(_p(grad, _f32p), _p(inputs, _f32p), B, D, degree, _p(dy_dx, _f32p), _p(gi, _f32p))
    return gi


# ---------------------------------------------------------------- ffmlp
def ffmlp_forward(inputs, weights, input_dim, output_dim, hidden_dim, num_layers, activation=0,
                  output_activation=6, rnd=0, want_buffer=True):
    inputs, weights = f32(inputs), f32(weights)
    B = inputs.shape[0]
    fb = np.empty((num_layers, B, hidden_dim), np.float32) if want_buffer else None
    out = np.empty((B, output_dim), np.float32)
    lib().orc_ffmlp_forward(_p(inputs, _f32p), _p(weights, _f32p), B, input_dim, output_dim, hidden_dim, num_layers,
                            activation, output_activation, _p(fb, _f32p), _p(out, _f32p), rnd)
    return out, fb


def ffmlp_backward(grad, inputs, weights, forward_buffer, input_dim, output_dim, hidden_dim, num_layers,
                   activation=0, calc_grad_inputs=True, rnd=0):
    grad, inputs, weights, forward_buffer = f32(grad), f32(inputs), f32(weights), f32(forward_buffer)
    B = inputs.shape[0]
    bb = np.zeros((num_layers, B, hidden_dim), np.float32)
    gi = np.zeros((B, input_dim), np.float32)
    gw = np.zeros_like(weights)
    lib().orc_ffmlp_backward(_p(grad, _f32p), _p(inputs, _f32p), _p(weights, _f32p), _p(forward_buffer, _f32p),
                             B, input_dim, output_dim, hidden_dim, num_layers, activation, int(calc_grad_inputs),
                             _p(bb, _f32p), _p(gi, _f32p), _p(gw, _f32p), rnd)
    return (gi if calc_grad_inputs else None), gw, bb
