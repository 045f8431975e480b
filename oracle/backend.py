"""CPU stand-ins for the four native extension modules (`_raymarching`, `_gridencoder`,
`_shencoder`, `_ffmlp`), backed by the C oracle and operating on CPU torch tensors.

TEST INFRASTRUCTURE ONLY.  Same function names, argument order and in-place output
conventions as the reference's pybind modules (raymarching/src/bindings.cpp:5-20,
gridencoder/src/bindings.cpp:5-8, shencoder/src/bindings.cpp:5-8, ffmlp/src/bindings.cpp:5-11),
so that (a) the reference's own Python wrappers can be driven on CPU when minting golden
fixtures (oracle/make_golden.py) and (b) tests can run enerf_amd's wrappers / renderer /
networks on CPU by monkeypatching their `_backend` with these objects.
"""
import types

import numpy as np
import torch

from . import oracle as O


def _np(t):
    assert t.device.type == "cpu" and t.is_contiguous(), "oracle backend needs contiguous CPU tensors"
    return t.detach().numpy()


def _store(dst, arr):
    dst.copy_(torch.from_numpy(np.ascontiguousarray(arr)).to(dst.dtype).view_as(dst))


# ------------------------------------------------------------------ _raymarching
def _rm_near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
    n, f = O.near_far_from_aabb(_np(rays_o), _np(rays_d), _np(aabb), float(min_near))
    _store(nears, n); _store(fars, f)


def _rm_polar_from_ray(rays_o, rays_d, radius, N, coords):
    _store(coords, O.polar_from_ray(_np(rays_o), _np(rays_d), float(radius)))


def _rm_morton3D(coords, N, indices):
    _store(indices, O.morton3D(_np(coords)))


def _rm_morton3D_invert(indices, N, coords):
    _store(coords, O.morton3D_invert(_np(indices)))


def _rm_packbits(grid, N, density_thresh, bitfield):
    _store(bitfield, O.packbits(_np(grid.contiguous()).reshape(-1)[: N * 8], float(density_thresh)))


def _rm_march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars,
                         xyzs, dirs, deltas, rays, counter, perturb):
    cnt = _np(counter).astype(np.int32).copy()
    x, d, dl, r, cnt = O.march_rays_train(_np(rays_o), _np(rays_d), _np(grid), float(bound), float(dt_gamma),
                                           int(max_steps), int(C), int(H), int(M), _np(nears), _np(fars),
                                           int(perturb), counter=cnt)
    _store(xyzs, x); _store(dirs, d); _store(deltas, dl); _store(rays, r); _store(counter, cnt)


def _rm_composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, weights_sum, depth, image):
    ws, dp, im = O.composite_rays_train_forward(_np(sigmas), _np(rgbs), _np(deltas), _np(rays))
    _store(weights_sum, ws); _store(depth, dp); _store(image, im)


def _rm_composite_rays_train_backward(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum,
                                      image, M, N, grad_sigmas, grad_rgbs):
    gs, gc = O.composite_rays_train_backward(_np(grad_weights_sum), _np(grad_image), _np(sigmas), _np(rgbs),
                                             _np(deltas), _np(rays), _np(weights_sum), _np(image))
    _store(grad_sigmas, gs); _store(grad_rgbs, gc)


def _rm_march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid,
                   nears, fars, xyzs, dirs, deltas, perturb):
    x, d, dl = O.march_rays(int(n_alive), int(n_step), _np(rays_alive), _np(rays_t), _np(rays_o), _np(rays_d),
                            float(bound), float(dt_gamma), int(max_steps), int(C), int(H), _np(grid),
                            _np(nears), _np(fars), xyzs.shape[0], int(perturb))
    _store(xyzs, x); _store(dirs, d); _store(deltas, dl)


def _rm_composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
    # rays_alive / rays_t may be row views of a [2,N] tensor: contiguous rows, operate in place through numpy views
    O.composite_rays(int(n_alive), int(n_step), _np(rays_alive), _np(rays_t), _np(sigmas.float().contiguous()),
                     _np(rgbs.float().contiguous()), _np(deltas), _np(weights_sum), _np(depth), _np(image))


def _rm_compact_rays(n_alive, rays_alive, rays_alive_old, rays_t, rays_t_old, alive_counter):
    ra, rt, cnt = O.compact_rays(int(n_alive), _np(rays_alive_old), _np(rays_t_old))
    rays_alive[: ra.shape[0]].copy_(torch.from_numpy(ra))
    rays_t[: rt.shape[0]].copy_(torch.from_numpy(rt))
    alive_counter[0] = int(alive_counter[0]) + cnt


raymarching_backend = types.SimpleNamespace(
    near_far_from_aabb=_rm_near_far_from_aabb, polar_from_ray=_rm_polar_from_ray, morton3D=_rm_morton3D,
    morton3D_invert=_rm_morton3D_invert, packbits=_rm_packbits, march_rays_train=_rm_march_rays_train,
    composite_rays_train_forward=_rm_composite_rays_train_forward,
    composite_rays_train_backward=_rm_composite_rays_train_backward, march_rays=_rm_march_rays,
    composite_rays=_rm_composite_rays, compact_rays=_rm_compact_rays)


# ------------------------------------------------------------------ _gridencoder
def _ge_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, calc_grad_inputs, dy_dx, gridtype):
    out, jac = O.grid_encode_forward(_np(inputs), _np(embeddings.float()), _np(offsets), float(S), int(H),
                                     bool(calc_grad_inputs), int(gridtype))
    _store(outputs, out)
    if calc_grad_inputs:
        _store(dy_dx, jac)


def _ge_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, calc_grad_inputs, dy_dx,
                 grad_inputs, gridtype):
    ge, gi = O.grid_encode_backward(_np(grad.float()), _np(inputs), _np(embeddings.float()), _np(offsets), float(S),
                                    int(H), _np(dy_dx.float()) if calc_grad_inputs else None, int(gridtype))
    grad_embeddings.add_(torch.from_numpy(ge).to(grad_embeddings.dtype))   # kernel atomically adds into zeros
    if calc_grad_inputs:
        _store(grad_inputs, gi)


gridencoder_backend = types.SimpleNamespace(grid_encode_forward=_ge_forward, grid_encode_backward=_ge_backward)


# ------------------------------------------------------------------ _shencoder
def _sh_forward(inputs, outputs, B, D, C, calc_grad_inputs, dy_dx):
    out, jac = O.sh_encode_forward(_np(inputs.float()), int(C), bool(calc_grad_inputs))
    _store(outputs, out)
    if calc_grad_inputs:
        _store(dy_dx, jac)


def _sh_backward(grad, inputs, B, D, C, dy_dx, grad_inputs):
    gi = O.sh_encode_backward(_np(grad.float()), _np(inputs.float()), int(C), _np(dy_dx.float()))
    grad_inputs.add_(torch.from_numpy(gi).to(grad_inputs.dtype))


shencoder_backend = types.SimpleNamespace(sh_encode_forward=_sh_forward, sh_encode_backward=_sh_backward)


# ------------------------------------------------------------------ _ffmlp
def _rnd_of(t):
    return {torch.bfloat16: 1, torch.float16: 2}.get(t.dtype, 0)


def _ff_forward(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                forward_buffer, outputs):
    out, fb = O.ffmlp_forward(_np(inputs.float()), _np(weights.float()), input_dim, output_dim, hidden_dim,
                              num_layers, activation, output_activation, rnd=_rnd_of(inputs))
    _store(outputs, out); _store(forward_buffer, fb)


def _ff_inference(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                  inference_buffer, outputs):
    out, _ = O.ffmlp_forward(_np(inputs.float()), _np(weights.float()), input_dim, output_dim, hidden_dim,
                             num_layers, activation, output_activation, rnd=_rnd_of(inputs), want_buffer=False)
    _store(outputs, out)


def _ff_backward(grad, inputs, weights, forward_buffer, B, input_dim, output_dim, hidden_dim, num_layers, activation,
                 output_activation, calc_grad_inputs, backward_buffer, grad_inputs, grad_weights):
    gi, gw, bb = O.ffmlp_backward(_np(grad.float()), _np(inputs.float()), _np(weights.float()),
                                  _np(forward_buffer.float()), input_dim, output_dim, hidden_dim, num_layers,
                                  activation, bool(calc_grad_inputs), rnd=_rnd_of(inputs))
    _store(backward_buffer, bb)
    grad_weights.add_(torch.from_numpy(gw).to(grad_weights.dtype))
    if calc_grad_inputs:
        _store(grad_inputs, gi)


ffmlp_backend = types.SimpleNamespace(ffmlp_forward=_ff_forward, ffmlp_inference=_ff_inference,
                                      ffmlp_backward=_ff_backward, allocate_splitk=lambda n: None,
                                      free_splitk=lambda: None)


def as_module(name, ns):
    """Wrap a backend namespace as an importable module object (for sys.modules injection)."""
    m = types.ModuleType(name)
    m.__dict__.update(vars(ns))
    return m
