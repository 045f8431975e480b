/*
 * enerf_hip.h -- C ABI of libenerf_hip.so, the MI355X (gfx950) implementation of the
 * instant-ngp hot path that knelk/enerf sits on.
 *
 * This is the drop-in boundary.  The reference binds its native code through four
 * pybind11 modules (`_raymarching`, `_gridencoder`, `_shencoder`, `_ffmlp`) whose
 * functions take at::Tensor arguments that the *caller* has already allocated
 * (and zero-filled where the kernels only partially write).  Each entry point below
 * replaces exactly one of those bound functions; the argument order is the
 * reference's, with every tensor flattened to a raw device pointer, followed by
 *   - explicit dtype selectors where the reference dispatches on tensor dtype,
 *   - `stream`: the hipStream_t to launch on (the reference uses the legacy default
 *     stream; pass torch's current stream).
 * All pointers are DEVICE pointers unless marked [host].  All functions are
 * asynchronous, allocate nothing the caller can see, and return 0 on success or a
 * negative ENERF_E_* / positive hipError_t code; enerf_last_error() describes the
 * failure (the reference throws c10::Error / std::runtime_error at the same places).
 * enerf_amd/backends/_*.py are the ctypes shims that re-create the reference's
 * pybind signatures on top of this ABI.
 *
 * Reference interfaces replaced (paths relative to the knelk/enerf tree):
 *   raymarching/src/raymarching.h:7-19, raymarching/src/bindings.cpp:5-20
 *   gridencoder/src/gridencoder.h:12-13, gridencoder/src/bindings.cpp:5-8
 *   shencoder/src/shencoder.h:9,12,     shencoder/src/bindings.cpp:5-8
 *   ffmlp/src/ffmlp.h:8-14,             ffmlp/src/bindings.cpp:5-11
 */
#ifndef ENERF_HIP_H
#define ENERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* enerf_stream_t; /* hipStream_t */

/* error codes (negative; positive values are hipError_t) */
#define ENERF_OK 0
#define ENERF_E_BADARG (-1)      /* unsupported D / C / degree / width / dtype ...           */
#define ENERF_E_NOMEM (-2)       /* internal workspace allocation failed                      */
#define ENERF_E_UNSUPPORTED (-3) /* feature the reference asserts off as well                 */

/* element types for entry points that the reference dispatches on tensor dtype */
#define ENERF_F32 0
#define ENERF_F16 1
#define ENERF_BF16 2

const char* enerf_last_error(void);
/* ABI version of this header; bumped on any signature change.  This #define is the ONE place the number lives:
 * the library returns it (csrc/runtime.hip), enerf_amd/_lib.py parses it and refuses a library that answers differently,
 * __graft_entry__.build() and tests/test_abi.py compare against the parsed value.
 * 2: enerf_train_step_args lost its RCCL-tail fields, enerf_dp_* retired, enerf_nerf_mlp_* added. */
#define ENERF_ABI_VERSION 2
int enerf_abi_version(void);
/* The library keeps grow-only scratch buffers per device (march chunk log, grid-backward record lists, ...).  Growing one
 * frees the old allocation; the counter returned here moves every time that happens.  A caller that captured library
 * launches into a HIP graph must re-capture when the counter has moved since the capture (the graph holds the old
 * pointers). */
uint64_t enerf_workspace_generation(void);

/* ------------------------------------------------------------------ raymarching
 * raymarching/src/raymarching.cu; all float tensors are fp32 (the wrappers
 * custom_fwd(cast_inputs=torch.float32), raymarching/raymarching.py:21,54,131,163). */

/* raymarching.cu:150-158  near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars) */
int enerf_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N,
                             float min_near, float* nears, float* fars, enerf_stream_t stream);

/* raymarching.cu:203-211  polar_from_ray(rays_o, rays_d, radius, N, coords) */
int enerf_polar_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords,
                         enerf_stream_t stream);

/* raymarching.cu:231-234  morton3D(coords, N, indices) */
int enerf_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, enerf_stream_t stream);

/* raymarching.cu:259-262  morton3D_invert(indices, N, coords) */
int enerf_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, enerf_stream_t stream);

/* raymarching.cu:294-302  packbits(grid, N, density_thresh, bitfield); N = number of output bytes */
int enerf_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield, enerf_stream_t stream);

/* ---- density-grid maintenance: the device side of NeRFRenderer.update_extra_state (nerf/renderer.py:472-560) --------
 * The reference builds the candidate cells and applies the results with ~40 torch ops per cascade and three host
 * synchronisations; these two entry points are that logic around the density network evaluation, which stays with the
 * caller (hash grid + sigma MLP on `xyzs`, giving `sigmas`).  H must be a power of two (the Morton layout), 1 <= C <= 8.
 *
 * enerf_density_grid_cells: Morton indices (int32, per cascade) and jittered query positions [P,3] of the cells to evaluate,
 * cascades concatenated.
 *   density_grid == NULL : full sweep (renderer.py:484-512), P = C * H^3, cells in x-fastest order.
 *   density_grid != NULL : partial update (renderer.py:514-538), P = C * 2 * n_uniform: per cascade n_uniform uniformly
 *                          drawn cells and n_uniform draws with replacement from the cells with density > 0 (all uniform
 *                          when none is), emitted sorted by Morton index.
 * Positions: (2 * coord / (H - 1) - 1) * (bound_c - half) + U(-half, half), bound_c = min(2^cas, bound), half =
 * bound_c / H.  Random numbers come from a counter-based generator keyed by `seed` (same distribution as the reference's
 * torch.randint / rand_like calls, not the same stream). */
int enerf_density_grid_cells(const float* density_grid, uint32_t C, uint32_t H, float bound, uint32_t n_uniform,
                             uint64_t seed, int32_t* indices, float* xyzs, enerf_stream_t stream);

/* enerf_mark_untrained_grid (renderer.py:408-469): density_grid[cas, cell] = -1 for every cell whose centre no camera
 * sees: cam = R^T (centre - t) for each c2w pose ([B,4,4] row-major fp32, pose_stride 16, or [B,3,4], 12); seen when
 * cam.z > 0 and |cam.x| < cx/fx * cam.z + 2*half_cell and |cam.y| < cy/fy * cam.z + 2*half_cell. */
int enerf_mark_untrained_grid(const float* poses, uint32_t n_poses, uint32_t pose_stride, float fx, float fy, float cx,
                              float cy, uint32_t C, uint32_t H, float bound, float* density_grid,
                              enerf_stream_t stream);

/* enerf_density_grid_update (renderer.py:541-558): tmp_grid = -1; tmp_grid[cas, indices] = sigmas * sigma_scale (n_per_cascade
 * entries per cascade); where density_grid >= 0 and tmp_grid >= 0: density_grid = max(density_grid * decay, tmp_grid);
 * mean = mean(clamp(density_grid, 0)); bitfield = packbits(density_grid, min(mean, density_thresh)).
 * stats (device, 2 doubles) receives {mean, sum of step_counter[0:total_step][0]} -- the one read-back of the update
 * (mean_density, mean_count).  step_counter: int32 [16,2] (may be NULL when total_step == 0). */
int enerf_density_grid_update(const int32_t* indices /* NULL: the full sweep's own order */, const float* sigmas, uint32_t n_per_cascade, uint32_t C, uint32_t H,
                              float sigma_scale, float decay, float density_thresh, float* density_grid,
                              uint8_t* bitfield, const int32_t* step_counter, uint32_t total_step, double* stats,
                              enerf_stream_t stream);

/* raymarching.cu:482-490  march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M,
 *                                          nears, fars, xyzs, dirs, deltas, rays, counter, perturb)
 * xyzs/dirs/deltas must be zero-filled by the caller (raymarching.py:205-207).
 * Slot allocation is deterministic here: rays[n] = (n, counter[0]@entry + exclusive_scan(num_steps)[n],
 * num_steps[n]); counter[0] += sum(num_steps); counter[1] += N.  (The reference's atomics give the same
 * multiset in hardware order.) */
int enerf_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound,
                           float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                           const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                           int32_t* rays, int32_t* counter, uint32_t perturb, enerf_stream_t stream);

/* Extension of march_rays_train for callers that hand over UNINITIALISED sample buffers (zero_unwritten != 0): every
 * row no ray writes -- past the last reservation, and a dropped ray's reservation clipped to M -- is zero-filled by the
 * write pass itself, so the three torch.zeros of raymarching.py:205-207 become torch.empty.  zero_unwritten == 0 is
 * exactly enerf_march_rays_train.  `zero_unwritten` is a flag word: bit 0 as above; bit 1 says the batch is being
 * prepared ahead of its step on a side stream: the count pass then runs with one wavefront per SIMD (rays in turn) so
 * that it leaves the registers of the chip to the step it runs beside; bit 2: test rays against the occupied cells'
 * bounding box first (enerf_occupied_box_update); bit 3 (value 8): `counter` is taken as (0, 0) whatever it holds --
 * the wrapper's counter.zero_() (raymarching.py:198) without its launch.  Same rows, same counts, bit for bit. */
int enerf_march_rays_train_ex(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound,
                              float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                              const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                              int32_t* rays, int32_t* counter, uint32_t perturb, uint32_t zero_unwritten,
                              enerf_stream_t stream);

/* Arms the NEXT enerf_march_rays_train(_ex / _count) call: the two words of `counter` it leaves on the device -- samples
 * reserved, rays marched -- are also written to host_counter[0], [1] (pinned, device-visible host memory, e.g. a torch
 * tensor with pin_memory=True), word 1 last: a host that pre-set both to -1 and watches them has the count without
 * queueing a copy (and an event) behind the march.  Batches above 16 384 rays get the copy. */
int enerf_march_mirror_count(int32_t* host_counter);

/* Arms the NEXT enerf_march_rays_train(_ex / _count) call: its count pass computes near / far of every ray against `aabb`
 * itself (enerf_near_far_from_aabb's arithmetic, bit for bit) and WRITES them to the nears / fars arrays the call is
 * given (they need not be initialised) -- the near_far launch at the head of a march that is prepared ahead of its step
 * disappears.  Marchers that do not take the wave-per-ray route run the near_far kernel first, on the call's stream.
 * `aabb` is read by that march's kernels, not here: it must stay valid until they have run. */
int enerf_march_fuse_near_far(const float* aabb, float min_near);

/* Bounding box of the occupied cells of a density bitfield, kept by the library for the fixed-step training marcher.
 * A sample can only be emitted inside an occupied cell's box, so a ray that misses the union's bounding box emits
 * nothing and a ray emits nothing after leaving it: with flag bit 2 (value 4) of enerf_march_rays_train_ex /
 * enerf_march_rays_train_count the count pass tests every ray against the box first and marches only to the box's far
 * side -- identical rays / counter / samples (tests), half of the synthetic scene's rays settled without marching.
 * The caller sets the flag only while the box computed here is current for `grid`: call this again (same stream
 * order as the marches that follow) whenever the bitfield's contents change.  The flag is ignored when the box was
 * computed for another pointer / C / H / bound. */
int enerf_occupied_box_update(const uint8_t* grid, uint32_t C, uint32_t H, float bound, enerf_stream_t stream);

/* The two halves of enerf_march_rays_train_ex, for callers that size the sample buffers from the count instead of the
 * reference's worst case (raymarching.py:195-228: while no sample budget exists the wrapper allocates and zero-fills
 * M = N * max_steps rows -- 134 MB at 4096 rays -- marches, reads counter[0] back and crops to that count rounded up to
 * `align`).  The count pass already knows the total on the device:
 *   _count: near/far -> per-ray sample counts -> deterministic scan: rays[n] = (n, offset, count), counter[0] += sum,
 *           counter[1] += N.  Nothing is written to sample buffers.  flags bit 1 = background launch (see _ex),
 *           bit 2 = use the occupied box (enerf_occupied_box_update), bit 3 = counter starts from (0, 0).
 *   _write: the write pass of the SAME batch (same rays / nears / fars / grid / perturb; the fixed-step marcher keeps a
 *           per-process chunk log between the two calls, so no other training march may run in between) into buffers of
 *           M rows; `M` is also the M of the reference's drop rule (`offset + count >= M`: the ray writes nothing).
 *           zero_unwritten bit 0 as in _ex.
 * _count followed by _write with the same M is exactly enerf_march_rays_train_ex. */
int enerf_march_rays_train_count(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound,
                                 float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                 const float* nears, const float* fars, int32_t* rays, int32_t* counter,
                                 uint32_t perturb, uint32_t flags, enerf_stream_t stream);
int enerf_march_rays_train_write(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound,
                                 float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                 const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                                 const int32_t* rays, const int32_t* counter, uint32_t perturb,
                                 uint32_t zero_unwritten, enerf_stream_t stream);

/* raymarching.cu:581-589  composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, weights_sum, depth, image) */
int enerf_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas,
                                       const int32_t* rays, uint32_t M, uint32_t N, float* weights_sum,
                                       float* depth, float* image, enerf_stream_t stream);

/* raymarching.cu:685-693  composite_rays_train_backward(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays,
 *                                                       weights_sum, image, M, N, grad_sigmas, grad_rgbs)
 * grad_sigmas / grad_rgbs must be zero-filled by the caller (raymarching.py:277-278). */
int enerf_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image, const float* sigmas,
                                        const float* rgbs, const float* deltas, const int32_t* rays,
                                        const float* weights_sum, const float* image, uint32_t M, uint32_t N,
                                        float* grad_sigmas, float* grad_rgbs, enerf_stream_t stream);

/* composite_rays_train_forward + the background blend that follows it in the renderer (nerf/renderer.py:352
 * `image = image + (1 - weights_sum).unsqueeze(-1) * bg_color`), one launch: out_image[N,3] receives the blend, `image`
 * still receives the unblended colour (the backward needs it).  bg_color: NULL -> the scalar bg_scalar; else [3]
 * (bg_stride 0) or [N,3] (bg_stride 3).  depth may be NULL (a training step that does not use it). */
int enerf_composite_rays_train_forward_blend(const float* sigmas, const float* rgbs, const float* deltas,
                                             const int32_t* rays, uint32_t M, uint32_t N, float* weights_sum,
                                             float* depth, float* image, const float* bg_color, uint32_t bg_stride,
                                             float bg_scalar, float* out_image, enerf_stream_t stream);

/* composite_rays_train_backward for loss = mean((out_image - target)^2) * upstream (nerf/utils.py:628 with the default
 * MSE criterion): grad_image = (out_image - target) * grad_scale with grad_scale = 2 / (3 N) * upstream, and
 * grad_weights_sum = -(grad_image . bg), both formed in the kernel.  grad_sigmas / grad_rgbs may be UNINITIALISED: rows no
 * ray covers are zero-filled here (counter = the march's counter, counter[0] = samples reserved).
 * target == NULL: `out_image` is taken to hold d loss / d out_image of some other loss (times grad_scale), everything
 * else as above -- the general backward of "composite + background blend" into uninitialised buffers.
 * loss (optional, device scalar): mean((out_image - target)^2) is ADDED to it (one float atomic per workgroup, so the
 * value is reproducible to rounding only) -- the caller zeroes it; the gradients do not depend on it. */
int enerf_composite_rays_train_backward_mse(const float* out_image, const float* target, float grad_scale,
                                            const float* bg_color, uint32_t bg_stride, float bg_scalar,
                                            const int32_t* counter, const float* sigmas, const float* rgbs,
                                            const float* deltas, const int32_t* rays, const float* weights_sum,
                                            const float* image, uint32_t M, uint32_t N, float* grad_sigmas,
                                            float* grad_rgbs, float* loss, enerf_stream_t stream);

/* composite_rays_train forward (+ blend) and its MSE backward in ONE launch, for steps whose loss is
 * mean((out_image - target)^2) * upstream (the reference's default criterion, nerf/utils.py:628): equals
 * enerf_composite_rays_train_forward_blend (depth not computed) followed by
 * enerf_composite_rays_train_backward_mse(out_image, target, ...), bit for bit; weights_sum / image / out_image are
 * still written. */
int enerf_composite_rays_train_fwd_bwd_mse(const float* sigmas, const float* rgbs, const float* deltas,
                                           const int32_t* rays, uint32_t M, uint32_t N, float* weights_sum,
                                           float* image, const float* bg_color, uint32_t bg_stride, float bg_scalar,
                                           float* out_image, const float* target, float grad_scale,
                                           const int32_t* counter, float* grad_sigmas, float* grad_rgbs, float* loss,
                                           enerf_stream_t stream);

/* raymarching.cu:807-813  march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma,
 *                                    max_steps, C, H, grid, nears, fars, xyzs, dirs, deltas, perturb) */
int enerf_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                     const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                     uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars,
                     float* xyzs, float* dirs, float* deltas, uint32_t perturb, enerf_stream_t stream);

/* Extension of march_rays for UNINITIALISED xyzs / dirs / deltas of `zero_rows_to` rows (>= n_alive * n_step; 0 =
 * plain enerf_march_rays): every slot a ray does not fill and the alignment rows past the last ray are zero-filled by
 * the march itself (delta == 0 is what ends a ray in composite_rays), so the three torch.zeros of
 * raymarching.py:318-320 become torch.empty. */
int enerf_march_rays_ex(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                        const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                        uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars,
                        float* xyzs, float* dirs, float* deltas, uint32_t perturb, uint32_t zero_rows_to,
                        enerf_stream_t stream);

/* raymarching.cu:903-909  composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas,
 *                                        weights_sum, depth, image)   -- in place */
int enerf_composite_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, float* rays_t,
                         const float* sigmas, const float* rgbs, const float* deltas, float* weights_sum,
                         float* depth, float* image, enerf_stream_t stream);

/* Whole-frame inference compositing -- the reference's inference loop (nerf/renderer.py:364-401: per round
 * compact_rays -> alive_counter.item() -> march_rays(n_step) -> network -> composite_rays) needs the rounds only because
 * it marches a fixed n_step slots per alive ray.  With every ray's samples marched contiguously up front
 * (enerf_march_rays_train_count / _write: rays = (id, offset, count); same cells, same lattice of t, perturb = 0) one
 * pass composites a ray exactly as the rounds do: the arithmetic and order of composite_rays (raymarching.cu:817-909:
 * T = 1 - weights_sum, termination once a sample's pre-sample T < 1e-5 -- that sample is still accumulated), starting
 * from t = nears[id], then image += (1 - weights_sum) * bg and depth = clamp(depth - near, 0) / (far - near).  The image
 * equals the iterated schedule's bit for bit whenever no ray reaches the loop's 1024-step cap.  bg: scalar (bg_color NULL),
 * one RGB (stride 0) or one per ray (stride 3).  used_samples (device u32, may be NULL) += samples accumulated. */
int enerf_composite_rays_frame(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays,
                               uint32_t N, uint32_t M, const float* nears, const float* fars, const float* bg_color,
                               uint32_t bg_stride, float bg_scalar, float* weights_sum, float* depth, float* image,
                               uint32_t* used_samples, enerf_stream_t stream);

/* raymarching.cu:933-939  compact_rays(n_alive, rays_alive, rays_alive_old, rays_t, rays_t_old, alive_counter)
 * Order-preserving (stable) here; alive_counter[0] += number of survivors. */
int enerf_compact_rays(uint32_t n_alive, int32_t* rays_alive, const int32_t* rays_alive_old, float* rays_t,
                       const float* rays_t_old, int32_t* alive_counter, enerf_stream_t stream);

/* ------------------------------------------------------------------ gridencoder
 * gridencoder/src/gridencoder.cu:416-471.  `inputs` is always fp32 (gridencoder.cu:437);
 * embeddings / outputs / dy_dx / grad share `dtype` (ENERF_F32 or ENERF_F16).
 * out_layout 0 = the reference's [L,B,C]; 1 = [B,L*C] written directly (saves the permute copy of
 * gridencoder/grid.py:52,70 -- used by enerf_amd's own wrapper, not by the reference's); 2 = [L,Bp,C] with
 * Bp = B rounded up to a multiple of 32 and the pad rows zero-filled (the level-major input of enerf_mlp32_*).
 * in_add / in_mul: the kernels read every input coordinate as (x + in_add) * in_mul -- (0, 1) for inputs already in
 * [0,1] as the reference passes them, (bound, 1/(2*bound)) to fold the wrapper's normalisation (grid.py:150) in.
 * NOTE with a non-identity transform dy_dx / grad_inputs are with respect to the transformed coordinates. */

/* grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, calc_grad_inputs, dy_dx, gridtype) */
int enerf_grid_encode_forward(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                              uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                              int calc_grad_inputs, void* dy_dx, uint32_t gridtype, int dtype, int out_layout,
                              float in_add, float in_mul, enerf_stream_t stream);

/* grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, calc_grad_inputs,
 *                      dy_dx, grad_inputs, gridtype); grad_embeddings must be zero-filled (grid.py:72);
 * grad_layout as out_layout above. */
int enerf_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                               void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                               uint32_t H, int calc_grad_inputs, const void* dy_dx, void* grad_inputs,
                               uint32_t gridtype, int dtype, int grad_layout, float in_add, float in_mul,
                               enerf_stream_t stream);

/* grid_encode_forward over the query points of a full density-grid sweep (NeRFRenderer.update_extra_state,
 * nerf/renderer.py:484-512: one uniformly drawn point inside every cell of every cascade, x fastest) WITHOUT an input
 * array: point b = (cascade b / grid_size^3, cell b % grid_size^3) is generated inside the kernel from `seed` -- the
 * same points, bit for bit, as enerf_density_grid_cells(density_grid = NULL, ..., seed) writes out (csrc/sweep_points.h).
 * fp32 tables, D = 3; outputs as enerf_grid_encode_forward with B = n_cascades * grid_size^3. */
int enerf_grid_encode_forward_sweep(const void* embeddings, const int32_t* offsets, void* outputs, uint32_t n_cascades,
                                    uint32_t grid_size, float bound, uint64_t seed, uint32_t C, uint32_t L, float S,
                                    uint32_t H, uint32_t gridtype, int out_layout, float in_add, float in_mul,
                                    enerf_stream_t stream);

/* Deferred flush: the table gradient's only consumer in training is the optimizer (main_nerf.py:211, Adam), which streams
 * over the whole table anyway.  With flags bit 0 set, enerf_grid_encode_backward_ex leaves the binned levels' record
 * lists pending instead of summing them into `grad_embeddings` (levels too small to bin, and batches below the binning
 * threshold, still add into it); further calls with the flag append to the same lists (`reserve_B`: samples of ALL calls
 * of the session, which sizes the lists; 0 = this call's B).  enerf_grid_adam_from_records then visits every 128-KiB
 * tile of the table once: sums the tile's records in LDS (fp64), adds the dense gradient where one was written (and
 * clears it), and applies torch.optim.Adam's update (no weight decay / amsgrad; `step` counts from 1) to the tile's
 * rows of p, m, v.  With nothing pending it is a plain fused Adam over the dense gradient.  Until it runs, only calls
 * that join the session are accepted.  flags == 0 is exactly enerf_grid_encode_backward. */
int enerf_grid_encode_backward_ex(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                                  void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                  uint32_t H, int calc_grad_inputs, const void* dy_dx, void* grad_inputs,
                                  uint32_t gridtype, int dtype, int grad_layout, float in_add, float in_mul,
                                  uint32_t flags, uint32_t reserve_B, enerf_stream_t stream);
int enerf_grid_adam_from_records(float* p, float* g, float* m, float* v, const int32_t* offsets, uint32_t L, uint32_t C,
                                 float lr, float beta1, float beta2, float eps, uint32_t step, enerf_stream_t stream);
/* The same with up to 8 further (small, dense-gradient) fp32 parameters updated in the same launch -- the model's MLP
 * weights: arrays of n_small device pointers / element counts / learning rates / step counts (host memory); their
 * gradients are read, not cleared. */
int enerf_grid_adam_from_records_ex(float* p, float* g, float* m, float* v, const int32_t* offsets, uint32_t L, uint32_t C,
                                    float lr, float beta1, float beta2, float eps, uint32_t step, uint32_t n_small,
                                    float* const* sp, const float* const* sg, float* const* sm, float* const* sv,
                                    const uint32_t* sn, const float* slr, const uint32_t* sstep, enerf_stream_t stream);
/* Sharded data-parallel tail ("scaling risk (b)"): elements [lo, hi) of the flat table -- multiples of 4,
 * this rank's slice of the reduce-scatter -- and `rec_scale` = 1 / ranks.  While a range is set (hi > lo):
 *   enerf_grid_encode_backward_ex(flags bit 0) keeps the record lists of the tiles wholly inside the range pending and
 *     flushes every other list into the dense gradient at once (that part travels through the reduce-scatter);
 *   enerf_grid_adam_from_records(_ex) updates the range only: rec_scale x (dense gradient -- what a SUM reduce-scatter
 *     delivered -- + this rank's own lists, summed in LDS as on one GPU); the dense gradient is cleared everywhere.
 * lo == hi clears the range.  The caller reduce-scatters the dense gradient between the two calls and all-gathers the
 * parameters after the second. */
int enerf_grid_owner_range(uint64_t lo, uint64_t hi, float rec_scale);
/* Abandon a pending deferred flush (error recovery: the record lists are emptied, nothing is applied). */
int enerf_grid_records_discard(enerf_stream_t stream);

/* ------------------------------------------------------------------ shencoder
 * shencoder/src/shencoder.cu:402-441; `dtype` ENERF_F32 or ENERF_F16 for every tensor. */

/* sh_encode_forward(inputs, outputs, B, D, C, calc_grad_inputs, dy_dx);  D must be 3, 1 <= C <= 8 */
int enerf_sh_encode_forward(const void* inputs, void* outputs, uint32_t B, uint32_t D, uint32_t C,
                            int calc_grad_inputs, void* dy_dx, int dtype, enerf_stream_t stream);

/* extension: fp32 forward, no Jacobian, rows of `outputs` out_stride floats apart (e.g. a 16-column slice of a
 * [B,32] buffer; 16-byte aligned rows when degree^2 % 4 == 0) */
int enerf_sh_encode_forward_strided(const float* inputs, float* outputs, uint32_t B, uint32_t C, uint32_t out_stride,
                                    enerf_stream_t stream);

/* sh_encode_backward(grad, inputs, B, D, C, dy_dx, grad_inputs); accumulates into grad_inputs */
int enerf_sh_encode_backward(const void* grad, const void* inputs, uint32_t B, uint32_t D, uint32_t C,
                             const void* dy_dx, void* grad_inputs, int dtype, enerf_stream_t stream);

/* ------------------------------------------------------------------ ffmlp
 * ffmlp/src/ffmlp.cu:632-895.  16-bit storage (`dtype` ENERF_F16 as the reference, or ENERF_BF16),
 * fp32 MFMA accumulation (the reference accumulates in fp16).  Weight blob layout
 * [W_in hid*in | W_h (k-1)*hid*hid | W_out out*hid], each W[out][in] row-major, y = x W^T, no bias.
 * B must be a multiple of 128 (ffmlp/ffmlp.py:157-159 pads), hidden_dim == 64, input_dim in {16,32,48,64},
 * output_dim == 16 (ffmlp.py:112-121 pads to 16), activation relu(0)/none(6). */

int enerf_ffmlp_forward(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                        uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                        void* forward_buffer, void* outputs, int dtype, enerf_stream_t stream);

int enerf_ffmlp_inference(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim,
                          uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation,
                          uint32_t output_activation, void* inference_buffer, void* outputs, int dtype,
                          enerf_stream_t stream);

/* grad_weights (same 16-bit dtype, zero-filled by the caller) receives the batch-summed weight gradient;
 * backward_buffer [k,B,hid] receives the per-layer activation gradients as in the reference. */
int enerf_ffmlp_backward(const void* grad, const void* inputs, const void* weights, const void* forward_buffer,
                         uint32_t B, uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim,
                         uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                         int calc_grad_inputs, void* backward_buffer, void* grad_inputs, void* grad_weights,
                         int dtype, enerf_stream_t stream);

/* 1 (default): enerf_ffmlp_forward / enerf_ffmlp_backward take the recomputing data flow where the shape allows (input_dim
 * 32, two or three hidden layers -- the two nets of nerf/network_ff.py --, ReLU / no hidden activation, no output
 * activation): the backward recomputes the hidden activations from the inputs (bit-identical to the forward's) and
 * `forward_buffer` / `backward_buffer` -- scratch to the reference's wrapper, ffmlp/ffmlp.py:34-83 -- are neither written nor
 * read: 160 B per sample instead of ~1.4 KB through HBM.  0: the buffered kernels (ffmlp.cu:410-518,711-895's data flow).
 * Forward and backward of a batch must run under the same setting.  Returns the previous setting; < 0 only queries. */
int enerf_ffmlp_recompute(int on);
/* ffmlp.cu:723-743: the reference (re)creates its split-K side streams here.  This implementation reduces weight
 * gradients inside the fused backward kernel, so these only (re)size the fp32 partial-sum workspace. */
int enerf_allocate_splitk(size_t size);
int enerf_free_splitk(void);

/* ------------------------------------------------------------------ extensions beyond the reference's native surface
 * (callers of the hot path: the nn.Linear MLPs of nerf/network.py:40-77 and the optimizer of main_nerf.py:211) */

/* Fused fp32 MLP on v_mfma_f32_32x32x2_f32 (exact fp32 fma chains): X -> 64-wide hidden x num_hidden (1..3) ->
 * Y [B,out_dim <= 32], no bias; W = [W0 64x32 | Wh (num_hidden-1)x64x64 | Wout out_dim x 64], each W[out][in].
 * B is ragged; Bp = B rounded up to a multiple of 32.  x_layout 0: X is [B,32] row-major; x_layout 1: X is the
 * level-major [16,Bp,2] tensor enerf_grid_encode_forward writes with out_layout 2 (column k = 2*level + c).
 * fb (num_hidden * Bp * 64 floats) receives the post-activation hidden states in the library's own tile order -- opaque
 * to the caller, read back only by enerf_mlp32_backward (NULL = inference).
 * activation: relu (0) / none (6); output_activation additionally sigmoid (3).  Rows of Y are y_stride floats apart
 * (0 = out_dim), so the result can land in a slice of a wider buffer; y0_exp (optional, [B]) receives
 * exp(Y[:,0]) -- the trunc_exp forward of the density column (activation.py:5-17); Y may be NULL when only y0_exp is
 * wanted (density-grid updates). */
/* The whole network of nerf/network_ff.py (network_ff.py:60-88) as one inference kernel on the matrix cores:
 *   sigma_net = FFMLP(32 -> 64 -> 64 -> 16), sigma = exp(h[0]), geo_feat = h[1:16];
 *   color_net = FFMLP([SH(4) 16 | geo_feat 15 | 0] -> 64 -> 64 -> 64 -> 16), rgb = sigmoid(h[0:3]).
 * feats: the grid encoder's level-major fp32 output [16, Mp, 2], Mp = M rounded up to 32, pad rows zero
 * (enerf_grid_encode_forward, out_layout 2); dirs [M,3] fp32; w_sigma / w_color: the two FFMLP `weights` parameters as
 * they are (fp32, [W0 64x32 | Wh 64x64 x (layers-1) | Wout 16x64], 7168 and 11264 floats), converted to `dtype`
 * (ENERF_BF16 / ENERF_F16) on the way into LDS.  sigma [M], rgb [M,3] fp32, holding the 16-bit-rounded values the
 * op-by-op route produces (16-bit net outputs; exp / sigmoid in fp32 on the rounded value, rounded again). */
int enerf_ffnerf_inference(const float* feats, const float* dirs, const float* w_sigma, const float* w_color, uint32_t M,
                           int dtype, float* sigma, float* rgb, enerf_stream_t stream);

/* enerf_mlp32_forward / _backward with the weights (and weight gradients) where the caller keeps them, instead of one
 * packed blob: wseg / dwseg = {first layer [64, w0_cols], hidden 0 [64,64], hidden 1 [64,64], output layer
 * [out_dim,64]} (unused hidden slots NULL).  nerf_perm != 0 (w0_cols = 31): the first layer is color_net[0].weight of
 * nerf/network.py:95 as it is -- memory columns [SH 16 | geo_feat 15] -- while the kernels' input rows are [raw density
 * | geo_feat 15 | SH 16]; the permutation and the zero column are applied while the weights are staged into LDS, and
 * undone when the weight gradient is written.  overwrite != 0: dwseg receives the gradient (may be uninitialised). */
int enerf_mlp32_forward_p(const float* X, const float* const* wseg, uint32_t w0_cols, uint32_t nerf_perm, uint32_t B,
                          uint32_t in_dim, uint32_t out_dim, uint32_t num_hidden, uint32_t activation,
                          uint32_t output_activation, float* fb, float* Y, uint32_t x_layout, uint32_t y_stride,
                          float* y0_exp, const float* sh_dirs, enerf_stream_t stream);
int enerf_mlp32_backward_p(const float* dY, const float* X, const float* const* wseg, float* const* dwseg,
                           uint32_t w0_cols, uint32_t nerf_perm, uint32_t overwrite, const float* fb, uint32_t B,
                           uint32_t in_dim, uint32_t out_dim, uint32_t num_hidden, uint32_t activation, float* bb,
                           float* dX, uint32_t x_layout, uint32_t dy_stride, const float* y_sigmoid,
                           uint32_t y_sigmoid_stride, const float* dsigma, const float* h0, uint32_t h0_stride,
                           enerf_stream_t stream);

/* The two networks of nerf/network.py:104-132 as ONE launch per direction (csrc/nerf_mlp.hip; replaces the
 * sigma_net / color_net loops and the torch glue between them, nerf/network.py:108-130):
 *   h = sigma_net(feats);  sigma = exp(h[0]);  rgb = sigmoid(color_net([SH4(dirs) | h[1:16]]))
 * feats: the grid encoder's level-major fp32 output [16, Bp, 2] (enerf_grid_encode_forward, out_layout 2; Bp = B rounded
 * up to 32, pad rows zero), dirs [B,3]; wseg_s = {sigma_net[0].weight [64,32], NULL, NULL, sigma_net[1].weight [16,64]},
 * wseg_c = {color_net[0].weight [64,w0_cols_c] with memory columns [SH 16 | geo_feat 15], color_net[1].weight [64,64],
 * NULL, color_net[2].weight [out_c,64]}; sigma [B], rgb [B,out_c] (out_c <= 16).  The sigma net's 16 outputs never leave
 * the registers: they are the first half of the colour net's input as they stand, the SH basis is evaluated in registers.
 * Arithmetic: split-bf16 (enerf_mlp32_precision 1) -- enerf_nerf_mlp_available() says whether the process's current
 * arithmetic is served (callers fall back to enerf_mlp32_*_p otherwise).  Both calls honour enerf_mlp32_valid_rows; the
 * backward recomputes both forwards, honours enerf_mlp32_signal_next_reduce and writes (overwrite != 0) or adds the weight
 * gradients through dwseg_* (same shapes as wseg_*), the feature gradient level-major into dfeat [16, Bp, 2] (pad rows
 * zero: ready for enerf_grid_encode_backward with grad_layout 2).  g_sigma is multiplied by sigma_scale on the fly.
 * flags bit 0: the operand fragments an earlier call built from these very weight tensors are still current (no
 * optimizer step since) -- skips the 44-wavefront rebuild.  Backward, flags bit 1 (library-internal: enerf_train_step_mse
 * sets it): the weight gradients stay per-workgroup partial sums in the library's workspace for the optimizer launch that
 * follows; dwseg_* are not written by this call. */
int enerf_nerf_mlp_available(void);
int enerf_debug_nerf_mlp_fused(int on);
/* Testing aid: the operand fragments (44 x 2048 bytes) as the last build left them, copied to device memory `dst`. */
int enerf_debug_nerf_frags_copy(void* dst, enerf_stream_t stream);
int enerf_nerf_mlp_forward(const float* feats, const float* dirs, const float* const* wseg_s, const float* const* wseg_c,
                           uint32_t w0_cols_c, uint32_t B, uint32_t out_c, float* sigma, float* rgb, uint32_t flags,
                           enerf_stream_t stream);
int enerf_nerf_mlp_backward(const float* g_rgb, const float* g_sigma, float sigma_scale, const float* feats,
                            const float* dirs, const float* rgb, const float* const* wseg_s, const float* const* wseg_c,
                            float* const* dwseg_s, float* const* dwseg_c, uint32_t w0_cols_c, uint32_t overwrite,
                            uint32_t B, uint32_t out_c, float* dfeat, uint32_t flags, enerf_stream_t stream);

/* enerf_mlp32_forward that also writes the degree-4 SH encoding of sh_dirs [B,3] (shencoder.cu:27-128) into columns
 * 16..31 of each row of Y: for the sigma net of nerf/network.py, whose output row is the colour net's input row
 * (num_hidden 1, level-major X, out_dim <= 16, y_stride >= 32).  sh_dirs == NULL: plain enerf_mlp32_forward. */
int enerf_mlp32_forward_sh(const float* X, const float* W, uint32_t B, uint32_t in_dim, uint32_t out_dim,
                           uint32_t num_hidden, uint32_t activation, uint32_t output_activation, float* fb, float* Y,
                           uint32_t x_layout, uint32_t y_stride, float* y0_exp, const float* sh_dirs,
                           enerf_stream_t stream);

/* dY [B,out_dim] with rows dy_stride floats apart (0 = out_dim); bb [num_hidden,Bp,64] is scratch (written); dX NULL
 * or laid out like X (x_layout 1: pad rows are written as zeros, ready for enerf_grid_encode_backward with
 * grad_layout 2); dW (fp32 blob) is accumulated (+=).  Optional fused epilogue gradients:
 *   y_sigmoid (rows y_sigmoid_stride apart): the forward's sigmoid output; dY is then the gradient of the sigmoid's
 *     output and (dY * (1 - y)) * y is back-propagated;
 *   dsigma [B] + h0 (rows h0_stride apart): the gradient of output column 0 is dsigma * exp(clamp(h0, -15, 15))
 *     (trunc_exp backward) instead of dY[:,0]. */
int enerf_mlp32_backward(const float* dY, const float* X, const float* W, const float* fb, uint32_t B, uint32_t in_dim,
                         uint32_t out_dim, uint32_t num_hidden, uint32_t activation, float* bb, float* dX, float* dW,
                         uint32_t x_layout, uint32_t dy_stride, const float* y_sigmoid, uint32_t y_sigmoid_stride,
                         const float* dsigma, const float* h0, uint32_t h0_stride, enerf_stream_t stream);

/* Rows of a training batch that are real: `device_count` points at a device int32 (the march's counter[0]); the
 * enerf_mlp32_* calls that follow skip the tiles beyond min(*device_count, B) -- the sample budget's padding, which no
 * ray covers (forward: left unwritten; fused backward: zero input gradients) -- until it is set again; NULL: every row
 * is real. */
int enerf_mlp32_valid_rows(const int32_t* device_count);
/* The same with real rows = base + min(*device_count, cap): two renders' samples as one batch (the first render's M rows,
 * its padding included, then the second's counter capped at its own M).  cap == 0: as enerf_mlp32_valid_rows. */
int enerf_mlp32_valid_rows_ex(const int32_t* device_count, uint32_t base, uint32_t cap);
/* Arithmetic of the enerf_mlp32_* kernels (the nn.Linear nets of nerf/network.py:40-77 are fp32):
 *   0  v_mfma_f32_32x32x2_f32: every dot product an fp32 fmaf chain, bit-comparable with an fp32 GEMM;
 *   1  (default) split-bf16: every fp32 operand as bf16 hi + lo, three bf16 MFMA products per fp32 product (hi*hi +
 *      hi*lo + lo*hi), fp32 accumulation: ~2^-16 relative per product, inside the path's 1e-4, at 3/16 of the fp32
 *      MFMA's pipe time.  Applies to forward, dgrad and wgrad (the fused backward; nets with three hidden layers or
 *      more than 16 outputs keep mode 0 in the backward).
 *   2  bf16 operands: inputs, weights and each layer's activations / activation gradients rounded to bf16, one MFMA
 *      product, fp32 accumulation, outputs (and exp / sigmoid of them) rounded to bf16 -- the arithmetic of the FFMLP
 *      nets (enerf_ffmlp_*, ffmlp/src/ffmlp.cu:410-895) over this family's fp32 buffers and fused epilogues; serves the
 *      fused training step of nerf/network_ff.py (sigma net: two hidden layers + SH epilogue, colour net: three).
 *   3  fp16 operands: mode 2's kernels and rounding points on IEEE half (v_mfma_f32_32x32x16_f16; trunc_exp of the density
 *      head stays in fp32, activation.py:5-17 `cast_inputs=torch.float`) -- the arithmetic autocast(float16) gives the
 *      nn.Linear nets in the reference's `fp16 = True` regime (nerf/utils.py:964-975).  fp16's range is why that regime
 *      scales its loss: use with enerf_amp_begin / enerf_amp_end.
 * Returns the previous mode (NOT a status); a negative `mode` only queries. */
int enerf_mlp32_precision(int mode);
/* 1 (default): in modes 1 / 2 the fused backward recomputes the hidden activations from X with the forward's own
 * instruction sequence (bit-identical values and ReLU masks) and the training forward does not store them -- `fb` is
 * then neither written nor read (two thirds of these kernels' HBM traffic at the training batch).  0: the forward
 * stores them and the backward loads them (the reference's data flow, ffmlp.cu:711-895 / autograd's saved tensors).
 * Forward and backward of a batch must run under the same setting.  Nets with three hidden layers (16-bit operands) recompute
 * under either setting.  Returns the previous setting (NOT a status);
 * a negative `on` only queries. */
int enerf_mlp32_recompute(int on);
/* Testing aid: 1 (default) lets enerf_mlp32_backward use its fused dgrad + wgrad kernel (num_hidden <= 2; `bb` is then
 * not written), 0 forces the separate dgrad / wgrad kernels. */
int enerf_debug_mlp32_fused_backward(int on);
/* One-shot: the NEXT enerf_mlp32_backward[_p] call on this process leaves its weight-gradient partial sums pending,
 * and the call after it reduces both networks' sums in a single launch (a network's two MLPs are always run
 * back to back: one reduce launch and one launch gap less per step).  The pending sums live in their own workspace;
 * nothing is written to the first call's dW until the second call. */
int enerf_mlp32_defer_reduce(int on);
/* Cross-stream hand-over without an event record in the launching stream: once armed (on = 1), the next weight-gradient
 * reduce launch of enerf_mlp32_backward[_p] carries a completion signal; enerf_stream_wait_mlp32_signal makes `stream`
 * wait for that launch (and, the launching stream being in order, for everything queued before it).  Error if no
 * launch has carried the signal since it was armed. */
int enerf_mlp32_signal_next_reduce(int on);
int enerf_stream_wait_mlp32_signal(enerf_stream_t stream);
/* Loss scaling of the reference's fp16 regime (nerf/utils.py:964-975: scaler.scale(loss).backward(); scaler.step(optimizer);
 * scaler.update()) around a closed-form step, on the device.  `scale` (fp32) and `growth_tracker` (int32) are the
 * torch.amp.GradScaler's own device tensors, `found_inf` and `skipped` two zero-initialised uint32 words kept by the caller.
 * Between enerf_amp_begin and enerf_amp_end: the compositing backward multiplies the loss gradient by *scale; the MLP
 * weight-gradient reduce launch raises *found_inf when a sum is not finite; enerf_grid_adam_from_records(_ex) divides the
 * gradients by *scale, counts its step as (step - *skipped) and leaves the parameters alone when *found_inf is set.
 * enerf_amp_end queues GradScaler.update() (backoff after a non-finite step, growth after growth_interval clean ones),
 * counts a skipped step in *skipped, clears *found_inf and disarms; enerf_amp_cancel only disarms.  No host
 * synchronisation.  Use with enerf_mlp32_precision(3) (fp16 operands). */
int enerf_amp_begin(float* scale, int32_t* growth_tracker, uint32_t* found_inf, uint32_t* skipped);
int enerf_amp_end(float growth_factor, float backoff_factor, int32_t growth_interval, enerf_stream_t stream);
int enerf_amp_cancel(void);
/* 1 while armed.  Only the kernels named above unscale and skip: the record-list Adam launch (table + up to 8 small tensors).
 * The table's gradient itself is not inspected for non-finite values -- it is dL/dfeature (whose overflow the first layer's
 * weight gradient, which IS inspected, shares) times trilinear weights <= 1. */
int enerf_amp_armed(void);
/* Tuning aid: number of workgroups (= partial weight-gradient sums) enerf_mlp32_backward launches; 0 restores the
 * default (768 for one hidden layer, 512 otherwise). */
int enerf_debug_mlp32_wgrad_blocks(uint32_t blocks);
/* tuning aid: largest alive-ray count for which enerf_march_rays runs one wavefront per ray (when n_step < 16) */
int enerf_debug_march_wave_max_rays(uint32_t n);
/* test / measurement aid: 0 switches off the occupied-box test (below) globally */
int enerf_debug_march_clip(int on);
/* While on, enerf_march_rays(_ex) trust the box enerf_occupied_box_update last computed when it was computed for the same
 * bitfield pointer / C / H / bound (the caller vouches that the bitfield has not changed since: a frame's rounds); off
 * (default), a round of 32 768+ rays computes the box for itself and smaller rounds march without it.  A ray emits no
 * sample outside the box, so the walk stops at its far side: same samples, slots and termination. */
int enerf_march_rays_use_box(int on);
/* tuning / test aid: smallest ray count for which the fixed-step enerf_march_rays_train* counts with one thread per ray
 * and a run log (instead of one wavefront per ray and a chunk log); 0 only reads.  Returns the previous value.  The
 * count and the write pass of a batch must see the same setting. */
int enerf_debug_march_thread_min_rays(uint32_t n);
/* Testing aid: 0 switches off the cross-stream ordering of the library's shared workspaces (a stream that is about to
 * use a kernel family's scratch waits for the family's previous user when that was another stream); 1 = default. */
int enerf_debug_workspace_ordering(int on);
/* tuning aid: workgroups of the background training march (enerf_march_rays_train_ex flag bit 1); 0 = one per CU */
int enerf_debug_march_bg_blocks(uint32_t n);
/* tuning aid: workgroup caps of the mlp32 forward and fused-backward grids (0 = built-in defaults) */
int enerf_debug_mlp32_grid_caps(uint32_t fwd_blocks, uint32_t bwd_blocks);

/* Event-pair ray generation (the caller that feeds the event step: EventNeRFDataset.collate, nerf/provider.py:1364-1441,
 * accumulate_evs branch with poses "computed online", + get_event_rays, nerf/utils.py:184-216), one thread per pair:
 *   s = start_draw[k] - no_successor[start_draw[k]];  ns = min(num_successor[s], acc_max_num_evs + 1) (0 = no cap);
 *   e = s + 1 + min(floor(u_end[k] * ns), ns - 1);    pols[k] = pol_cumsum[e + 1] - pol_cumsum[s + 1];
 *   pose(t) for t = events[s].t and events[e].t from the track: R = rot[i] * exp(alpha * rotvec[i]) (scipy Slerp),
 *   translation = cubic tcoef[i] in (t - knots[i]) (interp1d(kind="cubic") as a piecewise polynomial), segment i by
 *   binary search over knots[K], evaluated in double, rounded to fp32; rays: pixel (x, y) of event s -> unit camera
 *   direction -> rays_d = R d, rays_o = translation, at both poses.
 * events [N,4] fp32 rows (x, y, t, polarity) grouped by pixel (enerf_amd/event_sampler.build_event_tables);
 * no_successor u8 [N]; num_successor i64 [N]; pol_cumsum f64 [N+1]; start_draw i64 [M] in [0,N); u_end f64 [M] in [0,1);
 * rot f64 [K,9]; rotvec f64 [K-1,3]; tcoef f64 [K-1,4,3] (highest power first); outputs [M,3] fp32, pols [M] fp32,
 * start/end i64 [M]; *outside_track += pairs whose times fall outside [knots[0], knots[K-1]]. */
int enerf_event_pair_rays(const float* events, const uint8_t* no_successor, const int64_t* num_successor,
                          const double* pol_cumsum, uint32_t N, const int64_t* start_draw, const double* u_end,
                          uint32_t M, uint32_t acc_max_num_evs, const double* knots, const double* rot,
                          const double* rotvec, const double* tcoef, uint32_t K, float fx, float fy, float cx, float cy,
                          float* rays_o1, float* rays_d1, float* rays_o2, float* rays_d2, float* pols,
                          int64_t* start_out, int64_t* end_out, int32_t* outside_track, enerf_stream_t stream);

/* The event loss of Trainer.train_step_events (nerf/utils.py:499-516, C_thres != -1) and its gradient with respect to
 * the two rendered images, one launch: image1 / image2 [N,3] fp32, pols [N] -> delta [N,1] (use_luma) or [N,3],
 * grad_image1 / grad_image2 [N,3] (d (upstream * loss) / d image), loss [1] (may be NULL).  (utils/event_utils.py:23-66:
 * BT.601 luma, lin-log with threshold 20 on the 0..255 scale, or log(max(., log_thres)); without lin-log and with luma
 * the reference evaluates both terms on the first image -- kept.)  Replaces ~50 elementwise / reduction launches of
 * the autograd route per step. */
int enerf_event_loss_fwd_bwd(const float* image1, const float* image2, const float* pols, uint32_t N, uint32_t use_luma,
                             uint32_t linlog, float C_thres, float log_thres, float upstream, float* grad_image1,
                             float* grad_image2, float* delta, float* loss, enerf_stream_t stream);

/* One fused Adam update (torch.optim.Adam semantics, no weight decay / amsgrad) of a contiguous fp32 tensor:
 * reads p, g, m, v once and writes p, m, v (and g = 0 when zero_grad != 0).  `step` counts from 1. */
int enerf_adam_step(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                    uint32_t step, int zero_grad, enerf_stream_t stream);
/* The same update for up to 16 parameters in one launch: arrays of `count` device pointers / sizes / learning rates /
 * step counts (host memory), shared betas and eps. */
int enerf_adam_step_multi(uint32_t count, float* const* p, float* const* g, float* const* m, float* const* v,
                          const size_t* n, const float* lr, const uint32_t* step, float beta1, float beta2, float eps,
                          int zero_grad, enerf_stream_t stream);

/* ------------------------------------------------------------------ one training step as one call (not in the reference)
 * The closed-form RGB step of the nn.Linear / FFMLP networks on one GPU -- render of a batch whose samples have been
 * marched already, MSE against `target`, backward, the hash table's Adam from the backward's record lists together with
 * the MLP weights' Adam, and (optionally) near_far + march_rays_train of the NEXT batch on `side_stream` behind the MLP
 * backward -- issued by the library itself, in the order and with the arguments of the entry points above
 * (grid_encode_forward, mlp32_forward_p x 2, composite_rays_train_fwd_bwd_mse, mlp32_backward_p x 2, near_far_from_aabb,
 * march_rays_train_ex, grid_encode_backward_ex(defer), grid_adam_from_records_ex): results are those of the calls made
 * one by one.  What it removes is the host's work between the launches (nerf/utils.py:575-640 + main_nerf.py:211 run
 * ~60 framework calls per step).  All pointers are device pointers unless noted; every buffer is the caller's.
 * Scalar background (bg_scalar), density_scale 1, fp32 table with L = 16, C = 2, D = 3. */
typedef struct enerf_train_step_args {
    uint32_t struct_bytes;              /* sizeof(enerf_train_step_args): checked */
    int mlp_precision;                  /* enerf_mlp32_precision for this call's MLP launches; < 0: leave as is */
    enerf_stream_t stream, side_stream; /* side_stream only used when next_rays_o != NULL */
    /* this batch: samples from march_rays_train (M rows budgeted, counter[0] real), N rays */
    uint32_t N, M;
    const float *xyzs, *dirs, *deltas;
    const int32_t *rays, *counter;
    const float* target;                /* [N,3] */
    float bg_scalar, grad_scale;        /* d loss / d image = (out_image - target) * grad_scale */
    float* loss;                        /* device scalar the loss value is added into, or NULL */
    /* networks */
    const float* embeddings;            /* == table below */
    const int32_t* offsets;
    float level_scale_log2, bound, inv_two_bound;
    uint32_t base_resolution, gridtype;
    const float* const* wseg_s;         /* host arrays of 4 device pointers, as enerf_mlp32_forward_p takes them */
    const float* const* wseg_c;
    float* const* dwseg_s;              /* where the weight gradients are written (overwrite) */
    float* const* dwseg_c;
    uint32_t nh_s, nh_c, w0_cols_c, out_c;
    /* scratch of the step, all fp32: feats [16,Mp,2], h32 [M,32], fb_s [nh_s,Mp,64], fb_c [nh_c,Mp,64], sigma [M],
     * rgb [M,out_c], weights_sum [N], image [N,3], out_image [N,3], g_sigmas [M], g_rgbs [M,3], dx32 [M,32],
     * dfeat [16,Mp,2] (Mp = M rounded up to 32) */
    float *feats, *h32, *fb_s, *fb_c, *sigma, *rgb, *weights_sum, *image, *out_image, *g_sigmas, *g_rgbs, *dx32, *dfeat;
    /* next batch's march (next_rays_o == NULL: none).  march_flags: enerf_march_rays_train_ex's zero_unwritten bits 0-3,
     * plus bit 4 = keep this march on side_stream (the caller reads its counter back behind it there); without bit 4 the
     * march may ride in this call's own launches on `stream` (enerf_debug_carry_count) and side_stream sees nothing */
    const float *next_rays_o, *next_rays_d, *aabb;
    const uint8_t* bitfield;
    float min_near, dt_gamma;
    uint32_t next_N, next_M, cascade, grid_size, max_steps, perturb, march_flags;
    float *next_nears, *next_fars, *next_xyzs, *next_dirs, *next_deltas;
    int32_t *next_rays, *next_counter;
    /* optimizer: the table (dense gradient buffer for the levels too small to bin, zero-filled, comes back clean) and up
     * to 8 small tensors whose gradients the backward has just written through dwseg_* */
    float *table, *table_grad, *table_m, *table_v;
    float lr, beta1, beta2, eps;
    uint32_t table_step, n_small;
    float* const* small_p;              /* host arrays of n_small entries, as enerf_grid_adam_from_records_ex takes them */
    const float* const* small_g;
    float* const* small_m;
    float* const* small_v;
    const uint32_t* small_n;
    const float* small_lr;
    const uint32_t* small_step;
    /* bit 0: data parallel -- the table's gradient is SUMMED INTO the dense buffer table_grad (grid_encode_backward's
     * own flush, no record lists left behind) and no optimizer runs: the caller averages table_grad and the buffers
     * behind dwseg_* over the ranks (torch.distributed over RCCL: enerf_amd/trainer.py) and steps the optimizer itself.
     * bit 1 (with bit 0, enerf_grid_owner_range set): the sharded tail that keeps this rank's slice as record lists -- the
     * backward defers, flushing only the other slices into table_grad; the caller reduce-scatters, runs
     * enerf_grid_adam_from_records_ex and all-gathers.
     * (ABI 2: bit 2 and the three fields behind it -- the tail on the library's own RCCL communicator -- are gone.) */
    uint32_t flags, reserved;
} enerf_train_step_args;
int enerf_train_step_mse(const enerf_train_step_args* args);
/* Development aid: host microseconds enerf_train_step_mse spends in each of its calls (in call order, 16 slots, averaged
 * over the steps since timing was switched on); on >= 0 switches the timers (and clears them), on < 0 only reads. */
int enerf_debug_step_timing(int on, double* out16);
/* enerf_train_step_mse builds the fused MLP's operand fragments inside its grid forward's launch (sixteen extra
 * workgroups) instead of a launch of their own; on = 0 switches that off (testing aid; < 0 only reads).  Returns the
 * previous setting. */
int enerf_debug_carry_frags(int on);
/* ... and leaves the fused MLP backward's weight-gradient partial sums to the optimizer's launch, which sums them for its
 * small tensors (no k_mlp32_reduce_w2 launch in between; the sums land in the gradient tensors all the same); on = 0
 * switches that off (testing aid; < 0 only reads).  Returns the previous setting. */
int enerf_debug_fold_reduce(int on);
/* ... and marches the NEXT batch without a second stream: the march's count pass rides in the table optimizer's launch
 * (extra workgroups; enerf_debug_march_carry_blocks sets how many, 0 = two per compute unit), its scan + write are one launch
 * behind it on the step's own stream -- no cross-stream signal behind the MLP backward, no event wait at the head of the next
 * step.  Same samples, same offsets.  Served where the wave-per-ray fixed-step marcher is (dt_gamma = 0, at most 16384 rays)
 * and no count mirror is armed; everything else keeps the side-stream march.  on = 0 switches it off (-1 only reads).
 * Returns the previous setting; on = -2 returns the number of steps marched that way so far. */
int enerf_debug_carry_count(int on);
int enerf_debug_march_carry_blocks(uint32_t blocks);

/* The event-only step (Trainer.train_step_events, nerf/utils.py:482-546, event_only = 1, C_thres != -1) the same way: TWO
 * renders -- the event pairs' rays at the two poses -- blended with one background colour, the event loss on the two
 * images and its gradient (enerf_event_loss_fwd_bwd), both renders' backward, ONE optimizer pass; optionally the two
 * marches of the next step on `side_stream` behind the second render's MLP backward.  Calls, in order (what
 * enerf_amd/events.train_step_events_manual + FusedAdam.step_grid_table issue): per render grid_encode_forward,
 * mlp32_forward_p x 2, composite_rays_train_forward_blend; event_loss_fwd_bwd; per render
 * composite_rays_train_backward_mse(target = NULL), mlp32_backward_p x 2 (the second render's weight gradients are
 * added to the first's), grid_encode_backward_ex(defer, reserve = M1 + M2); grid_adam_from_records_ex. */
typedef struct enerf_step_render {
    uint32_t N, M;                      /* rays, sample rows budgeted (counter[0] real) */
    const float *xyzs, *dirs, *deltas;
    const int32_t *rays, *counter;
    /* scratch, fp32 (shapes as in enerf_train_step_args; g_image [N,3] receives d loss / d out_image) */
    float *feats, *h32, *fb_s, *fb_c, *sigma, *rgb, *weights_sum, *image, *out_image, *g_image, *g_sigmas, *g_rgbs, *dx32,
        *dfeat;
    /* the next step's march of this render's rays (next_rays_o == NULL: none) */
    const float *next_rays_o, *next_rays_d;
    uint32_t next_N, next_M;
    float *next_nears, *next_fars, *next_xyzs, *next_dirs, *next_deltas;
    int32_t *next_rays, *next_counter;
} enerf_step_render;
typedef struct enerf_event_step_args {
    uint32_t struct_bytes;              /* sizeof(enerf_event_step_args): checked */
    int mlp_precision;
    enerf_stream_t stream, side_stream;
    enerf_step_render r[2];
    const float* bg_color;              /* [3] device floats: one colour for both renders (nerf/utils.py:497) */
    const float* pols;                  /* [N] */
    uint32_t use_luma, linlog;
    float C_thres, log_thres, upstream;
    float *delta, *loss;                /* [N,1] (use_luma) or [N,3]; device scalar (written, may be NULL) */
    /* networks, march parameters, optimizer: as in enerf_train_step_args */
    const float* embeddings;
    const int32_t* offsets;
    float level_scale_log2, bound, inv_two_bound;
    uint32_t base_resolution, gridtype;
    const float* const* wseg_s;
    const float* const* wseg_c;
    float* const* dwseg_s;
    float* const* dwseg_c;
    uint32_t nh_s, nh_c, w0_cols_c, out_c;
    const float* aabb;
    const uint8_t* bitfield;
    float min_near, dt_gamma;
    uint32_t cascade, grid_size, max_steps, perturb, march_flags, reserved0;
    float *table, *table_grad, *table_m, *table_v;
    float lr, beta1, beta2, eps;
    uint32_t table_step, n_small;
    float* const* small_p;
    const float* const* small_g;
    float* const* small_m;
    float* const* small_v;
    const uint32_t* small_n;
    const float* small_lr;
    const uint32_t* small_step;
    /* flags bit 1: MERGED layout.  Both renders have the same M, r[1].xyzs / dirs / deltas are the M rows that follow
     * r[0]'s in one buffer, and the m_* pointers hold scratch for 2 M rows (shapes of the per-render scratch with 2 M
     * for M).  Then grid_encode_forward, the four mlp32 launches, grid_encode_backward run ONCE over the 2 M rows (rows
     * [counter0, M) are padding whose gradients the first render's composite backward zero-fills; the MLP kernels take
     * M + min(counter1, M) as their valid-row count: enerf_mlp32_valid_rows_ex); compositing and its backward stay per
     * render, on the halves.  Same per-sample values; the weight gradients are summed over both renders in one pass
     * instead of two passes added. */
    uint32_t flags, reserved;
    float *m_feats, *m_h32, *m_fb_s, *m_fb_c, *m_sigma, *m_rgb, *m_g_sigmas, *m_g_rgbs, *m_dx32, *m_dfeat;
} enerf_event_step_args;
int enerf_train_step_events(const enerf_event_step_args* args);

/* profiling aid: restrict grid_encode_forward/backward to the levels whose bit is set (default all) */
int enerf_debug_grid_level_mask(uint32_t mask);
/* Testing / profiling aid: fp32 grid_encode_backward batches of at least `min_batch` samples send the levels spanning
 * at least `min_tiles` 128-KiB tiles through the binned path (per-tile record lists summed in LDS, no global float
 * atomics); everything else takes the global-atomic kernel.  Defaults 16384 / 8. */
int enerf_debug_grid_bwd_binned(uint32_t min_batch, uint32_t min_tiles);

/* ------------------------------------------------------------------ measurement hooks (not in the reference)
 * When enabled, every launch of the selected kernel family is bracketed by hipEvents on its own stream so that
 * bench.py can report the average launch duration of the dominant kernel (roofline.achieved). */
#define ENERF_K_GRID_FWD 0
#define ENERF_K_GRID_BWD 1
#define ENERF_K_MARCH_TRAIN 2
#define ENERF_K_COMPOSITE_FWD 3
#define ENERF_K_COMPOSITE_BWD 4
#define ENERF_K_SH_FWD 5
#define ENERF_K_FFMLP_FWD 6
#define ENERF_K_FFMLP_BWD 7
#define ENERF_K_MARCH_INFER 8
#define ENERF_K_COMPOSITE_INFER 9
#define ENERF_K_TABLE_ADAM 10 /* enerf_grid_adam_from_records[_ex]: table flush + table Adam (+ the small tensors) */
#define ENERF_K_MLP_REDUCE 11 /* the weight-gradient reduce launch of enerf_mlp32_backward[_p] */
#define ENERF_K_COUNT 12

/* Measurement aid: samples reserved (counter[0] increments) by all enerf_march_rays_train[_ex] calls of this process
 * since the last reset, kept on the device by the march's own scan pass.  Synchronises `stream`, then reads (total may
 * be NULL) and optionally resets. */
int enerf_march_train_samples(uint64_t* total, int reset, enerf_stream_t stream);
int enerf_prof_enable(int on);
/* bit k of `mask` enables timing of kernel family k only (each timed call costs two event records on the stream) */
int enerf_prof_enable_mask(uint32_t mask);
int enerf_prof_reset(void);
/* Time one eligible call in `n` per kernel family (default 1: every call).  A timed launch carries completion signals
 * (and, for kernel-stamped families, a one-wavefront marker launch before it): ~10 us of queue time per timed call, so a
 * benchmark that must not perturb what it measures samples.  enerf_prof_read_units: the work units (grid_encode: points;
 * mlp32: samples) of the TIMED calls only -- the denominator that goes with enerf_prof_read's time -- and the number of
 * eligible calls seen since the last reset. [host ptrs] */
int enerf_prof_sample_every(uint32_t n);
int enerf_prof_read_units(int kernel_id, double* units, uint64_t* calls_seen);
/* Synchronises the recorded events; returns total milliseconds and launch count for `kernel_id`. [host ptrs] */
int enerf_prof_read(int kernel_id, double* total_ms, uint64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* ENERF_HIP_H */
