// CPython module `_raymarching`: the eleven functions the reference binds in raymarching/src/bindings.cpp:5-20 with the
// prototypes of raymarching/src/raymarching.h:7-19 (same names, argument order, pre-allocated outputs, void returns),
// each forwarding to its C-ABI entry point in libenerf_hip.so.  The reference's raymarching/raymarching.py picks this
// module up unchanged (`import _raymarching as _backend`).  Unlike the reference (which defines its CHECK_* macros but
// never applies them here) the tensors are validated: a CPU or strided tensor raises instead of corrupting memory.
#include "shim_common.h"

using shim::Launch;

static const float* f32(const at::Tensor& t, const char* n) { shim::need_f32(t, n); return t.data_ptr<float>(); }
static float* f32w(at::Tensor& t, const char* n) { shim::need_f32(t, n); return t.data_ptr<float>(); }
static const int32_t* i32(const at::Tensor& t, const char* n) { shim::need_i32(t, n); return t.data_ptr<int32_t>(); }
static int32_t* i32w(at::Tensor& t, const char* n) { shim::need_i32(t, n); return t.data_ptr<int32_t>(); }
static const uint8_t* u8(const at::Tensor& t, const char* n) {
    shim::need_dense(t, n);
    TORCH_CHECK(t.scalar_type() == at::kByte, n, " must be a uint8 tensor");
    return t.data_ptr<uint8_t>();
}

void near_far_from_aabb(at::Tensor rays_o, at::Tensor rays_d, at::Tensor aabb, const uint32_t N, const float min_near,
                        at::Tensor nears, at::Tensor fars) {
    Launch l(rays_o);
    shim::ok(enerf_near_far_from_aabb(f32(rays_o, "rays_o"), f32(rays_d, "rays_d"), f32(aabb, "aabb"), N, min_near,
                                      f32w(nears, "nears"), f32w(fars, "fars"), l.stream), "near_far_from_aabb");
}

void polar_from_ray(at::Tensor rays_o, at::Tensor rays_d, const float radius, const uint32_t N, at::Tensor coords) {
    Launch l(rays_o);
    shim::ok(enerf_polar_from_ray(f32(rays_o, "rays_o"), f32(rays_d, "rays_d"), radius, N, f32w(coords, "coords"),
                                  l.stream), "polar_from_ray");
}

void morton3D(at::Tensor coords, const uint32_t N, at::Tensor indices) {
    Launch l(coords);
    shim::ok(enerf_morton3D(i32(coords, "coords"), N, i32w(indices, "indices"), l.stream), "morton3D");
}

void morton3D_invert(at::Tensor indices, const uint32_t N, at::Tensor coords) {
    Launch l(indices);
    shim::ok(enerf_morton3D_invert(i32(indices, "indices"), N, i32w(coords, "coords"), l.stream), "morton3D_invert");
}

void packbits(at::Tensor grid, const uint32_t N, const float density_thresh, at::Tensor bitfield) {
    Launch l(grid);
    shim::need_dense(bitfield, "bitfield");
    TORCH_CHECK(bitfield.scalar_type() == at::kByte, "bitfield must be a uint8 tensor");
    shim::ok(enerf_packbits(f32(grid, "grid"), N, density_thresh, bitfield.data_ptr<uint8_t>(), l.stream), "packbits");
}

void march_rays_train(at::Tensor rays_o, at::Tensor rays_d, at::Tensor grid, const float bound, const float dt_gamma,
                      const uint32_t max_steps, const uint32_t N, const uint32_t C, const uint32_t H, const uint32_t M,
                      at::Tensor nears, at::Tensor fars, at::Tensor xyzs, at::Tensor dirs, at::Tensor deltas,
                      at::Tensor rays, at::Tensor counter, const uint32_t perturb) {
    Launch l(rays_o);
    shim::ok(enerf_march_rays_train(f32(rays_o, "rays_o"), f32(rays_d, "rays_d"), u8(grid, "grid"), bound, dt_gamma,
                                    max_steps, N, C, H, M, f32(nears, "nears"), f32(fars, "fars"), f32w(xyzs, "xyzs"),
                                    f32w(dirs, "dirs"), f32w(deltas, "deltas"), i32w(rays, "rays"),
                                    i32w(counter, "counter"), perturb, l.stream), "march_rays_train");
}

void composite_rays_train_forward(at::Tensor sigmas, at::Tensor rgbs, at::Tensor deltas, at::Tensor rays,
                                  const uint32_t M, const uint32_t N, at::Tensor weights_sum, at::Tensor depth,
                                  at::Tensor image) {
    Launch l(sigmas);
    shim::ok(enerf_composite_rays_train_forward(f32(sigmas, "sigmas"), f32(rgbs, "rgbs"), f32(deltas, "deltas"),
                                                i32(rays, "rays"), M, N, f32w(weights_sum, "weights_sum"),
                                                f32w(depth, "depth"), f32w(image, "image"), l.stream),
             "composite_rays_train_forward");
}

void composite_rays_train_backward(at::Tensor grad_weights_sum, at::Tensor grad_image, at::Tensor sigmas,
                                   at::Tensor rgbs, at::Tensor deltas, at::Tensor rays, at::Tensor weights_sum,
                                   at::Tensor image, const uint32_t M, const uint32_t N, at::Tensor grad_sigmas,
                                   at::Tensor grad_rgbs) {
    Launch l(sigmas);
    shim::ok(enerf_composite_rays_train_backward(f32(grad_weights_sum, "grad_weights_sum"), f32(grad_image, "grad_image"),
                                                 f32(sigmas, "sigmas"), f32(rgbs, "rgbs"), f32(deltas, "deltas"),
                                                 i32(rays, "rays"), f32(weights_sum, "weights_sum"), f32(image, "image"),
                                                 M, N, f32w(grad_sigmas, "grad_sigmas"), f32w(grad_rgbs, "grad_rgbs"),
                                                 l.stream), "composite_rays_train_backward");
}

void march_rays(const uint32_t n_alive, const uint32_t n_step, at::Tensor rays_alive, at::Tensor rays_t,
                at::Tensor rays_o, at::Tensor rays_d, const float bound, const float dt_gamma, const uint32_t max_steps,
                const uint32_t C, const uint32_t H, at::Tensor grid, at::Tensor nears, at::Tensor fars, at::Tensor xyzs,
                at::Tensor dirs, at::Tensor deltas, const uint32_t perturb) {
    Launch l(rays_o);
    shim::ok(enerf_march_rays(n_alive, n_step, i32(rays_alive, "rays_alive"), f32(rays_t, "rays_t"), f32(rays_o, "rays_o"),
                              f32(rays_d, "rays_d"), bound, dt_gamma, max_steps, C, H, u8(grid, "grid"),
                              f32(nears, "nears"), f32(fars, "fars"), f32w(xyzs, "xyzs"), f32w(dirs, "dirs"),
                              f32w(deltas, "deltas"), perturb, l.stream), "march_rays");
}

void composite_rays(const uint32_t n_alive, const uint32_t n_step, at::Tensor rays_alive, at::Tensor rays_t,
                    at::Tensor sigmas, at::Tensor rgbs, at::Tensor deltas, at::Tensor weights_sum, at::Tensor depth,
                    at::Tensor image) {
    Launch l(sigmas);
    shim::ok(enerf_composite_rays(n_alive, n_step, i32(rays_alive, "rays_alive"), f32w(rays_t, "rays_t"),
                                  f32(sigmas, "sigmas"), f32(rgbs, "rgbs"), f32(deltas, "deltas"),
                                  f32w(weights_sum, "weights_sum"), f32w(depth, "depth"), f32w(image, "image"), l.stream),
             "composite_rays");
}

void compact_rays(const uint32_t n_alive, at::Tensor rays_alive, at::Tensor rays_alive_old, at::Tensor rays_t,
                  at::Tensor rays_t_old, at::Tensor alive_counter) {
    Launch l(rays_alive);
    shim::ok(enerf_compact_rays(n_alive, i32w(rays_alive, "rays_alive"), i32(rays_alive_old, "rays_alive_old"),
                                f32w(rays_t, "rays_t"), f32(rays_t_old, "rays_t_old"),
                                i32w(alive_counter, "alive_counter"), l.stream), "compact_rays");
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("packbits", &packbits, "packbits (HIP, gfx950)");
    m.def("near_far_from_aabb", &near_far_from_aabb, "near_far_from_aabb (HIP, gfx950)");
    m.def("polar_from_ray", &polar_from_ray, "polar_from_ray (HIP, gfx950)");
    m.def("morton3D", &morton3D, "morton3D (HIP, gfx950)");
    m.def("morton3D_invert", &morton3D_invert, "morton3D_invert (HIP, gfx950)");
    m.def("march_rays_train", &march_rays_train, "march_rays_train (HIP, gfx950)");
    m.def("composite_rays_train_forward", &composite_rays_train_forward, "composite_rays_train_forward (HIP, gfx950)");
    m.def("composite_rays_train_backward", &composite_rays_train_backward, "composite_rays_train_backward (HIP, gfx950)");
    m.def("march_rays", &march_rays, "march rays (HIP, gfx950)");
    m.def("composite_rays", &composite_rays, "composite rays (HIP, gfx950)");
    m.def("compact_rays", &compact_rays, "compact rays (HIP, gfx950)");
}
