// raymarching.hip -- occupancy-grid ray marching and compositing for gfx950 (wave64).
//
// Replaces raymarching/src/raymarching.cu of the reference (11 entry points, see include/enerf_hip.h).
// Integer outputs (rays, counter, morton codes, bitfield) and the sample positions are bit-exact with
// oracle/enerf_oracle.c; compositing is evaluated with wave-parallel scans (one wavefront per ray) and
// agrees with the sequential recurrence to fp32 round-off.
//
// Compiled with -ffp-contract=off: every fused multiply-add below is an explicit fmaf().
#include <float.h>
#include <cstdlib>

#include "common.h"

using namespace enerf;

namespace {

#include "march_lattice.h"

__global__ void __launch_bounds__(256) k_near_far(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                  const float* __restrict__ aabb, uint32_t N, float min_near,
                                                  float* nears, float* fars) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float near, far;
    (void)near_far_of(rays_o[n * 3], rays_o[n * 3 + 1], rays_o[n * 3 + 2], rays_d[n * 3], rays_d[n * 3 + 1],
                      rays_d[n * 3 + 2], aabb, min_near, near, far);
    nears[n] = near;
    fars[n] = far;
}

__global__ void __launch_bounds__(256) k_polar(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                               float radius, uint32_t N, float* coords) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
    const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
    const float A = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    const float B = fmaf(oz, dz, fmaf(oy, dy, ox * dx));
    const float C = fmaf(oz, oz, fmaf(oy, oy, ox * ox)) - radius * radius;
    const float t = (-B + sqrtf(fmaf(B, B, -(A * C)))) / A;
    const float x = fmaf(t, dx, ox), y = fmaf(t, dy, oy), z = fmaf(t, dz, oz);
    const float theta = atan2f(sqrtf(fmaf(z, z, x * x)), y);
    const float phi = atan2f(z, x);
    coords[n * 2] = fmaf(2 * theta, kRPi, -1.0f);
    coords[n * 2 + 1] = phi * kRPi;
}

__global__ void __launch_bounds__(256) k_morton3D(const int32_t* __restrict__ coords, uint32_t N, int32_t* indices) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    indices[n] = (int32_t)morton3((uint32_t)coords[n * 3], (uint32_t)coords[n * 3 + 1], (uint32_t)coords[n * 3 + 2]);
}

__global__ void __launch_bounds__(256) k_morton3D_invert(const int32_t* __restrict__ indices, uint32_t N,
                                                         int32_t* coords) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int ind = indices[n];
    coords[n * 3 + 0] = (int32_t)morton3_inv((uint32_t)(ind >> 0));
    coords[n * 3 + 1] = (int32_t)morton3_inv((uint32_t)(ind >> 1));
    coords[n * 3 + 2] = (int32_t)morton3_inv((uint32_t)(ind >> 2));
}

// One thread per output byte: two 16-byte loads in, one byte out (a wave stores 64 contiguous bytes).
__global__ void __launch_bounds__(256) k_packbits(const float* __restrict__ grid, uint32_t N, float thresh,
                                                  uint8_t* bitfield) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float4 a = reinterpret_cast<const float4*>(grid)[(size_t)n * 2];
    const float4 b = reinterpret_cast<const float4*>(grid)[(size_t)n * 2 + 1];
    uint32_t bits = 0;
    bits |= a.x > thresh ? 1u : 0u;
    bits |= a.y > thresh ? 2u : 0u;
    bits |= a.z > thresh ? 4u : 0u;
    bits |= a.w > thresh ? 8u : 0u;
    bits |= b.x > thresh ? 16u : 0u;
    bits |= b.y > thresh ? 32u : 0u;
    bits |= b.z > thresh ? 64u : 0u;
    bits |= b.w > thresh ? 128u : 0u;
    bitfield[n] = (uint8_t)bits;
}

__global__ void k_aabb_init(int* __restrict__ keys) {
    if (threadIdx.x < 6) keys[threadIdx.x] = 0x7f7f7f7f;
}

__global__ void __launch_bounds__(256) k_occupied_aabb(const uint8_t* __restrict__ grid, uint32_t C, uint32_t H,
                                                       float bound, int* __restrict__ keys) {
    // one byte = 8 consecutive morton indices = one 2x2x2 block of cells of one level
    const uint32_t bytes_per_level = H * H * H / 8;
    const uint32_t total = C * bytes_per_level;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, nhi[3] = {3.0e38f, 3.0e38f, 3.0e38f};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        if (grid[i] == 0) continue;
        const uint32_t level = i / bytes_per_level;
        const uint32_t m = (i - level * bytes_per_level) * 8u;
        const uint32_t cx = morton3_inv(m), cy = morton3_inv(m >> 1), cz = morton3_inv(m >> 2);
        const float pw = (float)(1u << level);
        const float mb = pw > bound ? bound : pw;
        const float cell = 2.0f * mb / (float)H;
        const uint32_t cc[3] = {cx, cy, cz};
        for (int a = 0; a < 3; a++) {
            lo[a] = fminf(lo[a], fmaf((float)cc[a], cell, -mb));
            nhi[a] = fminf(nhi[a], -fmaf((float)(cc[a] + 2u), cell, -mb));
        }
    }
    for (int a = 0; a < 3; a++) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], off, 64));
            nhi[a] = fminf(nhi[a], __shfl_xor(nhi[a], off, 64));
        }
    }
    // wave -> workgroup (LDS) -> one global atomic per workgroup and key (same-address device atomics cost ~12 ns each)
    __shared__ int s_keys[6];
    if (threadIdx.x < 6) s_keys[threadIdx.x] = 0x7f7f7f7f;
    __syncthreads();
    if (lane_id() == 0) {
        for (int a = 0; a < 3; a++) {
            if (lo[a] < 3.0e38f) atomicMin(&s_keys[a], aabb_key(lo[a]));
            if (nhi[a] < 3.0e38f) atomicMin(&s_keys[3 + a], aabb_key(nhi[a]));
        }
    }
    __syncthreads();
    if (threadIdx.x < 6 && s_keys[threadIdx.x] != 0x7f7f7f7f) atomicMin(keys + threadIdx.x, s_keys[threadIdx.x]);
}

__global__ void __launch_bounds__(256) k_march_count_w(MarchCountJob a) {
    __shared__ float s_face[kTabH + 1];
    __shared__ uint32_t s_expand[kTabH];
    march_count_block(a, blockIdx.x, gridDim.x, s_face, s_expand);
}

// rows [lo, hi) of the three sample buffers <- 0 (thread `tid` of `nthreads`)
__device__ __forceinline__ void zero_rows(float* xyzs, float* dirs, float* deltas, uint32_t lo, uint32_t hi,
                                          uint32_t tid, uint32_t nthreads) {
    for (size_t i = (size_t)lo * 3 + tid; i < (size_t)hi * 3; i += nthreads) { xyzs[i] = 0.0f; dirs[i] = 0.0f; }
    for (size_t i = (size_t)lo * 2 + tid; i < (size_t)hi * 2; i += nthreads) deltas[i] = 0.0f;
}

__global__ void __launch_bounds__(256) k_march_write_w(const float* __restrict__ rays_o,
                                                       const float* __restrict__ rays_d,
                                                       const uint8_t* __restrict__ grid, float bound, uint32_t max_steps,
                                                       uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                                       const float* __restrict__ nears, const float* __restrict__ fars,
                                                       float* xyzs, float* dirs, float* deltas,
                                                       const int32_t* __restrict__ rays, uint32_t perturb,
                                                       const ChunkEntry* __restrict__ log,
                                                       const uint32_t* __restrict__ nlog,
                                                       const int32_t* __restrict__ counter, uint32_t ray_blocks) {
    if (blockIdx.x >= ray_blocks) {
        // zero_unwritten: rows past the last reserved sample (the caller handed over uninitialised buffers)
        const uint32_t used = min((uint32_t)counter[0], M);
        zero_rows(xyzs, dirs, deltas, used, M, (blockIdx.x - ray_blocks) * blockDim.x + threadIdx.x,
                  (gridDim.x - ray_blocks) * blockDim.x);
        return;
    }
    const uint32_t n = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (n >= N) return;
    const uint32_t point_index = (uint32_t)rays[(size_t)n * 3 + 1];
    const uint32_t num_steps = (uint32_t)rays[(size_t)n * 3 + 2];
    if (num_steps == 0) return;
    if (point_index + num_steps >= M) {
        // dropped for lack of room: its reservation, clipped to the buffer, is the other region nobody writes
        if (counter && point_index < M) zero_rows(xyzs, dirs, deltas, point_index, M, lane_id(), 64);
        return;
    }
    RayCtx c;
    ray_ctx_init(c, rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, grid, bound, 0.0f, max_steps, C, H);
    float t0 = nears[n];
    if (perturb) t0 = fmaf(c.dt_min, pcg_first_float((uint64_t)n, 1u), t0);   // contracted by the reference's compiler (:351)
    const uint32_t entries = __builtin_amdgcn_readfirstlane(nlog[n]);
    if (entries != kLogOverflow)
        lattice_replay(c, t0, log + (size_t)n * kLogCap, entries, xyzs + (size_t)point_index * 3,
                       dirs + (size_t)point_index * 3, deltas + (size_t)point_index * 2);
    else
        (void)lattice_march<true>(c, t0, fars[n], num_steps, xyzs + (size_t)point_index * 3,
                                  dirs + (size_t)point_index * 3, deltas + (size_t)point_index * 2);
}

__global__ void __launch_bounds__(64) k_march_count(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                    const uint8_t* __restrict__ grid, float bound, float dt_gamma,
                                                    uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                                    const float* __restrict__ nears, const float* __restrict__ fars,
                                                    int32_t* rays, uint32_t perturb) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    RayCtx c;
    ray_ctx_init(c, rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, grid, bound, dt_gamma, max_steps, C, H);
    float t0 = nears[n];
    if (perturb) t0 = fmaf(c.dt_min, pcg_first_float((uint64_t)n, 1u), t0);   // contracted by the reference's compiler (:351)
    rays[(size_t)n * 3 + 2] = (int32_t)march_one_ray<false>(c, t0, fars[n], max_steps, nullptr, nullptr, nullptr);
}

// samples reserved by all march_rays_train calls since the last reset (enerf_march_train_samples): lets a harness
// report ray-samples/s without a bookkeeping launch per step
__device__ unsigned long long g_train_samples = 0ull;

// `fresh`: the counter is taken as (0, 0) whatever it holds -- the caller's counter.zero_() without its launch
__global__ void __launch_bounds__(1024) k_march_scan(int32_t* rays, int32_t* counter, uint32_t N, uint32_t fresh,
                                                     int32_t* mirror) {
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry_s;
    const int lane = lane_id();
    const int wid = threadIdx.x >> 6;
    const uint32_t c0 = fresh ? 0u : (uint32_t)counter[0];
    if (threadIdx.x == 0) carry_s = c0;
    __syncthreads();
    for (uint32_t base = 0; base < N; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < N ? (uint32_t)rays[(size_t)i * 3 + 2] : 0u;
        const uint32_t incl = wave_incl_scan_add_u32(v, lane);
        if (lane == 63) wave_tot[wid] = incl;
        __syncthreads();
        uint32_t wave_off = 0;
        for (int w = 0; w < wid; w++) wave_off += wave_tot[w];
        const uint32_t carry = carry_s;
        if (i < N) {
            rays[(size_t)i * 3 + 0] = (int32_t)i;
            rays[(size_t)i * 3 + 1] = (int32_t)(carry + wave_off + incl - v);
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + wave_off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        g_train_samples += (unsigned long long)(carry_s - c0);
        counter[0] = (int32_t)carry_s;
        const int32_t n_rays = (fresh ? 0 : counter[1]) + (int32_t)N;
        counter[1] = n_rays;
        if (mirror) {
            // enerf_march_mirror_count: the counter also goes straight into (pinned, device-visible) host memory -- the host
            // that watches the two words sees the count without a copy queued behind this kernel
            __hip_atomic_store(mirror, (int32_t)carry_s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(mirror + 1, n_rays, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Scan + write pass as ONE launch (enerf::march_carry_end: behind a launch that carried the count pass).  No workgroup waits
// for another: every workgroup sums the counts of the rays in front of its own itself (N <= 16384 counts, out of L2: 16
// loads per thread at 4096 rays) -- offsets are those of k_march_scan, integer sums in any order.  The counter is taken as
// (0, 0) before the call (`fresh`, the only form the training step uses): counter = (total, N), rays[n] = (n, offset, count).
__global__ void __launch_bounds__(256) k_march_scan_write_w(const float* __restrict__ rays_o,
                                                            const float* __restrict__ rays_d,
                                                            const uint8_t* __restrict__ grid, float bound,
                                                            uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                                            uint32_t M, const float* __restrict__ nears,
                                                            const float* __restrict__ fars, float* xyzs, float* dirs,
                                                            float* deltas, int32_t* rays, uint32_t perturb,
                                                            const ChunkEntry* __restrict__ log,
                                                            const uint32_t* __restrict__ nlog, int32_t* counter,
                                                            uint32_t zero_unwritten, uint32_t ray_blocks) {
    __shared__ uint32_t s_part[4];
    const bool zero_block = blockIdx.x >= ray_blocks;
    const uint32_t first = zero_block ? N : min(blockIdx.x * 4u, N);          // this workgroup's first ray (zero blocks: all)
    uint32_t acc = 0;
    for (uint32_t i = threadIdx.x; i < first; i += 256u) acc += (uint32_t)rays[(size_t)i * 3 + 2];
    acc = wave_incl_scan_add_u32(acc, lane_id());
    if (lane_id() == 63) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    const uint32_t base = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    if (zero_block) {
        // zero_unwritten: rows past the last reserved sample (the caller handed over uninitialised buffers); base = total
        const uint32_t used = min(base, M);
        zero_rows(xyzs, dirs, deltas, used, M, (blockIdx.x - ray_blocks) * blockDim.x + threadIdx.x,
                  (gridDim.x - ray_blocks) * blockDim.x);
        return;
    }
    const uint32_t w = threadIdx.x >> 6;
    const uint32_t n = __builtin_amdgcn_readfirstlane(first + w);
    // (the workgroup's four counts: every wavefront reads them all -- the last workgroup also needs their sum)
    uint32_t cnt[4];
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) cnt[j] = first + j < N ? (uint32_t)rays[(size_t)(first + j) * 3 + 2] : 0u;
    if (blockIdx.x == ray_blocks - 1u && threadIdx.x == 0) {
        const uint32_t total = base + cnt[0] + cnt[1] + cnt[2] + cnt[3];
        g_train_samples += (unsigned long long)total;
        counter[0] = (int32_t)total;
        counter[1] = (int32_t)N;
    }
    if (n >= N) return;
    uint32_t point_index = base;
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) point_index += j < w ? cnt[j] : 0u;
    const uint32_t num_steps = cnt[w];
    if (lane_id() == 0) {
        rays[(size_t)n * 3 + 0] = (int32_t)n;
        rays[(size_t)n * 3 + 1] = (int32_t)point_index;
    }
    if (num_steps == 0) return;
    if (point_index + num_steps >= M) {
        // dropped for lack of room: its reservation, clipped to the buffer, is the other region nobody writes
        if (zero_unwritten && point_index < M) zero_rows(xyzs, dirs, deltas, point_index, M, lane_id(), 64);
        return;
    }
    RayCtx c;
    ray_ctx_init(c, rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, grid, bound, 0.0f, max_steps, C, H);
    float t0 = nears[n];
    if (perturb) t0 = fmaf(c.dt_min, pcg_first_float((uint64_t)n, 1u), t0);   // contracted by the reference's compiler (:351)
    const uint32_t entries = __builtin_amdgcn_readfirstlane(nlog[n]);
    if (entries != kLogOverflow)
        lattice_replay(c, t0, log + (size_t)n * kLogCap, entries, xyzs + (size_t)point_index * 3,
                       dirs + (size_t)point_index * 3, deltas + (size_t)point_index * 2);
    else
        (void)lattice_march<true>(c, t0, fars[n], num_steps, xyzs + (size_t)point_index * 3,
                                  dirs + (size_t)point_index * 3, deltas + (size_t)point_index * 2);
}

// Large ray counts (a whole 640x480 frame marched at once: 307 200 rays): the one-workgroup scan above would walk 300
// tiles one after the other.  Three launches instead -- per-tile sums, one-workgroup scan of the sums (which also
// updates the counter), per-tile scan + tile offset.  Same offsets (integer sums: order does not matter).
__device__ __forceinline__ uint32_t block_excl_scan_1024(uint32_t v, uint32_t* wave_tot, uint32_t& block_total) {
    const int lane = lane_id();
    const int wid = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan_add_u32(v, lane);
    if (lane == 63) wave_tot[wid] = incl;
    __syncthreads();
    uint32_t wave_off = 0, tot = 0;
    for (int w = 0; w < 16; w++) {
        const uint32_t x = wave_tot[w];
        if (w < wid) wave_off += x;
        tot += x;
    }
    block_total = tot;
    return wave_off + incl - v;
}

__global__ void __launch_bounds__(1024) k_march_scan_tile_sums(const int32_t* __restrict__ rays, uint32_t N,
                                                               uint32_t* __restrict__ tile_sums) {
    __shared__ uint32_t wave_tot[16];
    const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
    const uint32_t v = i < N ? (uint32_t)rays[(size_t)i * 3 + 2] : 0u;
    uint32_t tot;
    (void)block_excl_scan_1024(v, wave_tot, tot);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// tile_sums[ntiles] -> exclusive offsets (in place, counter[0] at entry included); counter += (sum, N)
__global__ void __launch_bounds__(1024) k_march_scan_tiles(uint32_t* tile_sums, uint32_t ntiles, int32_t* counter,
                                                           uint32_t N, uint32_t fresh) {
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry_s;
    const uint32_t c0 = fresh ? 0u : (uint32_t)counter[0];
    if (threadIdx.x == 0) carry_s = c0;
    __syncthreads();
    for (uint32_t base = 0; base < ntiles; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < ntiles ? tile_sums[i] : 0u;
        uint32_t tot;
        const uint32_t excl = block_excl_scan_1024(v, wave_tot, tot);
        const uint32_t carry = carry_s;
        if (i < ntiles) tile_sums[i] = carry + excl;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        g_train_samples += (unsigned long long)(carry_s - c0);
        counter[0] = (int32_t)carry_s;
        const int32_t n_rays = (fresh ? 0 : counter[1]) + (int32_t)N;
        counter[1] = n_rays;
    }
}

__global__ void __launch_bounds__(1024) k_march_scan_apply(int32_t* rays, uint32_t N,
                                                           const uint32_t* __restrict__ tile_offsets) {
    __shared__ uint32_t wave_tot[16];
    const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
    const uint32_t v = i < N ? (uint32_t)rays[(size_t)i * 3 + 2] : 0u;
    uint32_t tot;
    const uint32_t excl = block_excl_scan_1024(v, wave_tot, tot);
    if (i < N) {
        rays[(size_t)i * 3 + 0] = (int32_t)i;
        rays[(size_t)i * 3 + 1] = (int32_t)(tile_offsets[blockIdx.x] + excl);
    }
}

__global__ void __launch_bounds__(64) k_march_write(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                    const uint8_t* __restrict__ grid, float bound, float dt_gamma,
                                                    uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                                    const float* __restrict__ nears, const float* __restrict__ fars,
                                                    float* xyzs, float* dirs, float* deltas,
                                                    const int32_t* __restrict__ rays, uint32_t perturb) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const uint32_t point_index = (uint32_t)rays[(size_t)n * 3 + 1];
    const uint32_t num_steps = (uint32_t)rays[(size_t)n * 3 + 2];
    if (num_steps == 0) return;
    if (point_index + num_steps >= M) return;
    RayCtx c;
    ray_ctx_init(c, rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, grid, bound, dt_gamma, max_steps, C, H);
    float t0 = nears[n];
    if (perturb) t0 = fmaf(c.dt_min, pcg_first_float((uint64_t)n, 1u), t0);   // contracted by the reference's compiler (:351)
    (void)march_one_ray<true>(c, t0, fars[n], num_steps, xyzs + (size_t)point_index * 3,
                              dirs + (size_t)point_index * 3, deltas + (size_t)point_index * 2);
}

// ------------------------------------------------------------------ composite_rays_train (one wavefront per ray)
// Lanes take consecutive samples (coalesced 4/8/12-byte-per-lane loads); transmittance is a wave prefix
// product, the running sums are wave prefix sums, chunk to chunk carries live in every lane.
struct CompCarry {
    float T, t, r, g, b, ws, d;
};

// Processes one chunk of <= 64 samples; returns per-lane inclusive quantities and updates the carry.
__device__ __forceinline__ void comp_chunk(bool active, float sigma, float dl0, float dl1, float c0, float c1,
                                           float c2, int lane, CompCarry& k, float& w, float& T_post, float& r_i,
                                           float& g_i, float& b_i, float& ws_i) {
    const float alpha = active ? 1.0f - __expf(-sigma * dl0) : 0.0f;
    const float om = 1.0f - alpha;
    const float P = wave_incl_scan_mul(om, lane);  // prod_{j<=i} (1 - alpha_j) within the chunk
    const float Pex = wave_prev(P, 1.0f);
    w = alpha * (k.T * Pex);
    T_post = k.T * P;
    const float tt = k.t + wave_incl_scan_add(active ? dl1 : 0.0f, lane);
    r_i = k.r + wave_incl_scan_add(w * c0, lane);
    g_i = k.g + wave_incl_scan_add(w * c1, lane);
    b_i = k.b + wave_incl_scan_add(w * c2, lane);
    ws_i = k.ws + wave_incl_scan_add(w, lane);
    const float d_i = k.d + wave_incl_scan_add(w * tt, lane);
    k.T = wave_bcast(T_post, 63);
    k.t = wave_bcast(tt, 63);
    k.r = wave_bcast(r_i, 63);
    k.g = wave_bcast(g_i, 63);
    k.b = wave_bcast(b_i, 63);
    k.ws = wave_bcast(ws_i, 63);
    k.d = wave_bcast(d_i, 63);
}

// background colour of the blend `image + (1 - weights_sum) * bg`: a scalar (color == nullptr), one RGB (stride 0) or one
// RGB per ray (stride 3)
struct Background {
    const float* color;
    uint32_t stride;
    float scalar;
    __device__ __forceinline__ float at(uint32_t ray, int ch) const {
        return color ? color[(size_t)ray * stride + ch] : scalar;
    }
};

__global__ void __launch_bounds__(256) k_composite_train_fwd(const float* __restrict__ sigmas,
                                                             const float* __restrict__ rgbs,
                                                             const float* __restrict__ deltas,
                                                             const int32_t* __restrict__ rays, uint32_t M, uint32_t N,
                                                             float* weights_sum, float* depth, float* image,
                                                             Background bg, float* out_image) {
    const uint32_t n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (n >= N) return;
    const int lane = lane_id();
    const uint32_t index = (uint32_t)rays[(size_t)n * 3], offset = (uint32_t)rays[(size_t)n * 3 + 1];
    const uint32_t num_steps = (uint32_t)rays[(size_t)n * 3 + 2];
    if (num_steps == 0 || offset + num_steps >= M) {
        if (lane == 0) {
            weights_sum[index] = 0;
            if (depth) depth[index] = 0;
            image[(size_t)index * 3] = 0; image[(size_t)index * 3 + 1] = 0; image[(size_t)index * 3 + 2] = 0;
            if (out_image) {
                out_image[(size_t)index * 3] = bg.at(index, 0);
                out_image[(size_t)index * 3 + 1] = bg.at(index, 1);
                out_image[(size_t)index * 3 + 2] = bg.at(index, 2);
            }
        }
        return;
    }
    CompCarry k = {1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (uint32_t s0 = 0; s0 < num_steps; s0 += 64) {
        const uint32_t s = s0 + lane;
        const bool active = s < num_steps;
        const size_t p = (size_t)offset + (active ? s : 0);
        const float sigma = sigmas[p];
        const float2 dl = reinterpret_cast<const float2*>(deltas)[p];
        const float c0 = rgbs[p * 3], c1 = rgbs[p * 3 + 1], c2 = rgbs[p * 3 + 2];
        float w, T_post, r_i, g_i, b_i, ws_i;
        comp_chunk(active, sigma, dl.x, dl.y, c0, c1, c2, lane, k, w, T_post, r_i, g_i, b_i, ws_i);
    }
    if (lane == 0) {
        weights_sum[index] = k.ws;
        if (depth) depth[index] = k.d;
        image[(size_t)index * 3] = k.r; image[(size_t)index * 3 + 1] = k.g; image[(size_t)index * 3 + 2] = k.b;
        if (out_image) {
            // image + (1 - weights_sum) * bg_color, in torch's operation order (nerf/renderer.py:352)
            const float rest = 1.0f - k.ws;
            // (separately rounded multiply and add: bit-identical to the elementwise route, no fma contraction)
            out_image[(size_t)index * 3] = __fadd_rn(k.r, __fmul_rn(rest, bg.at(index, 0)));
            out_image[(size_t)index * 3 + 1] = __fadd_rn(k.g, __fmul_rn(rest, bg.at(index, 1)));
            out_image[(size_t)index * 3 + 2] = __fadd_rn(k.b, __fmul_rn(rest, bg.at(index, 2)));
        }
    }
}

// the fused form of the backward: gradient of an MSE loss on the blended image, computed in place of reading it
struct MseTail {
    const float* out_image;
    const float* target;
    float scale;
    Background bg;
    const int32_t* counter;
    float* loss;            // optional: += loss_scale * sum((out_image - target)^2) over the rays (one atomic per workgroup)
    float loss_scale;
    const float* scale_mul; // optional (enerf_amp_begin): the gradient -- not the reported loss -- is multiplied by *scale_mul
};

template <bool MSE>
__global__ void __launch_bounds__(MSE ? 1024 : 256) k_composite_train_bwd(
    const float* __restrict__ grad_weights_sum, const float* __restrict__ grad_image,
    const float* __restrict__ sigmas, const float* __restrict__ rgbs, const float* __restrict__ deltas,
    const int32_t* __restrict__ rays, const float* __restrict__ weights_sum, const float* __restrict__ image,
    uint32_t M, uint32_t N, float* grad_sigmas, float* grad_rgbs, MseTail mse, uint32_t ray_blocks) {
    if (MSE && blockIdx.x >= ray_blocks) {
        // rows past the last reserved sample get zero gradients (the caller handed over uninitialised buffers)
        const uint32_t used = min((uint32_t)mse.counter[0], M);
        const uint32_t tid = (blockIdx.x - ray_blocks) * blockDim.x + threadIdx.x;
        const uint32_t nth = (gridDim.x - ray_blocks) * blockDim.x;
        for (size_t i = (size_t)used + tid; i < (size_t)M; i += nth) grad_sigmas[i] = 0.0f;
        for (size_t i = (size_t)used * 3 + tid; i < (size_t)M * 3; i += nth) grad_rgbs[i] = 0.0f;
        return;
    }
    const uint32_t n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (MSE && mse.loss && mse.target) {
        // the loss value itself, for whoever logs it: every ray contributes, marched or not
        __shared__ float s_err[16];
        float e = 0.0f;
        if (n < N && lane_id() < 3) {
            const uint32_t ray = (uint32_t)rays[(size_t)n * 3];
            const float d = mse.out_image[(size_t)ray * 3 + lane_id()] - mse.target[(size_t)ray * 3 + lane_id()];
            e = d * d;
        }
        e += dpp_take<0x102>(0.0f, e);                        // lanes 0..2 -> lane 0 (row_shl 2, 1)
        e += dpp_take<0x101>(0.0f, e);
        if (lane_id() == 0) s_err[threadIdx.x >> 6] = e;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.0f;
            for (uint32_t w = 0; w < (blockDim.x >> 6); w++) t += s_err[w];
            atomicAdd(mse.loss, t * mse.loss_scale);      // same-address atomics cost ~12 ns each: 16 rays share one
        }
    }
    if (n >= N) return;
    const int lane = lane_id();
    const uint32_t index = (uint32_t)rays[(size_t)n * 3], offset = (uint32_t)rays[(size_t)n * 3 + 1];
    const uint32_t num_steps = (uint32_t)rays[(size_t)n * 3 + 2];
    if (num_steps == 0 || offset + num_steps >= M) {
        if (MSE && num_steps != 0 && offset < M) {          // a dropped ray's reservation, clipped to the buffer
            for (size_t i = (size_t)offset + lane; i < (size_t)M; i += 64) grad_sigmas[i] = 0.0f;
            for (size_t i = (size_t)offset * 3 + lane; i < (size_t)M * 3; i += 64) grad_rgbs[i] = 0.0f;
        }
        return;
    }
    float gws, gi0, gi1, gi2;
    if (MSE) {
        // loss = mean((out_image - target)^2): d/d(out_image) = (out_image - target) * scale, scale = 2 / (3 N) * upstream;
        // out_image = image + (1 - weights_sum) * bg  ->  d/d(weights_sum) = -(g . bg)
        // (target == nullptr: out_image already holds d loss / d out_image of some other loss)
        gi0 = mse.out_image[(size_t)index * 3], gi1 = mse.out_image[(size_t)index * 3 + 1];
        gi2 = mse.out_image[(size_t)index * 3 + 2];
        if (mse.target) {
            gi0 -= mse.target[(size_t)index * 3];
            gi1 -= mse.target[(size_t)index * 3 + 1];
            gi2 -= mse.target[(size_t)index * 3 + 2];
        }
        const float gsc = mse.scale_mul ? mse.scale * mse.scale_mul[0] : mse.scale;
        gi0 *= gsc; gi1 *= gsc; gi2 *= gsc;
        gws = -(gi0 * mse.bg.at(index, 0) + gi1 * mse.bg.at(index, 1) + gi2 * mse.bg.at(index, 2));
    } else {
        gws = grad_weights_sum[index];
        gi0 = grad_image[(size_t)index * 3]; gi1 = grad_image[(size_t)index * 3 + 1];
        gi2 = grad_image[(size_t)index * 3 + 2];
    }
    const float r_final = image[(size_t)index * 3], g_final = image[(size_t)index * 3 + 1],
                b_final = image[(size_t)index * 3 + 2];
    const float ws_final = weights_sum[index];
    CompCarry k = {1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (uint32_t s0 = 0; s0 < num_steps; s0 += 64) {
        const uint32_t s = s0 + lane;
        const bool active = s < num_steps;
        const size_t p = (size_t)offset + (active ? s : 0);
        const float sigma = sigmas[p];
        const float2 dl = reinterpret_cast<const float2*>(deltas)[p];
        const float c0 = rgbs[p * 3], c1 = rgbs[p * 3 + 1], c2 = rgbs[p * 3 + 2];
        float w, T, r_i, g_i, b_i, ws_i;
        comp_chunk(active, sigma, dl.x, dl.y, c0, c1, c2, lane, k, w, T, r_i, g_i, b_i, ws_i);
        if (active) {
            grad_rgbs[p * 3] = gi0 * w;
            grad_rgbs[p * 3 + 1] = gi1 * w;
            grad_rgbs[p * 3 + 2] = gi2 * w;
            grad_sigmas[p] = dl.x * (gi0 * (T * c0 - (r_final - r_i)) + gi1 * (T * c1 - (g_final - g_i)) +
                                     gi2 * (T * c2 - (b_final - b_i)) + gws * (T - (ws_final - ws_i)));
        }
    }
}

// Forward, blend, MSE gradient and backward of composite_rays_train for one ray in one wavefront: the closed-form
// training step needs nothing between the two compositing kernels, so the second pass over the ray's samples follows
// the first while they are still in L2, image / weights_sum stay in registers, and one launch (and the gap before it)
// disappears.  Same arithmetic, in the same order, as k_composite_train_fwd + k_composite_train_bwd<true>.
__global__ void __launch_bounds__(1024) k_composite_train_fwd_bwd_mse(
    const float* __restrict__ sigmas, const float* __restrict__ rgbs, const float* __restrict__ deltas,
    const int32_t* __restrict__ rays, uint32_t M, uint32_t N, float* weights_sum, float* image, float* out_image,
    float* grad_sigmas, float* grad_rgbs, MseTail mse, uint32_t ray_blocks) {
    if (blockIdx.x >= ray_blocks) {
        const uint32_t used = min((uint32_t)mse.counter[0], M);
        const uint32_t tid = (blockIdx.x - ray_blocks) * blockDim.x + threadIdx.x;
        const uint32_t nth = (gridDim.x - ray_blocks) * blockDim.x;
        for (size_t i = (size_t)used + tid; i < (size_t)M; i += nth) grad_sigmas[i] = 0.0f;
        for (size_t i = (size_t)used * 3 + tid; i < (size_t)M * 3; i += nth) grad_rgbs[i] = 0.0f;
        return;
    }
    const uint32_t n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = lane_id();
    const bool live = n < N;
    uint32_t index = 0, offset = 0, num_steps = 0;
    if (live) {
        index = (uint32_t)rays[(size_t)n * 3];
        offset = (uint32_t)rays[(size_t)n * 3 + 1];
        num_steps = (uint32_t)rays[(size_t)n * 3 + 2];
    }
    const bool marched = live && num_steps != 0 && offset + num_steps < M;
    // ---- forward
    CompCarry k = {1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (marched) {
        for (uint32_t s0 = 0; s0 < num_steps; s0 += 64) {
            const uint32_t s = s0 + lane;
            const bool active = s < num_steps;
            const size_t p = (size_t)offset + (active ? s : 0);
            const float sigma = sigmas[p];
            const float2 dl = reinterpret_cast<const float2*>(deltas)[p];
            const float c0 = rgbs[p * 3], c1 = rgbs[p * 3 + 1], c2 = rgbs[p * 3 + 2];
            float w, T_post, r_i, g_i, b_i, ws_i;
            comp_chunk(active, sigma, dl.x, dl.y, c0, c1, c2, lane, k, w, T_post, r_i, g_i, b_i, ws_i);
        }
    }
    float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
    if (live) {
        if (marched) {
            const float rest = 1.0f - k.ws;
            o0 = __fadd_rn(k.r, __fmul_rn(rest, mse.bg.at(index, 0)));
            o1 = __fadd_rn(k.g, __fmul_rn(rest, mse.bg.at(index, 1)));
            o2 = __fadd_rn(k.b, __fmul_rn(rest, mse.bg.at(index, 2)));
        } else {
            o0 = mse.bg.at(index, 0); o1 = mse.bg.at(index, 1); o2 = mse.bg.at(index, 2);
        }
        if (lane == 0) {
            weights_sum[index] = marched ? k.ws : 0.0f;
            image[(size_t)index * 3] = marched ? k.r : 0.0f;
            image[(size_t)index * 3 + 1] = marched ? k.g : 0.0f;
            image[(size_t)index * 3 + 2] = marched ? k.b : 0.0f;
            out_image[(size_t)index * 3] = o0; out_image[(size_t)index * 3 + 1] = o1; out_image[(size_t)index * 3 + 2] = o2;
        }
    }
    // ---- loss value (every ray contributes, marched or not)
    if (mse.loss) {
        __shared__ float s_err[16];
        float e = 0.0f;
        if (live && lane < 3) {
            const float d = (lane == 0 ? o0 : lane == 1 ? o1 : o2) - mse.target[(size_t)index * 3 + lane];
            e = d * d;
        }
        e += dpp_take<0x102>(0.0f, e);                        // lanes 0..2 -> lane 0 (row_shl 2, 1)
        e += dpp_take<0x101>(0.0f, e);
        if (lane == 0) s_err[threadIdx.x >> 6] = e;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.0f;
            for (uint32_t w = 0; w < (blockDim.x >> 6); w++) t += s_err[w];
            atomicAdd(mse.loss, t * mse.loss_scale);
        }
    }
    // ---- backward
    if (!marched) {
        if (live && num_steps != 0 && offset < M) {          // a dropped ray's reservation, clipped to the buffer
            for (size_t i = (size_t)offset + lane; i < (size_t)M; i += 64) grad_sigmas[i] = 0.0f;
            for (size_t i = (size_t)offset * 3 + lane; i < (size_t)M * 3; i += 64) grad_rgbs[i] = 0.0f;
        }
        return;
    }
    float gi0 = o0 - mse.target[(size_t)index * 3], gi1 = o1 - mse.target[(size_t)index * 3 + 1],
          gi2 = o2 - mse.target[(size_t)index * 3 + 2];
    const float gsc = mse.scale_mul ? mse.scale * mse.scale_mul[0] : mse.scale;
    gi0 *= gsc; gi1 *= gsc; gi2 *= gsc;
    const float gws = -(gi0 * mse.bg.at(index, 0) + gi1 * mse.bg.at(index, 1) + gi2 * mse.bg.at(index, 2));
    const float r_final = k.r, g_final = k.g, b_final = k.b, ws_final = k.ws;
    CompCarry kb = {1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (uint32_t s0 = 0; s0 < num_steps; s0 += 64) {
        const uint32_t s = s0 + lane;
        const bool active = s < num_steps;
        const size_t p = (size_t)offset + (active ? s : 0);
        const float sigma = sigmas[p];
        const float2 dl = reinterpret_cast<const float2*>(deltas)[p];
        const float c0 = rgbs[p * 3], c1 = rgbs[p * 3 + 1], c2 = rgbs[p * 3 + 2];
        float w, T, r_i, g_i, b_i, ws_i;
        comp_chunk(active, sigma, dl.x, dl.y, c0, c1, c2, lane, kb, w, T, r_i, g_i, b_i, ws_i);
        if (active) {
            grad_rgbs[p * 3] = gi0 * w;
            grad_rgbs[p * 3 + 1] = gi1 * w;
            grad_rgbs[p * 3 + 2] = gi2 * w;
            grad_sigmas[p] = dl.x * (gi0 * (T * c0 - (r_final - r_i)) + gi1 * (T * c1 - (g_final - g_i)) +
                                     gi2 * (T * c2 - (b_final - b_i)) + gws * (T - (ws_final - ws_i)));
        }
    }
}

// ------------------------------------------------------------------ inference: march / composite / compact
// One thread per ray, dt_gamma == 0, with the fast cell evaluator and the empty-voxel skip in closed form.  The
// reference's `do { t += dt; } while (t < tt);` walks the lattice t, fl(t + dt), ... one addition at a time; inside a
// binade those points are t + k * delta exactly (see the lattice marcher below for the conditions), so the last point
// below min(voxel exit, top of the binade) is reached with one fma and only the step that crosses it is a real
// addition.  Same samples, bit for bit; an empty stretch costs O(binades crossed) instead of O(steps skipped).
__device__ __forceinline__ float skip_empty(float t, float tt, float dt) {
    t += dt;                                                  // do { ... } executes at least once
    while (t < tt) {
        const float delta = (t + dt) - t;
        const float delta2 = ((t + delta) + dt) - (t + delta);
        if (t >= 2.0f * dt && delta2 == delta) {
            int e;
            (void)frexpf(t, &e);
            const float lim = fminf(tt, ldexpf(1.0f, e));     // t < lim: t is below the voxel exit and inside its binade
            float m = floorf((lim - t) * __builtin_amdgcn_rcpf(delta));
            while (m > 0.0f && !(fmaf(m, delta, t) < lim)) m -= 1.0f;
            while (fmaf(m + 1.0f, delta, t) < lim) m += 1.0f;
            t = fmaf(m, delta, t);                            // last lattice point below lim (exact)
        }
        t += dt;
    }
    return t;
}
template <bool WRITE>
__device__ __forceinline__ uint32_t march_one_ray_fast(const RayCtx& c, const RayFixed& rf, const MarchTabs& tb, float t0,
                                                       float far, uint32_t limit, float* xyzs, float* dirs,
                                                       float* deltas) {
    const float dt = c.dt_min;
    float t = t0, last_t = t0;
    uint32_t step = 0;
    while (t < far && step < limit) {
        float x, y, z, tt;
        if (eval_cell_fixed(c, rf, tb, t, x, y, z, tt)) {
            t += dt;
            if (WRITE) {
                xyzs[0] = x; xyzs[1] = y; xyzs[2] = z;
                dirs[0] = c.dx; dirs[1] = c.dy; dirs[2] = c.dz;
                deltas[0] = dt;
                deltas[1] = t - last_t;
                last_t = t;
                xyzs += 3; dirs += 3; deltas += 2;
            }
            step++;
        } else {
            t = skip_empty(t, tt, dt);
        }
    }
    return step;
}

// ------------------------------------------------------------------ march_rays_train, one thread per ray (many rays)
// The wave-per-ray lattice marcher evaluates every lattice point of a ray, 64 at a time; the reference's loop only
// VISITS one point per empty voxel crossed (a jump of 4 - 9 lattice points at the usual cell sizes), so the lattice does
// ~6 evaluations per visited point.  That buys latency when rays are few (4096 rays = 4096 wavefronts of parallel work
// instead of 64), and costs throughput when they are many: at 65 536+ rays every SIMD has several wavefronts of
// one-thread-per-ray work to hide its loads behind, and the sequential walk (fast cell evaluator, closed-form skip of an
// empty stretch) issues a sixth of the instructions.  So large batches -- a whole 640 x 480 frame, a 65 536-ray training
// step -- count with one thread per ray, logging the RUNS of consecutive samples it emits (first lattice point, length);
// the write pass stays one wavefront per ray and replays the runs with coalesced stores (the points of a run are
// t_first + k * delta exactly inside a binade, as in the lattice marcher).  Same samples, offsets and counters, bit for
// bit: the thread walks the reference's own loop.
struct RunEntry {
    float t_first;
    uint32_t len;
};
constexpr uint32_t kRunCap = 64;                 // runs per ray; a ray with more is re-marched by the write pass

__global__ void __launch_bounds__(256) k_march_count_t(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                       const uint8_t* __restrict__ grid, float bound, uint32_t max_steps,
                                                       uint32_t N, uint32_t C, uint32_t H,
                                                       const float* __restrict__ nears, const float* __restrict__ fars,
                                                       int32_t* rays, uint32_t perturb, RunEntry* __restrict__ log,
                                                       uint32_t* __restrict__ nlog, const int* __restrict__ occ_keys) {
    __shared__ float s_face[kTabH + 1];
    __shared__ uint32_t s_expand[kTabH];
    build_march_tabs(s_face, s_expand, H);
    const MarchTabs tabs = {s_face, s_expand};
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    RayCtx c;
    ray_ctx_init(c, rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, grid, bound, 0.0f, max_steps, C, H);
    float t = nears[n];
    if (perturb) t = fmaf(c.dt_min, pcg_first_float((uint64_t)n, 1u), t);     // contracted by the reference's compiler (:351)
    float far = fars[n];
    uint32_t cnt = 0, nruns = 0;
    if (!occ_keys || clip_to_occupied(c, occ_keys, far)) {
        RayFixed rf;
        ray_fixed_init(rf, c);
        RunEntry* lg = log + (size_t)n * kRunCap;
        const float dt = c.dt_min;
        float run_first = 0.0f;
        uint32_t run_len = 0;
        while (t < far && cnt < max_steps) {
            float x, y, z, tt;
            if (eval_cell_fixed(c, rf, tabs, t, x, y, z, tt)) {
                if (run_len == 0) run_first = t;
                run_len++;
                cnt++;
                t += dt;
            } else {
                if (run_len) {
                    if (nruns < kRunCap) lg[nruns] = RunEntry{run_first, run_len};
                    nruns++;
                    run_len = 0;
                }
                t = skip_empty(t, tt, dt);
            }
        }
        if (run_len) {
            if (nruns < kRunCap) lg[nruns] = RunEntry{run_first, run_len};
            nruns++;
        }
    }
    nlog[n] = nruns <= kRunCap ? nruns : kLogOverflow;
    rays[(size_t)n * 3 + 2] = (int32_t)cnt;
}

// replay of a ray's runs by one wavefront: 64 consecutive lattice points per round while the progression holds
__device__ __forceinline__ void run_replay(const RayCtx& c, float t0, const RunEntry* __restrict__ log, uint32_t nruns,
                                           float* xyzs, float* dirs, float* deltas) {
    const int lane = lane_id();
    const float dt = c.dt_min;
    float last_t = t0;
    uint32_t count = 0;
    for (uint32_t e = 0; e < nruns; e++) {
        float base = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(log[e].t_first)));
        uint32_t rem = __builtin_amdgcn_readfirstlane(log[e].len);
        while (rem) {
            const float delta = (base + dt) - base;
            const float delta2 = ((base + delta) + dt) - (base + delta);
            int ex;
            (void)frexpf(base, &ex);
            const float bin_top = ldexpf(1.0f, ex);
            const bool progression = base >= 2.0f * dt && delta2 == delta;
            const float ti = progression ? fmaf((float)lane, delta, base) : base;
            const bool ok = lane == 0 || (progression && ti < bin_top);
            const unsigned long long okm = __ballot(ok);
            const uint32_t nvalid = okm == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~okm);
            const uint32_t take = nvalid < rem ? nvalid : rem;
            const float t_next = ti + dt;
            const float prev_next = __shfl(t_next, lane ? lane - 1 : 0, 64);
            if ((uint32_t)lane < take) {
                const size_t k = (size_t)count + (uint32_t)lane;
                xyzs[k * 3] = clampf_(fmaf(ti, c.dx, c.ox), -c.bound, c.bound);
                xyzs[k * 3 + 1] = clampf_(fmaf(ti, c.dy, c.oy), -c.bound, c.bound);
                xyzs[k * 3 + 2] = clampf_(fmaf(ti, c.dz, c.oz), -c.bound, c.bound);
                dirs[k * 3] = c.dx; dirs[k * 3 + 1] = c.dy; dirs[k * 3 + 2] = c.dz;
                deltas[k * 2] = dt;
                deltas[k * 2 + 1] = t_next - (lane ? prev_next : last_t);
            }
            last_t = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t_next), (int)take - 1));
            base = last_t;                               // the true next lattice value after the last point taken
            count += take;
            rem -= take;
        }
    }
}

__global__ void __launch_bounds__(256) k_march_write_r(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                       const uint8_t* __restrict__ grid, float bound, uint32_t max_steps,
                                                       uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                                       const float* __restrict__ nears, const float* __restrict__ fars,
                                                       float* xyzs, float* dirs, float* deltas,
                                                       const int32_t* __restrict__ rays, uint32_t perturb,
                                                       const RunEntry* __restrict__ log, const uint32_t* __restrict__ nlog,
                                                       const int32_t* __restrict__ counter, uint32_t ray_blocks) {
    if (blockIdx.x >= ray_blocks) {
        const uint32_t used = min((uint32_t)counter[0], M);
        zero_rows(xyzs, dirs, deltas, used, M, (blockIdx.x - ray_blocks) * blockDim.x + threadIdx.x,
                  (gridDim.x - ray_blocks) * blockDim.x);
        return;
    }
    const uint32_t n = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (n >= N) return;
    const uint32_t point_index = (uint32_t)rays[(size_t)n * 3 + 1];
    const uint32_t num_steps = (uint32_t)rays[(size_t)n * 3 + 2];
    if (num_steps == 0) return;
    if (point_index + num_steps >= M) {
        if (counter && point_index < M) zero_rows(xyzs, dirs, deltas, point_index, M, lane_id(), 64);
        return;
    }
    RayCtx c;
    ray_ctx_init(c, rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, grid, bound, 0.0f, max_steps, C, H);
    float t0 = nears[n];
    if (perturb) t0 = fmaf(c.dt_min, pcg_first_float((uint64_t)n, 1u), t0);   // contracted by the reference's compiler (:351)
    const uint32_t entries = __builtin_amdgcn_readfirstlane(nlog[n]);
    if (entries != kLogOverflow)
        run_replay(c, t0, log + (size_t)n * kRunCap, entries, xyzs + (size_t)point_index * 3,
                   dirs + (size_t)point_index * 3, deltas + (size_t)point_index * 2);
    else
        (void)lattice_march<true>(c, t0, fars[n], num_steps, xyzs + (size_t)point_index * 3,
                                  dirs + (size_t)point_index * 3, deltas + (size_t)point_index * 2);
}

__global__ void __launch_bounds__(256) k_march_rays(uint32_t n_alive, uint32_t n_step,
                                                    const int32_t* __restrict__ rays_alive,
                                                    const float* __restrict__ rays_t,
                                                    const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                    float bound, float dt_gamma, uint32_t max_steps, uint32_t C,
                                                    uint32_t H, const uint8_t* __restrict__ grid,
                                                    const float* __restrict__ fars, float* xyzs, float* dirs,
                                                    float* deltas, uint32_t perturb, uint32_t zero_rows_to,
                                                    const int* __restrict__ occ_keys) {
    __shared__ float s_face[kTabH + 1];
    __shared__ uint32_t s_expand[kTabH];
    const bool fast = dt_gamma == 0.0f && march_fast_ok(H);
    if (fast) build_march_tabs(s_face, s_expand, H);
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (zero_rows_to && blockIdx.x == gridDim.x - 1)          // alignment rows past the last ray's slots
        zero_rows(xyzs, dirs, deltas, n_alive * n_step, zero_rows_to, threadIdx.x, blockDim.x);
    if (n >= n_alive) return;
    const int index = rays_alive[n];
    float t = rays_t[n];
    RayCtx c;
    ray_ctx_init(c, rays_o + (size_t)index * 3, rays_d + (size_t)index * 3, grid, bound, dt_gamma, max_steps, C, H);
    if (perturb) t = fmaf(c.dt_min, pcg_first_float((uint64_t)n, (uint64_t)perturb), t);   // :744, contracted
    const size_t base = (size_t)n * n_step;
    uint32_t got;
    // the occupied cells' box (as in the training count pass): no sample lies outside it, so the walk may stop at its far
    // side -- or not start at all; the slots stay zero either way, which is what ends the ray in composite_rays
    float far = fars[index];
    if (occ_keys && !clip_to_occupied(c, occ_keys, far)) far = t;
    if (fast) {
        RayFixed rf;
        ray_fixed_init(rf, c);
        const MarchTabs tabs = {s_face, s_expand};
        got = march_one_ray_fast<true>(c, rf, tabs, t, far, n_step, xyzs + base * 3, dirs + base * 3,
                                       deltas + base * 2);
    } else {
        got = march_one_ray<true>(c, t, far, n_step, xyzs + base * 3, dirs + base * 3, deltas + base * 2);
    }
    if (zero_rows_to && got < n_step)                          // the slots this ray did not fill (delta == 0: end)
        zero_rows(xyzs, dirs, deltas, (uint32_t)base + got, (uint32_t)base + n_step, 0, 1);
}

// wave-per-ray variant (dt_gamma == 0): same lattice marcher as training; pays off once rays are few or n_step is
// large, i.e. when the one-thread-per-ray loop is latency-bound
__global__ void __launch_bounds__(256) k_march_rays_w(uint32_t n_alive, uint32_t n_step,
                                                      const int32_t* __restrict__ rays_alive,
                                                      const float* __restrict__ rays_t,
                                                      const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                      float bound, uint32_t max_steps, uint32_t C, uint32_t H,
                                                      const uint8_t* __restrict__ grid, const float* __restrict__ fars,
                                                      float* xyzs, float* dirs, float* deltas, uint32_t perturb,
                                                      uint32_t zero_rows_to, const int* __restrict__ occ_keys) {
    __shared__ float s_face[kTabH + 1];
    __shared__ uint32_t s_expand[kTabH];
    const bool fast = march_fast_ok(H);
    if (fast) build_march_tabs(s_face, s_expand, H);
    const MarchTabs tabs = {s_face, s_expand};
    const uint32_t n = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (zero_rows_to && blockIdx.x == gridDim.x - 1)
        zero_rows(xyzs, dirs, deltas, n_alive * n_step, zero_rows_to, threadIdx.x, blockDim.x);
    if (n >= n_alive) return;
    const int index = rays_alive[n];
    float t = rays_t[n];
    RayCtx c;
    ray_ctx_init(c, rays_o + (size_t)index * 3, rays_d + (size_t)index * 3, grid, bound, 0.0f, max_steps, C, H);
    if (perturb) t = fmaf(c.dt_min, pcg_first_float((uint64_t)n, (uint64_t)perturb), t);   // :744, contracted
    const size_t base = (size_t)n * n_step;
    uint32_t got;
    float far = fars[index];
    if (occ_keys && !clip_to_occupied(c, occ_keys, far)) far = t;           // (see k_march_rays)
    if (fast)
        got = lattice_march_fast<true, false>(c, tabs, t, far, n_step, xyzs + base * 3, dirs + base * 3,
                                              deltas + base * 2, nullptr, nullptr);
    else
        got = lattice_march<true>(c, t, far, n_step, xyzs + base * 3, dirs + base * 3, deltas + base * 2);
    got = __builtin_amdgcn_readfirstlane(got);
    if (zero_rows_to && got < n_step)
        zero_rows(xyzs, dirs, deltas, (uint32_t)base + got, (uint32_t)base + n_step, lane_id(), 64);
}

__global__ void __launch_bounds__(256) k_composite_rays(uint32_t n_alive, uint32_t n_step,
                                                        const int32_t* __restrict__ rays_alive, float* rays_t,
                                                        const float* __restrict__ sigmas,
                                                        const float* __restrict__ rgbs,
                                                        const float* __restrict__ deltas, float* weights_sum,
                                                        float* depth, float* image) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    const int index = rays_alive[n];
    float t = rays_t[n];
    const float* s = sigmas + (size_t)n * n_step;
    const float* c = rgbs + (size_t)n * n_step * 3;
    const float* dl = deltas + (size_t)n * n_step * 2;
    float weight_sum = weights_sum[index], d = depth[index];
    float r = image[(size_t)index * 3], g = image[(size_t)index * 3 + 1], b = image[(size_t)index * 3 + 2];
    uint32_t step = 0;
    while (step < n_step) {
        if (dl[0] == 0) break;
        const float alpha = 1.0f - __expf(-s[0] * dl[0]);
        const float T = 1 - weight_sum;
        const float weight = alpha * T;
        weight_sum += weight;
        t += dl[1];
        d = fmaf(weight, t, d);
        r = fmaf(weight, c[0], r);
        g = fmaf(weight, c[1], g);
        b = fmaf(weight, c[2], b);
        if ((double)T < 1e-5) break;
        s++; c += 3; dl += 2; step++;
    }
    rays_t[n] = (step < n_step) ? -1.0f : t;
    weights_sum[index] = weight_sum;
    depth[index] = d;
    image[(size_t)index * 3] = r; image[(size_t)index * 3 + 1] = g; image[(size_t)index * 3 + 2] = b;
}

// Whole-frame inference compositing (enerf_composite_rays_frame): every ray's samples lie contiguously (the training
// marcher's layout, rays = (id, offset, count)), so the round structure of the reference's inference loop --
// march n_step samples, evaluate, accumulate, compact, repeat -- collapses into one pass.  Per ray the arithmetic is
// that of k_composite_rays applied to the same samples in the same order, sequentially (T = 1 - weight_sum is a running
// fp32 sum: a parallel scan would round differently), including its termination rule: the sample whose pre-sample
// transmittance is already < 1e-5 is still accumulated, then the ray stops.  The background blend and the depth
// normalisation of run_cuda (nerf/renderer.py:398-401) are the epilogue.
__global__ void __launch_bounds__(256) k_composite_rays_frame(const float* __restrict__ sigmas,
                                                              const float* __restrict__ rgbs,
                                                              const float* __restrict__ deltas,
                                                              const int32_t* __restrict__ rays, uint32_t N, uint32_t M,
                                                              const float* __restrict__ nears,
                                                              const float* __restrict__ fars, Background bg,
                                                              float* weights_sum, float* depth, float* image,
                                                              uint32_t* __restrict__ used) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[(size_t)n * 3];
    const uint32_t offset = (uint32_t)rays[(size_t)n * 3 + 1];
    uint32_t count = (uint32_t)rays[(size_t)n * 3 + 2];
    if (offset + count >= M) count = 0;                         // the marcher's drop rule (cannot happen with M exact)
    const float near = nears[index], far = fars[index];
    float t = near;
    float weight_sum = 0.0f, d = 0.0f, r = 0.0f, g = 0.0f, b = 0.0f;
    const float* s = sigmas + offset;
    const float* c = rgbs + (size_t)offset * 3;
    const float* dl = deltas + (size_t)offset * 2;
    // samples are fetched eight at a time (one memory latency per eight samples instead of one per sample: the ray's
    // recurrence itself is a dozen flops) and consumed in order
    constexpr int PF = 8;
    uint32_t step = 0;
    bool done = false;
    while (step < count && !done) {
        float ps[PF], pd0[PF], pd1[PF], pc0[PF], pc1[PF], pc2[PF];
        const uint32_t nb = count - step < (uint32_t)PF ? count - step : (uint32_t)PF;
        if (nb == (uint32_t)PF) {
            // a whole group: 12 sixteen-byte loads (dword-aligned is all gfx950 asks of them) instead of 48 four-byte ones --
            // every lane walks its own segment, so each load instruction is 64 separate line look-ups either way
            const float4 s0 = *reinterpret_cast<const float4*>(s), s1 = *reinterpret_cast<const float4*>(s + 4);
            ps[0] = s0.x; ps[1] = s0.y; ps[2] = s0.z; ps[3] = s0.w; ps[4] = s1.x; ps[5] = s1.y; ps[6] = s1.z; ps[7] = s1.w;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 v = *reinterpret_cast<const float4*>(dl + 4 * q);
                pd0[2 * q] = v.x; pd1[2 * q] = v.y; pd0[2 * q + 1] = v.z; pd1[2 * q + 1] = v.w;
            }
            float cc[24];
#pragma unroll
            for (int q = 0; q < 6; q++) {
                const float4 v = *reinterpret_cast<const float4*>(c + 4 * q);
                cc[4 * q] = v.x; cc[4 * q + 1] = v.y; cc[4 * q + 2] = v.z; cc[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int k = 0; k < PF; k++) { pc0[k] = cc[3 * k]; pc1[k] = cc[3 * k + 1]; pc2[k] = cc[3 * k + 2]; }
        } else {
#pragma unroll
            for (int k = 0; k < PF; k++) {
                const uint32_t kk = (uint32_t)k < nb ? (uint32_t)k : nb - 1;
                ps[k] = s[kk];
                pd0[k] = dl[kk * 2]; pd1[k] = dl[kk * 2 + 1];
                pc0[k] = c[kk * 3]; pc1[k] = c[kk * 3 + 1]; pc2[k] = c[kk * 3 + 2];
            }
        }
#pragma unroll
        for (int k = 0; k < PF; k++) {
            if ((uint32_t)k < nb && !done) {
                const float alpha = 1.0f - __expf(-ps[k] * pd0[k]);
                const float T = 1 - weight_sum;
                const float weight = alpha * T;
                weight_sum += weight;
                t += pd1[k];
                d = fmaf(weight, t, d);
                r = fmaf(weight, pc0[k], r);
                g = fmaf(weight, pc1[k], g);
                b = fmaf(weight, pc2[k], b);
                step++;
                if ((double)T < 1e-5) done = true;
            }
        }
        s += nb; c += (size_t)nb * 3; dl += (size_t)nb * 2;
    }
    if (used) atomicAdd(used, step);
    weights_sum[index] = weight_sum;
    const float rest = 1.0f - weight_sum;
    image[(size_t)index * 3] = r + rest * bg.at(index, 0);
    image[(size_t)index * 3 + 1] = g + rest * bg.at(index, 1);
    image[(size_t)index * 3 + 2] = b + rest * bg.at(index, 2);
    depth[index] = fmaxf(d - near, 0.0f) / (far - near);
}

// Stable stream compaction in three small launches: per-block survivor counts (ballot + popcount),
// one-workgroup scan of the block counts, order-preserving scatter.
constexpr int kCompactBlock = 1024;

__global__ void __launch_bounds__(kCompactBlock) k_compact_count(uint32_t n_alive, const float* __restrict__ rays_t_old,
                                                                 uint32_t* block_counts) {
    __shared__ uint32_t wave_cnt[kCompactBlock / 64];
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    const bool keep = n < n_alive && rays_t_old[n] >= 0;
    const unsigned long long m = __ballot(keep);
    if (lane_id() == 0) wave_cnt[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < kCompactBlock / 64; w++) tot += wave_cnt[w];
        block_counts[blockIdx.x] = tot;
    }
}

__global__ void __launch_bounds__(1024) k_compact_scan(uint32_t nblocks, uint32_t* block_counts, int32_t* alive_counter) {
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry_s;
    const int lane = lane_id();
    const int wid = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblocks; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nblocks ? block_counts[i] : 0u;
        const uint32_t incl = wave_incl_scan_add_u32(v, lane);
        if (lane == 63) wave_tot[wid] = incl;
        __syncthreads();
        uint32_t wave_off = 0;
        for (int w = 0; w < wid; w++) wave_off += wave_tot[w];
        const uint32_t carry = carry_s;
        if (i < nblocks) block_counts[i] = carry + wave_off + incl - v;  // exclusive block offset
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + wave_off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) alive_counter[0] += (int32_t)carry_s;
}

__global__ void __launch_bounds__(kCompactBlock) k_compact_scatter(uint32_t n_alive, int32_t* rays_alive,
                                                                   const int32_t* __restrict__ rays_alive_old,
                                                                   float* rays_t, const float* __restrict__ rays_t_old,
                                                                   const uint32_t* __restrict__ block_offsets,
                                                                   const uint32_t* __restrict__ base_count) {
    __shared__ uint32_t wave_cnt[kCompactBlock / 64];
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = lane_id();
    const int wid = threadIdx.x >> 6;
    const float t = n < n_alive ? rays_t_old[n] : -1.0f;
    const bool keep = n < n_alive && t >= 0;
    const unsigned long long m = __ballot(keep);
    if (lane == 0) wave_cnt[wid] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t off = block_offsets[blockIdx.x] + base_count[0];
    for (int w = 0; w < wid; w++) off += wave_cnt[w];
    if (keep) {
        const uint32_t pos = off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        rays_alive[pos] = rays_alive_old[n];
        rays_t[pos] = t;
    }
}

}  // namespace

// ====================================================================== C ABI
extern "C" {

int enerf_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N, float min_near,
                             float* nears, float* fars, enerf_stream_t stream) {
    if (N == 0) return 0;
    k_near_far<<<div_up(N, 256), 256, 0, (hipStream_t)stream>>>(rays_o, rays_d, aabb, N, min_near, nears, fars);
    ENERF_LAUNCH_CHECK("near_far_from_aabb");
    return 0;
}

int enerf_polar_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords,
                         enerf_stream_t stream) {
    if (N == 0) return 0;
    k_polar<<<div_up(N, 256), 256, 0, (hipStream_t)stream>>>(rays_o, rays_d, radius, N, coords);
    ENERF_LAUNCH_CHECK("polar_from_ray");
    return 0;
}

int enerf_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, enerf_stream_t stream) {
    if (N == 0) return 0;
    k_morton3D<<<div_up(N, 256), 256, 0, (hipStream_t)stream>>>(coords, N, indices);
    ENERF_LAUNCH_CHECK("morton3D");
    return 0;
}

int enerf_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, enerf_stream_t stream) {
    if (N == 0) return 0;
    k_morton3D_invert<<<div_up(N, 256), 256, 0, (hipStream_t)stream>>>(indices, N, coords);
    ENERF_LAUNCH_CHECK("morton3D_invert");
    return 0;
}

int enerf_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield, enerf_stream_t stream) {
    if (N == 0) return 0;
    if (((uintptr_t)grid & 15) != 0) ENERF_BADARG("packbits: grid must be 16-byte aligned");
    k_packbits<<<div_up(N, 256), 256, 0, (hipStream_t)stream>>>(grid, N, density_thresh, bitfield);
    ENERF_LAUNCH_CHECK("packbits");
    return 0;
}

static uint32_t g_march_bg_blocks = 0;
static int g_march_clip = 1;        // enerf_debug_march_clip: test rays against the occupied cells' bounding box first
static int g_infer_box = 0;         // enerf_march_rays_use_box: the inference march may trust the cached box
// the bitfield the box in WS_AABB was last computed for (enerf_occupied_box_update)
static const uint8_t* g_box_grid = nullptr;
static uint32_t g_box_C = 0, g_box_H = 0;
static float g_box_bound = 0.0f;

int enerf_march_train_samples(uint64_t* total, int reset, enerf_stream_t stream) {
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) ENERF_BADARG("march_train_samples: stream sync failed");
    unsigned long long v = 0ull;
    if (total) {
        if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_train_samples), sizeof(v)) != hipSuccess)
            ENERF_BADARG("march_train_samples: read failed");
        *total = (uint64_t)v;
    }
    if (reset) {
        v = 0ull;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_train_samples), &v, sizeof(v)) != hipSuccess)
            ENERF_BADARG("march_train_samples: reset failed");
    }
    return 0;
}

int enerf_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                           uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears,
                           const float* fars, float* xyzs, float* dirs, float* deltas, int32_t* rays,
                           int32_t* counter, uint32_t perturb, enerf_stream_t stream) {
    return enerf_march_rays_train_ex(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs,
                                     dirs, deltas, rays, counter, perturb, 0, stream);
}

// The wave-per-ray lattice marcher steps by dt_min: valid when the step is fixed (dt_gamma == 0) and the clamp
// `clamp(t * dt_gamma, dt_min, dt_max)` really yields dt_min, i.e. dt_min <= dt_max <=> max_steps * 2^(C-1) >= H (always
// true at the reference's max_steps = 1024, H = 128); anything else takes the one-thread-per-ray loop.
static inline bool march_uses_lattice(float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H) {
    return dt_gamma == 0.0f && (uint64_t)max_steps * (1ull << (C - 1)) >= (uint64_t)H;
}
// ... and of the fixed-step marchers, the one-thread-per-ray walk with its run log (k_march_count_t / k_march_write_r)
// takes over from the wave-per-ray lattice once there are enough rays to keep every SIMD busy that way
// (enerf_debug_march_thread_min_rays; count and write pass of a batch see the same N, hence the same choice)
static uint32_t g_march_thread_min_rays = 65536u;
static inline bool march_uses_threads(uint32_t N, uint32_t H) {
    return N >= g_march_thread_min_rays && H <= kTabH && (H & (H - 1u)) == 0u;
}
static inline size_t march_log_bytes(uint32_t N, uint32_t H) {
    return march_uses_threads(N, H) ? (size_t)N * kRunCap * sizeof(RunEntry) : (size_t)N * kLogCap * sizeof(ChunkEntry);
}

// enerf_march_fuse_near_far: the next march_rays_train count pass computes near / far itself (written to the nears / fars
// arrays it is given, for the write pass and the renderer) -- one launch less at the head of the side stream's chain
static const float* g_nf_aabb = nullptr;
static float g_nf_min_near = 0.0f;
static int32_t* g_count_mirror = nullptr;      // enerf_march_mirror_count (armed for one count pass)
// Both are one-shot requests for "the next march".  Every public march entry point takes them -- consumes AND disarms --
// as its first statement, before any early return (N == 0, bad arguments, a workspace failure), so that a request can never
// outlive the call it was made for and reach an unrelated march with a stale aabb / host pointer.  The whole-step entry
// points disarm again on their way out (train_step.hip `done:`) in case they failed between arming and marching.
struct MarchOneShot {
    const float* nf_aabb;
    float nf_min_near;
    int32_t* mirror;
};
static inline MarchOneShot march_take_oneshot() {
    MarchOneShot o = {g_nf_aabb, g_nf_min_near, g_count_mirror};
    g_nf_aabb = nullptr;
    g_count_mirror = nullptr;
    return o;
}

// count pass (+ scan): rays[n] = (n, offset, count), counter += (sum, N); the fixed-step marcher also fills the chunk log
static int march_train_count(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                             uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, const float* nears,
                             const float* fars, int32_t* rays, int32_t* counter, uint32_t perturb, bool background,
                             bool use_box, bool fresh_counter, const MarchOneShot& once, hipStream_t s,
                             MarchCountJob* carry_job = nullptr, int ws_slot = WS_MARCH) {
    if (int e = workspace_family_enter(0, s)) return e;
    const float* nf_aabb = once.nf_aabb;
    const float g_nf_min_near = once.nf_min_near;
    int32_t* mirror = once.mirror;
    if (nf_aabb && !(march_uses_lattice(dt_gamma, max_steps, C, H) && !march_uses_threads(N, H))) {
        k_near_far<<<div_up(N, 256), 256, 0, s>>>(rays_o, rays_d, nf_aabb, N, g_nf_min_near, (float*)nears, (float*)fars);
        nf_aabb = nullptr;
    }
    if (march_uses_lattice(dt_gamma, max_steps, C, H)) {
        // fixed step: wave-per-ray lattice marcher (bit-identical results, 64 lattice points per ray in flight)
        // the count pass logs every emitting chunk; the write pass replays the log
        const size_t log_bytes = march_log_bytes(N, H);
        char* ws = (char*)workspace(ws_slot, log_bytes + (size_t)N * sizeof(uint32_t) + 512);
        if (!ws) return ENERF_E_NOMEM;
        ChunkEntry* log = (ChunkEntry*)ws;
        uint32_t* nlog = (uint32_t*)(ws + log_bytes);
        // `background`: the batch is prepared ahead on a side stream.  There the marcher's latency is hidden anyway, and
        // what it costs the step running beside it is its register footprint (66 VGPRs x 4 resident waves per SIMD
        // leave the fused-MLP kernels one wave per SIMD instead of two): one marching wave per SIMD, rays in turn.
        // (measured, ms/step at 2048 / 8192 rays: 0.50 / 0.87 against 0.51 / 0.93 with every ray in flight; at 16384
        // rays the turn-taking only just fits the window -- 1.46 to 1.62 from run to run against a steady 1.52 -- and
        // at 65536 it does not: 5.50 against 4.94; larger batches keep the full launch)
        const uint32_t count_blocks = g_march_bg_blocks ? g_march_bg_blocks : (N <= 8192u ? num_cus() : div_up(N, 4));
        // the occupied cells' box: the cached one if the caller says the one it had computed for this bitfield is current
        // (the training loop: enerf_occupied_box_update once per bitfield), otherwise -- a plain march_rays_train call, whose
        // bitfield may have changed behind the same pointer -- computed here for this call (two small launches, ~8 us,
        // against a third to a half of the count pass: worth it from a few hundred rays up)
        const int* occ_keys = nullptr;
        if (g_march_clip && (H * H * H) % 8 == 0) {
            if (use_box && g_box_grid == grid && g_box_C == C && g_box_H == H && g_box_bound == bound) {
                occ_keys = (const int*)workspace(WS_AABB, 6 * sizeof(int));
            } else if (N >= 512u) {
                int* keys = (int*)(ws + log_bytes + (((size_t)N * sizeof(uint32_t) + 255) & ~(size_t)255));
                k_aabb_init<<<1, 64, 0, s>>>(keys);
                k_occupied_aabb<<<min(div_up(C * H * H * H / 8, 256), 2u * num_cus()), 256, 0, s>>>(grid, C, H, bound, keys);
                occ_keys = keys;
            }
        }
        if (march_uses_threads(N, H))
            k_march_count_t<<<div_up(N, 256), 256, 0, s>>>(rays_o, rays_d, grid, bound, max_steps, N, C, H, nears, fars,
                                                           rays, perturb, (RunEntry*)ws, nlog, occ_keys);
        else
        {
            MarchCountJob job = {rays_o, rays_d, grid, bound, max_steps, N, C, H, nears, fars, rays, perturb, log, nlog,
                                 occ_keys, nf_aabb, g_nf_min_near, (float*)nears, (float*)fars,
                                 background ? min(div_up(N, 4), count_blocks) : div_up(N, 4)};
            if (carry_job) {
                // (enerf::march_carry_begin: the count pass rides in another launch; nothing else of this function runs)
                *carry_job = job;
                return 0;
            }
            k_march_count_w<<<job.blocks, 256, 0, s>>>(job);
        }
    } else {
        k_march_count<<<div_up(N, 64), 64, 0, s>>>(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, nears,
                                                   fars, rays, perturb);
    }
    if (N <= 16384u) {
        k_march_scan<<<1, 1024, 0, s>>>(rays, counter, N, fresh_counter ? 1u : 0u, mirror);
    } else {
        const uint32_t ntiles = div_up(N, 1024);
        uint32_t* tiles = (uint32_t*)workspace(WS_SCAN, (size_t)ntiles * sizeof(uint32_t));
        if (!tiles) return ENERF_E_NOMEM;
        k_march_scan_tile_sums<<<ntiles, 1024, 0, s>>>(rays, N, tiles);
        k_march_scan_tiles<<<1, 1024, 0, s>>>(tiles, ntiles, counter, N, fresh_counter ? 1u : 0u);
        k_march_scan_apply<<<ntiles, 1024, 0, s>>>(rays, N, tiles);
        if (mirror && hipMemcpyAsync(mirror, counter, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess)
            ENERF_BADARG("march_rays_train: could not mirror the counter to the host");      // (large batches: a copy)
    }
    return 0;
}

// write pass: needs `rays` / `counter` (and, for the fixed-step marcher, the chunk log) of the matching count pass
static int march_train_write(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                             uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears,
                             const float* fars, float* xyzs, float* dirs, float* deltas, const int32_t* rays,
                             const int32_t* counter, uint32_t perturb, uint32_t zero_unwritten, hipStream_t s) {
    if (int e = workspace_family_enter(0, s)) return e;
    if (march_uses_lattice(dt_gamma, max_steps, C, H)) {
        const size_t log_bytes = march_log_bytes(N, H);
        char* ws = (char*)workspace(WS_MARCH, log_bytes + (size_t)N * sizeof(uint32_t) + 512);
        if (!ws) return ENERF_E_NOMEM;
        const ChunkEntry* log = (const ChunkEntry*)ws;
        const uint32_t* nlog = (const uint32_t*)(ws + log_bytes);
        const uint32_t ray_blocks = div_up(N, 4);
        if (march_uses_threads(N, H))
            k_march_write_r<<<ray_blocks + (zero_unwritten ? 128u : 0u), 256, 0, s>>>(
                rays_o, rays_d, grid, bound, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays, perturb,
                (const RunEntry*)ws, nlog, zero_unwritten ? counter : nullptr, ray_blocks);
        else
            k_march_write_w<<<ray_blocks + (zero_unwritten ? 128u : 0u), 256, 0, s>>>(
                rays_o, rays_d, grid, bound, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays, perturb, log,
                nlog, zero_unwritten ? counter : nullptr, ray_blocks);
    } else {
        if (zero_unwritten) {
            (void)hipMemsetAsync(xyzs, 0, (size_t)M * 12, s);
            (void)hipMemsetAsync(dirs, 0, (size_t)M * 12, s);
            (void)hipMemsetAsync(deltas, 0, (size_t)M * 8, s);
        }
        k_march_write<<<div_up(N, 64), 64, 0, s>>>(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears,
                                                   fars, xyzs, dirs, deltas, rays, perturb);
    }
    return 0;
}

int enerf_march_mirror_count(int32_t* host_counter) {
    g_count_mirror = host_counter;
    return 0;
}

int enerf_march_fuse_near_far(const float* aabb, float min_near) {
    g_nf_aabb = aabb;
    g_nf_min_near = min_near;
    return 0;
}

int enerf_march_rays_train_ex(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound,
                              float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                              const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                              int32_t* rays, int32_t* counter, uint32_t perturb, uint32_t zero_unwritten,
                              enerf_stream_t stream) {
    const MarchOneShot once = march_take_oneshot();
    if (N == 0) {
        if (zero_unwritten && M) {
            (void)hipMemsetAsync(xyzs, 0, (size_t)M * 12, (hipStream_t)stream);
            (void)hipMemsetAsync(dirs, 0, (size_t)M * 12, (hipStream_t)stream);
            (void)hipMemsetAsync(deltas, 0, (size_t)M * 8, (hipStream_t)stream);
        }
        return 0;
    }
    if (C == 0 || H < 2 || max_steps == 0) ENERF_BADARG("march_rays_train: bad C=%u H=%u max_steps=%u", C, H, max_steps);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(ENERF_K_MARCH_TRAIN, s);
    int rc = march_train_count(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, nears, fars, rays, counter,
                               perturb, (zero_unwritten & 2u) != 0, (zero_unwritten & 4u) != 0, (zero_unwritten & 8u) != 0,
                               once, s);
    if (rc) return rc;
    rc = march_train_write(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas,
                           rays, counter, perturb, zero_unwritten & 1u, s);
    if (rc) return rc;
    ENERF_LAUNCH_CHECK("march_rays_train");
    return 0;
}

int enerf_march_rays_train_count(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound,
                                 float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                 const float* nears, const float* fars, int32_t* rays, int32_t* counter,
                                 uint32_t perturb, uint32_t flags, enerf_stream_t stream) {
    const MarchOneShot once = march_take_oneshot();
    if (N == 0) return 0;
    if (C == 0 || H < 2 || max_steps == 0)
        ENERF_BADARG("march_rays_train_count: bad C=%u H=%u max_steps=%u", C, H, max_steps);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(ENERF_K_MARCH_TRAIN, s);
    const int rc = march_train_count(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, nears, fars, rays, counter,
                                     perturb, (flags & 2u) != 0, (flags & 4u) != 0, (flags & 8u) != 0, once, s);
    if (rc) return rc;
    ENERF_LAUNCH_CHECK("march_rays_train_count");
    return 0;
}

int enerf_march_rays_train_write(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound,
                                 float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                 const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                                 const int32_t* rays, const int32_t* counter, uint32_t perturb, uint32_t zero_unwritten,
                                 enerf_stream_t stream) {
    if (N == 0) {
        if (zero_unwritten && M) {
            (void)hipMemsetAsync(xyzs, 0, (size_t)M * 12, (hipStream_t)stream);
            (void)hipMemsetAsync(dirs, 0, (size_t)M * 12, (hipStream_t)stream);
            (void)hipMemsetAsync(deltas, 0, (size_t)M * 8, (hipStream_t)stream);
        }
        return 0;
    }
    if (C == 0 || H < 2 || max_steps == 0)
        ENERF_BADARG("march_rays_train_write: bad C=%u H=%u max_steps=%u", C, H, max_steps);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(ENERF_K_MARCH_TRAIN, s);
    const int rc = march_train_write(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs,
                                     deltas, rays, counter, perturb, zero_unwritten & 1u, s);
    if (rc) return rc;
    ENERF_LAUNCH_CHECK("march_rays_train_write");
    return 0;
}

int enerf_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas,
                                       const int32_t* rays, uint32_t M, uint32_t N, float* weights_sum, float* depth,
                                       float* image, enerf_stream_t stream) {
    if (N == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(ENERF_K_COMPOSITE_FWD, s);
    k_composite_train_fwd<<<div_up(N, 4), 256, 0, s>>>(sigmas, rgbs, deltas, rays, M, N, weights_sum, depth, image,
                                                       Background{nullptr, 0, 0.0f}, nullptr);
    ENERF_LAUNCH_CHECK("composite_rays_train_forward");
    return 0;
}

int enerf_composite_rays_train_forward_blend(const float* sigmas, const float* rgbs, const float* deltas,
                                             const int32_t* rays, uint32_t M, uint32_t N, float* weights_sum,
                                             float* depth, float* image, const float* bg_color, uint32_t bg_stride,
                                             float bg_scalar, float* out_image, enerf_stream_t stream) {
    if (N == 0) return 0;
    if (!out_image) ENERF_BADARG("composite_rays_train_forward_blend: out_image is required");
    if (bg_color && bg_stride != 0 && bg_stride != 3) ENERF_BADARG("composite_rays_train_forward_blend: bg_stride %u", bg_stride);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(ENERF_K_COMPOSITE_FWD, s);
    k_composite_train_fwd<<<div_up(N, 4), 256, 0, s>>>(sigmas, rgbs, deltas, rays, M, N, weights_sum, depth, image,
                                                       Background{bg_color, bg_stride, bg_scalar}, out_image);
    ENERF_LAUNCH_CHECK("composite_rays_train_forward_blend");
    return 0;
}

int enerf_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image, const float* sigmas,
                                        const float* rgbs, const float* deltas, const int32_t* rays,
                                        const float* weights_sum, const float* image, uint32_t M, uint32_t N,
                                        float* grad_sigmas, float* grad_rgbs, enerf_stream_t stream) {
    if (N == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(ENERF_K_COMPOSITE_BWD, s);
    k_composite_train_bwd<false><<<div_up(N, 4), 256, 0, s>>>(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays,
                                                              weights_sum, image, M, N, grad_sigmas, grad_rgbs,
                                                              MseTail{}, div_up(N, 4));
    ENERF_LAUNCH_CHECK("composite_rays_train_backward");
    return 0;
}

int enerf_composite_rays_train_backward_mse(const float* out_image, const float* target, float grad_scale,
                                            const float* bg_color, uint32_t bg_stride, float bg_scalar,
                                            const int32_t* counter, const float* sigmas, const float* rgbs,
                                            const float* deltas, const int32_t* rays, const float* weights_sum,
                                            const float* image, uint32_t M, uint32_t N, float* grad_sigmas,
                                            float* grad_rgbs, float* loss, enerf_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) {
        if (M) {
            (void)hipMemsetAsync(grad_sigmas, 0, (size_t)M * 4, s);
            (void)hipMemsetAsync(grad_rgbs, 0, (size_t)M * 12, s);
        }
        return 0;
    }
    if (!counter) ENERF_BADARG("composite_rays_train_backward_mse: counter is required");
    if (bg_color && bg_stride != 0 && bg_stride != 3) ENERF_BADARG("composite_rays_train_backward_mse: bg_stride %u", bg_stride);
    ProfScope prof(ENERF_K_COMPOSITE_BWD, s);
    const uint32_t ray_blocks = div_up(N, 16);
    k_composite_train_bwd<true><<<ray_blocks + 16, 1024, 0, s>>>(
        nullptr, nullptr, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, grad_sigmas, grad_rgbs,
        MseTail{out_image, target, grad_scale, Background{bg_color, bg_stride, bg_scalar}, counter, loss,
                1.0f / (3.0f * (float)N), amp_state().scale},
        ray_blocks);
    ENERF_LAUNCH_CHECK("composite_rays_train_backward_mse");
    return 0;
}

int enerf_composite_rays_train_fwd_bwd_mse(const float* sigmas, const float* rgbs, const float* deltas,
                                           const int32_t* rays, uint32_t M, uint32_t N, float* weights_sum,
                                           float* image, const float* bg_color, uint32_t bg_stride, float bg_scalar,
                                           float* out_image, const float* target, float grad_scale,
                                           const int32_t* counter, float* grad_sigmas, float* grad_rgbs, float* loss,
                                           enerf_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) {
        if (M) {
            (void)hipMemsetAsync(grad_sigmas, 0, (size_t)M * 4, s);
            (void)hipMemsetAsync(grad_rgbs, 0, (size_t)M * 12, s);
        }
        return 0;
    }
    if (!counter || !target || !out_image || !weights_sum || !image)
        ENERF_BADARG("composite_rays_train_fwd_bwd_mse: counter, target, out_image, weights_sum and image are required");
    if (bg_color && bg_stride != 0 && bg_stride != 3) ENERF_BADARG("composite_rays_train_fwd_bwd_mse: bg_stride %u", bg_stride);
    ProfScope prof(ENERF_K_COMPOSITE_BWD, s);
    const uint32_t ray_blocks = div_up(N, 16);
    k_composite_train_fwd_bwd_mse<<<ray_blocks + 16, 1024, 0, s>>>(
        sigmas, rgbs, deltas, rays, M, N, weights_sum, image, out_image, grad_sigmas, grad_rgbs,
        MseTail{nullptr, target, grad_scale, Background{bg_color, bg_stride, bg_scalar}, counter, loss,
                1.0f / (3.0f * (float)N), amp_state().scale},
        ray_blocks);
    ENERF_LAUNCH_CHECK("composite_rays_train_fwd_bwd_mse");
    return 0;
}

// tuning aid: workgroups of the background training march (0: one per CU)
int enerf_debug_march_bg_blocks(uint32_t n) {
    g_march_bg_blocks = n;
    return 0;
}

// tuning aid: largest ray count for which the inference march uses one wavefront per ray (n_step < 16)
static uint32_t g_march_wave_max_rays = 65536u;
static uint32_t g_march_wave_min_steps = 16u;
int enerf_occupied_box_update(const uint8_t* grid, uint32_t C, uint32_t H, float bound, enerf_stream_t stream) {
    if (!grid || C == 0 || H < 2 || (H * H * H) % 8 != 0) ENERF_BADARG("occupied_box_update: bad C=%u H=%u", C, H);
    hipStream_t s = (hipStream_t)stream;
    if (int eg = single_device_guard("occupied_box_update")) return eg;
    if (int e = workspace_family_enter(0, s)) return e;
    int* keys = (int*)workspace(WS_AABB, 6 * sizeof(int));
    if (!keys) return ENERF_E_NOMEM;
    (void)hipMemsetAsync(keys, 0x7f, 6 * sizeof(int), s);
    k_occupied_aabb<<<min(div_up(C * H * H * H / 8, 256), 2u * num_cus()), 256, 0, s>>>(grid, C, H, bound, keys);
    g_box_grid = grid; g_box_C = C; g_box_H = H; g_box_bound = bound;
    ENERF_LAUNCH_CHECK("occupied_box_update");
    return 0;
}

int enerf_debug_march_thread_min_rays(uint32_t n) {
    const uint32_t prev = g_march_thread_min_rays;
    if (n) g_march_thread_min_rays = n;
    return (int)prev;
}

int enerf_march_rays_use_box(int on) {
    g_infer_box = on ? 1 : 0;
    return 0;
}

int enerf_debug_march_clip(int on) {
    g_march_clip = on ? 1 : 0;
    return 0;
}

int enerf_debug_march_wave_max_rays(uint32_t n) {
    g_march_wave_max_rays = n & 0xffffffu;          // bits 24..31: minimum n_step for the wave marcher (0 = keep)
    if (n >> 24) g_march_wave_min_steps = n >> 24;
    return 0;
}

int enerf_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                     const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                     uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars, float* xyzs,
                     float* dirs, float* deltas, uint32_t perturb, enerf_stream_t stream) {
    return enerf_march_rays_ex(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H,
                               grid, nears, fars, xyzs, dirs, deltas, perturb, 0, stream);
}

int enerf_march_rays_ex(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                        const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                        uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars,
                        float* xyzs, float* dirs, float* deltas, uint32_t perturb, uint32_t zero_rows_to,
                        enerf_stream_t stream) {
    (void)nears;
    if (zero_rows_to && zero_rows_to < n_alive * n_step)
        ENERF_BADARG("march_rays_ex: zero_rows_to %u < n_alive * n_step", zero_rows_to);
    if (n_alive == 0 || n_step == 0) {
        if (zero_rows_to) {
            (void)hipMemsetAsync(xyzs, 0, (size_t)zero_rows_to * 12, (hipStream_t)stream);
            (void)hipMemsetAsync(dirs, 0, (size_t)zero_rows_to * 12, (hipStream_t)stream);
            (void)hipMemsetAsync(deltas, 0, (size_t)zero_rows_to * 8, (hipStream_t)stream);
        }
        return 0;
    }
    if (C == 0 || H < 2 || max_steps == 0) ENERF_BADARG("march_rays: bad C=%u H=%u max_steps=%u", C, H, max_steps);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(ENERF_K_MARCH_INFER, s);
    // the occupied cells' box: the cached one while the caller vouches for it (enerf_march_rays_use_box: a frame's rounds
    // march one bitfield), else computed for this call when the round is large enough to repay two small launches
    const int* occ_keys = nullptr;
    if (g_march_clip && (H * H * H) % 8 == 0) {
        if (g_infer_box && g_box_grid == grid && g_box_C == C && g_box_H == H && g_box_bound == bound) {
            occ_keys = (const int*)workspace(WS_AABB, 6 * sizeof(int));
        } else if (n_alive >= 32768u) {
            if (int e = workspace_family_enter(0, s)) return e;
            int* keys = (int*)workspace(WS_AABB_CALL, 64);
            if (keys) {
                k_aabb_init<<<1, 64, 0, s>>>(keys);
                k_occupied_aabb<<<min(div_up(C * H * H * H / 8, 256), 2u * num_cus()), 256, 0, s>>>(grid, C, H, bound, keys);
                occ_keys = keys;
            }
        }
    }
    // Same samples either way (bit-identical).  One thread per ray wins while there are enough rays to fill the chip with
    // short loops; one wavefront per ray wins when rays are few or each must produce many samples.
    if (dt_gamma == 0.0f && (n_alive <= g_march_wave_max_rays || n_step >= g_march_wave_min_steps))
        k_march_rays_w<<<div_up(n_alive, 4), 256, 0, s>>>(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound,
                                                          max_steps, C, H, grid, fars, xyzs, dirs, deltas, perturb,
                                                          zero_rows_to, occ_keys);
    else
        k_march_rays<<<div_up(n_alive, 256), 256, 0, s>>>(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound,
                                                          dt_gamma, max_steps, C, H, grid, fars, xyzs, dirs, deltas,
                                                          perturb, zero_rows_to, occ_keys);
    ENERF_LAUNCH_CHECK("march_rays");
    return 0;
}

int enerf_composite_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, float* rays_t,
                         const float* sigmas, const float* rgbs, const float* deltas, float* weights_sum, float* depth,
                         float* image, enerf_stream_t stream) {
    if (n_alive == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(ENERF_K_COMPOSITE_INFER, s);
    k_composite_rays<<<div_up(n_alive, 256), 256, 0, s>>>(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas,
                                                          weights_sum, depth, image);
    ENERF_LAUNCH_CHECK("composite_rays");
    return 0;
}

int enerf_composite_rays_frame(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays,
                               uint32_t N, uint32_t M, const float* nears, const float* fars, const float* bg_color,
                               uint32_t bg_stride, float bg_scalar, float* weights_sum, float* depth, float* image,
                               uint32_t* used_samples, enerf_stream_t stream) {
    if (N == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(ENERF_K_COMPOSITE_INFER, s);
    const Background bg = {bg_color, bg_stride, bg_scalar};
    k_composite_rays_frame<<<div_up(N, 256), 256, 0, s>>>(sigmas, rgbs, deltas, rays, N, M, nears, fars, bg, weights_sum,
                                                          depth, image, used_samples);
    ENERF_LAUNCH_CHECK("composite_rays_frame");
    return 0;
}

int enerf_compact_rays(uint32_t n_alive, int32_t* rays_alive, const int32_t* rays_alive_old, float* rays_t,
                       const float* rays_t_old, int32_t* alive_counter, enerf_stream_t stream) {
    if (n_alive == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (int e0 = workspace_family_enter(0, s)) return e0;
    const uint32_t nb = div_up(n_alive, kCompactBlock);
    // slot layout: [0] = alive_counter value at entry (snapshot), [1..nb] = block counts / offsets
    uint32_t* ws = (uint32_t*)workspace(WS_COMPACT, sizeof(uint32_t) * (nb + 1));
    if (!ws) return ENERF_E_NOMEM;
    // snapshot alive_counter[0] so that survivors are appended after any existing entries (reference: atomicAdd)
    int e = check_hip(hipMemcpyAsync(ws, alive_counter, sizeof(uint32_t), hipMemcpyDeviceToDevice, s), "compact_rays");
    if (e) return e;
    k_compact_count<<<nb, kCompactBlock, 0, s>>>(n_alive, rays_t_old, ws + 1);
    k_compact_scan<<<1, 1024, 0, s>>>(nb, ws + 1, alive_counter);
    // survivors are appended after alive_counter's entry value (ws[0]); the renderer zeroes it first
    // (nerf/renderer.py:372), so this is 0 in practice.
    k_compact_scatter<<<nb, kCompactBlock, 0, s>>>(n_alive, rays_alive, rays_alive_old, rays_t, rays_t_old, ws + 1, ws);
    ENERF_LAUNCH_CHECK("compact_rays");
    return 0;
}

}  // extern "C"

// ---- the count pass carried by another launch (common.h: MarchCountJob) ------------------------------------------------------
namespace {
struct CarriedMarch {
    bool pending = false;
    const float *rays_o = nullptr, *rays_d = nullptr, *nears = nullptr, *fars = nullptr;
    const uint8_t* grid = nullptr;
    float bound = 0.0f;
    uint32_t max_steps = 0, N = 0, C = 0, H = 0, M = 0, perturb = 0, zero_unwritten = 0;
    float *xyzs = nullptr, *dirs = nullptr, *deltas = nullptr;
    int32_t *rays = nullptr, *counter = nullptr;
    const ChunkEntry* log = nullptr;
    const uint32_t* nlog = nullptr;
};
CarriedMarch g_carried[2];
uint32_t g_carried_n = 0;
}  // namespace
// workgroups of the carrying launch that count (enerf_debug_march_carry_blocks; 0 = two per compute unit)
static uint32_t g_march_carry_blocks = getenv("ENERF_MARCH_CARRY_BLOCKS") ? (uint32_t)atoi(getenv("ENERF_MARCH_CARRY_BLOCKS")) : 0u;
extern "C" int enerf_debug_march_carry_blocks(uint32_t blocks) {
    g_march_carry_blocks = blocks;
    return 0;
}

int enerf::march_carry_begin(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                             uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears,
                             const float* fars, float* xyzs, float* dirs, float* deltas, int32_t* rays, int32_t* counter,
                             uint32_t perturb, uint32_t flags, hipStream_t s, MarchCountJob* job, uint32_t share) {
    // what the carried form serves: the wave-per-ray lattice marcher, one-launch scan sizes, a counter taken as (0, 0)
    // (flags bit 3), no count mirror waiting (the cold window's host watches for k_march_scan's store)
    if (N == 0 || N > 16384u || C == 0 || H < 2 || max_steps == 0 || !(flags & 8u) || g_count_mirror != nullptr ||
        g_carried_n >= 2u || !march_uses_lattice(dt_gamma, max_steps, C, H) || march_uses_threads(N, H))
        return 1;
    const MarchOneShot once = march_take_oneshot();
    MarchCountJob j{};
    // (a second pending march logs into a workspace of its own: the first's log waits for its write pass)
    const int rc = march_train_count(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, nears, fars, rays, counter,
                                     perturb, true, (flags & 4u) != 0, true, once, s, &j, g_carried_n == 0 ? WS_MARCH : WS_MARCH2);
    if (rc) return rc;
    if (!j.log) {
        set_error("march_carry_begin: the count pass was not handed over");
        return ENERF_E_BADARG;
    }
    const uint32_t want = (g_march_carry_blocks ? g_march_carry_blocks : 2u * num_cus()) / (share ? share : 1u);
    j.blocks = min(div_up(N, 4), want ? want : 1u);
    *job = j;
    CarriedMarch& c = g_carried[g_carried_n++];
    c = CarriedMarch();
    c.pending = true;
    c.rays_o = rays_o; c.rays_d = rays_d; c.grid = grid; c.bound = bound;
    c.max_steps = max_steps; c.N = N; c.C = C; c.H = H; c.M = M;
    c.nears = nears; c.fars = fars; c.xyzs = xyzs; c.dirs = dirs; c.deltas = deltas;
    c.rays = rays; c.counter = counter; c.perturb = perturb; c.zero_unwritten = flags & 1u;
    c.log = static_cast<const ChunkEntry*>(j.log);
    c.nlog = j.nlog;
    return 0;
}

// the job after all as a launch of its own on `s` (the carrying launch did not take it)
int enerf::march_carry_count_now(const MarchCountJob* job, hipStream_t s) {
    if (!job || job->blocks == 0) return 0;
    k_march_count_w<<<job->blocks, 256, 0, s>>>(*job);
    ENERF_LAUNCH_CHECK("march_rays_train (count, uncarried)");
    return 0;
}
// a step that failed between begin and end
void enerf::march_carry_abort() { g_carried_n = 0; }

int enerf::march_carry_end(hipStream_t s) {
    if (g_carried_n == 0) {
        set_error("march_carry_end: no carried march is pending");
        return ENERF_E_BADARG;
    }
    const uint32_t n = g_carried_n;
    g_carried_n = 0;
    if (int e = workspace_family_enter(0, s)) return e;
    ProfScope prof(ENERF_K_MARCH_TRAIN, s);
    for (uint32_t k = 0; k < n; k++) {
        const CarriedMarch m = g_carried[k];
        const uint32_t ray_blocks = div_up(m.N, 4);
        k_march_scan_write_w<<<ray_blocks + (m.zero_unwritten ? 128u : 0u), 256, 0, s>>>(
            m.rays_o, m.rays_d, m.grid, m.bound, m.max_steps, m.N, m.C, m.H, m.M, m.nears, m.fars, m.xyzs, m.dirs, m.deltas,
            m.rays, m.perturb, m.log, m.nlog, m.counter, m.zero_unwritten, ray_blocks);
    }
    ENERF_LAUNCH_CHECK("march_rays_train (carried)");
    return 0;
}
