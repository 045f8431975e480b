// The query points of a full density-grid sweep (NeRFRenderer.update_extra_state, nerf/renderer.py:484-512): one
// uniformly drawn point inside every cell of every cascade, x fastest.  Shared by density_update.hip, which writes them
// out for networks it does not know, and gridencoder.hip, whose forward kernel can generate them in place of reading
// an input array (a 6.3 M-point sweep otherwise writes 75 MB of positions and reads them once per level).
#pragma once
#include <math.h>
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace enerf {

__host__ __device__ inline uint32_t sweep_expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__host__ __device__ inline uint32_t sweep_morton3(uint32_t x, uint32_t y, uint32_t z) {
    return sweep_expand_bits(x) | (sweep_expand_bits(y) << 1) | (sweep_expand_bits(z) << 2);
}

// counter-based generator: four independent 32-bit words per (seed, counter) -- splitmix64 finaliser, two rounds
__host__ __device__ inline uint64_t sweep_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
struct Rand4 {
    uint32_t w[4];
};
__host__ __device__ inline Rand4 rand4(uint64_t seed, uint64_t counter) {
    const uint64_t a = sweep_mix64(seed + 0x9e3779b97f4a7c15ULL * (2 * counter + 1));
    const uint64_t b = sweep_mix64(a + 0x9e3779b97f4a7c15ULL * (2 * counter + 2) + seed);
    return Rand4{{(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)}};
}
__host__ __device__ inline float unit_float(uint32_t w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }   // [0, 1)

struct Cascades {
    float span[8];      // bound_c - half_grid_size
    float half[8];      // half_grid_size = bound_c / H
};
inline Cascades make_cascades(uint32_t C, uint32_t H, float bound) {
    Cascades cs;
    for (uint32_t c = 0; c < 8; c++) {
        // renderer.py:498-501  bound = min(2 ** cas, self.bound); half_grid_size = bound / self.grid_size
        const double b = fmin((double)(1u << c), (double)bound);
        const double half = b / (double)H;
        cs.span[c] = c < C ? (float)(b - half) : 0.0f;
        cs.half[c] = c < C ? (float)half : 0.0f;
    }
    return cs;
}

// query position of cell (x, y, z) of cascade `cas`: 2 * c / (H - 1) - 1, scaled to the cascade, jittered inside the cell
__host__ __device__ inline void cell_position(const Cascades& cs, uint32_t cas, uint32_t H, uint32_t x, uint32_t y,
                                              uint32_t z, const Rand4& r, float* out) {
    const float inv = 1.0f / (float)(H - 1);
    const float span = cs.span[cas], half = cs.half[cas];
    out[0] = (2.0f * (float)x * inv - 1.0f) * span + (unit_float(r.w[0]) * 2.0f - 1.0f) * half;
    out[1] = (2.0f * (float)y * inv - 1.0f) * span + (unit_float(r.w[1]) * 2.0f - 1.0f) * half;
    out[2] = (2.0f * (float)z * inv - 1.0f) * span + (unit_float(r.w[2]) * 2.0f - 1.0f) * half;
}

// point p of the full sweep (cascade = p / H^3, cell = p % H^3 with x fastest): position, and its Morton cell index
struct SweepGen {
    Cascades cs;
    uint64_t seed;
    uint32_t H, logH, enabled;
};
__host__ __device__ inline uint32_t sweep_cell(const SweepGen& g, uint32_t p, uint32_t& cas, uint32_t& x, uint32_t& y,
                                               uint32_t& z) {
    const uint32_t H3 = 1u << (3 * g.logH);
    cas = p >> (3 * g.logH);
    const uint32_t cell = p & (H3 - 1u);
    x = cell & (g.H - 1u);
    y = (cell >> g.logH) & (g.H - 1u);
    z = cell >> (2 * g.logH);
    return sweep_morton3(x, y, z);
}
__host__ __device__ inline void sweep_point(const SweepGen& g, uint32_t p, float* out) {
    uint32_t cas, x, y, z;
    sweep_cell(g, p, cas, x, y, z);
    cell_position(g.cs, cas, g.H, x, y, z, rand4(g.seed, p), out);
}

}  // namespace enerf
