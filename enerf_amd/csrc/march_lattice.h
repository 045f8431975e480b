// march_lattice.h -- the device side of the fixed-step ("lattice") training marcher that more than one translation unit needs:
// ray set-up, the occupancy-cell evaluation, the wave-per-ray lattice march with its chunk log, the replay, the occupied-box
// clip, and the count pass as a workgroup-level function.  raymarching.hip builds its kernels from it; gridencoder.hip lets
// k_grid_tile_adam's launch carry the NEXT batch's count pass in extra workgroups (enerf::tile_adam_carry_count, common.h), so
// that in the one-call training step the march needs no second stream: the optimizer's launch counts, one small launch behind
// it scans and writes.  Include INSIDE the translation unit's anonymous namespace, after common.h.
// (reference: raymarching/src/raymarching.cu:320-478 march_rays_train; line citations at the functions)
#pragma once
#include <float.h>

constexpr float kSqrt3 = 1.7320508075688772f;
constexpr float kRPi = 0.3183098861837907f;

__device__ __forceinline__ float signf_(float x) { return copysignf(1.0f, x); }
__device__ __forceinline__ float clampf_(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

__device__ __forceinline__ uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton3(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
__device__ __forceinline__ uint32_t morton3_inv(uint32_t x) {
    x = x & 0x49249249;
    x = (x | (x >> 2)) & 0xc30c30c3;
    x = (x | (x >> 4)) & 0x0f00f00f;
    x = (x | (x >> 8)) & 0xff0000ff;
    x = (x | (x >> 16)) & 0x0000ffff;
    return x;
}

// exponent e with |v| in [2^(e-1), 2^e), 0 for v == 0 (== frexpf's exponent), clamped to [0, C-1]
__device__ __forceinline__ int mip_exponent(float v, uint32_t C) {
    int e;
    (void)frexpf(v, &e);
    return (int)fminf((float)C - 1.0f, fmaxf(0.0f, (float)e));
}

// PCG32 (XSH-RR 64/32), seed(initstate, initseq) then one next_float()
__device__ __forceinline__ uint32_t pcg_next(uint64_t& state, uint64_t inc) {
    const uint64_t old = state;
    state = old * 0x5851f42d4c957f2dULL + inc;
    const uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    const uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}
__device__ __forceinline__ float pcg_first_float(uint64_t initstate, uint64_t initseq) {
    uint64_t state = 0U;
    const uint64_t inc = (initseq << 1u) | 1u;
    (void)pcg_next(state, inc);
    state += initstate;
    (void)pcg_next(state, inc);
    const uint32_t u = (pcg_next(state, inc) >> 9) | 0x3f800000u;
    return __uint_as_float(u) - 1.0f;
}

struct RayCtx {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
    float bound, dt_gamma, dt_min, dt_max;
    uint32_t C, H;
    const uint8_t* grid;
};

__device__ __forceinline__ void ray_ctx_init(RayCtx& c, const float* o, const float* d, const uint8_t* grid,
                                             float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H) {
    c.ox = o[0]; c.oy = o[1]; c.oz = o[2];
    c.dx = d[0]; c.dy = d[1]; c.dz = d[2];
    c.rdx = 1 / c.dx; c.rdy = 1 / c.dy; c.rdz = 1 / c.dz;
    c.bound = bound; c.dt_gamma = dt_gamma;
    c.dt_min = 2 * kSqrt3 / max_steps;
    c.dt_max = 2 * kSqrt3 * (1 << (C - 1)) / H;
    c.C = C; c.H = H; c.grid = grid;
}

// One evaluation of the marching loop body at parameter t: sample position, step size, occupancy bit of the cell and,
// for an empty cell, the parameter `tt` of the cell's exit face (with the reference's (H - 1) quirk).
__device__ __forceinline__ bool eval_cell(const RayCtx& c, float t, float& x, float& y, float& z, float& dt, float& tt) {
    const float bound = c.bound;
    const uint32_t H = c.H;
    x = clampf_(fmaf(t, c.dx, c.ox), -bound, bound);
    y = clampf_(fmaf(t, c.dy, c.oy), -bound, bound);
    z = clampf_(fmaf(t, c.dz, c.oz), -bound, bound);
    dt = clampf_(t * c.dt_gamma, c.dt_min, c.dt_max);
    const int lp = mip_exponent(fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z))), c.C);
    const int ld = mip_exponent((float)((double)(dt * (float)H) * 0.5), c.C);
    const int level = lp > ld ? lp : ld;
    const float mip_bound = fminf((float)(1 << level), bound);
    const float mip_rbound = 1 / mip_bound;
    const float hm1 = (float)(H - 1);
    const int nx = (int)clampf_((float)(0.5 * (double)fmaf(x, mip_rbound, 1.0f) * (double)H), 0.0f, hm1);
    const int ny = (int)clampf_((float)(0.5 * (double)fmaf(y, mip_rbound, 1.0f) * (double)H), 0.0f, hm1);
    const int nz = (int)clampf_((float)(0.5 * (double)fmaf(z, mip_rbound, 1.0f) * (double)H), 0.0f, hm1);
    const uint32_t index = (uint32_t)level * H * H * H + morton3((uint32_t)nx, (uint32_t)ny, (uint32_t)nz);
    const bool occ = (c.grid[index >> 3] & (1u << (index & 7u))) != 0;
    const float tx = fmaf(fmaf((nx + 0.5f + 0.5f * signf_(c.dx)) / hm1, 2.0f, -1.0f), mip_bound, -x) * c.rdx;
    const float ty = fmaf(fmaf((ny + 0.5f + 0.5f * signf_(c.dy)) / hm1, 2.0f, -1.0f), mip_bound, -y) * c.rdy;
    const float tz = fmaf(fmaf((nz + 0.5f + 0.5f * signf_(c.dz)) / hm1, 2.0f, -1.0f), mip_bound, -z) * c.rdz;
    tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    return occ;
}

// Sequential form (one thread per ray): on an empty cell t_skip receives t after the voxel-skipping do/while.
__device__ __forceinline__ bool eval_step(const RayCtx& c, float t, float& x, float& y, float& z, float& dt,
                                          float& t_skip) {
    float tt;
    const bool occ = eval_cell(c, t, x, y, z, dt, tt);
    if (!occ) {
        do {
            t += clampf_(t * c.dt_gamma, c.dt_min, c.dt_max);
        } while (t < tt);
        t_skip = t;
    }
    return occ;
}

// Fixed-step (dt_gamma == 0), power-of-two H <= kTabH form of eval_cell for the wave-per-ray marchers: the same values
// bit for bit, from far fewer instructions.
//   * dt == dt_min and the level bound from it are constants of the ray (RayFixed).
//   * 1 / mip_bound: 2^-level exactly below the clamp, the ray constant 1 / bound at it.
//   * 0.5 * (double)f * (double)H == f * (0.5f * H) exactly when H is a power of two (both scalings are exact).
//   * the voxel-face coordinate (n + 0.5 + 0.5 sign) / (H - 1) * 2 - 1 only takes the H + 1 values n' = n or n + 1:
//     a table in LDS (face_tab, built per workgroup with the reference's own float expression) replaces three IEEE
//     divisions per lattice point; expand_tab does the same for the Morton bit spreading.
constexpr uint32_t kTabH = 256;
struct MarchTabs {
    const float* face;        // [H + 1]
    const uint32_t* expand;   // [H]
};
struct RayFixed {
    int ld;                   // level bound from dt_min
    float rbound;             // 1 / bound
    float half_h;             // 0.5f * H
    int sx, sy, sz;           // 1 where the direction component is positive (copysign semantics), else 0
};
__device__ __forceinline__ bool march_fast_ok(uint32_t H) { return H <= kTabH && (H & (H - 1u)) == 0u; }
__device__ __forceinline__ void build_march_tabs(float* face, uint32_t* expand, uint32_t H) {
    const float hm1 = (float)(H - 1);
    for (uint32_t k = threadIdx.x; k <= H; k += blockDim.x) face[k] = fmaf((float)k / hm1, 2.0f, -1.0f);
    for (uint32_t k = threadIdx.x; k < H; k += blockDim.x) expand[k] = expand_bits(k);
    __syncthreads();
}
__device__ __forceinline__ void ray_fixed_init(RayFixed& f, const RayCtx& c) {
    f.ld = mip_exponent((float)((double)(c.dt_min * (float)c.H) * 0.5), c.C);
    f.rbound = 1 / c.bound;
    f.half_h = 0.5f * (float)c.H;
    f.sx = signf_(c.dx) > 0 ? 1 : 0;
    f.sy = signf_(c.dy) > 0 ? 1 : 0;
    f.sz = signf_(c.dz) > 0 ? 1 : 0;
}
__device__ __forceinline__ bool eval_cell_fixed(const RayCtx& c, const RayFixed& f, const MarchTabs& tb, float t, float& x,
                                                float& y, float& z, float& tt) {
    const float bound = c.bound;
    const uint32_t H = c.H;
    x = clampf_(fmaf(t, c.dx, c.ox), -bound, bound);
    y = clampf_(fmaf(t, c.dy, c.oy), -bound, bound);
    z = clampf_(fmaf(t, c.dz, c.oz), -bound, bound);
    const int lp = mip_exponent(fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z))), c.C);
    const int level = lp > f.ld ? lp : f.ld;
    const float pw = (float)(1 << level);
    const bool clamped = pw > bound;                                  // mip_bound = min(2^level, bound)
    const float mip_bound = clamped ? bound : pw;
    const float mip_rbound = clamped ? f.rbound : __int_as_float((127 - level) << 23);   // 2^-level
    const float hm1 = (float)(H - 1);
    const int nx = (int)clampf_(fmaf(x, mip_rbound, 1.0f) * f.half_h, 0.0f, hm1);
    const int ny = (int)clampf_(fmaf(y, mip_rbound, 1.0f) * f.half_h, 0.0f, hm1);
    const int nz = (int)clampf_(fmaf(z, mip_rbound, 1.0f) * f.half_h, 0.0f, hm1);
    const uint32_t index = (uint32_t)level * H * H * H + (tb.expand[nx] | (tb.expand[ny] << 1) | (tb.expand[nz] << 2));
    const bool occ = (c.grid[index >> 3] & (1u << (index & 7u))) != 0;
    const float tx = fmaf(tb.face[nx + f.sx], mip_bound, -x) * c.rdx;
    const float ty = fmaf(tb.face[ny + f.sy], mip_bound, -y) * c.rdy;
    const float tz = fmaf(tb.face[nz + f.sz], mip_bound, -z) * c.rdz;
    tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    return occ;
}

// ------------------------------------------------------------------ small per-element kernels
// near / far of one ray against the box (raymarching.cu:94-136); false: the ray misses it (both are FLT_MAX then)
__device__ __forceinline__ bool near_far_of(float ox, float oy, float oz, float dx, float dy, float dz,
                                            const float* __restrict__ aabb, float min_near, float& near, float& far) {
    const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
    float tmp;
    near = (aabb[0] - ox) * rdx; far = (aabb[3] - ox) * rdx;
    if (near > far) { tmp = near; near = far; far = tmp; }
    float near_y = (aabb[1] - oy) * rdy, far_y = (aabb[4] - oy) * rdy;
    if (near_y > far_y) { tmp = near_y; near_y = far_y; far_y = tmp; }
    if (near > far_y || near_y > far) { near = far = FLT_MAX; return false; }
    if (near_y > near) near = near_y;
    if (far_y < far) far = far_y;
    float near_z = (aabb[2] - oz) * rdz, far_z = (aabb[5] - oz) * rdz;
    if (near_z > far_z) { tmp = near_z; near_z = far_z; far_z = tmp; }
    if (near > far_z || near_z > far) { near = far = FLT_MAX; return false; }
    if (near_z > near) near = near_z;
    if (far_z < far) far = far_z;
    if (near < min_near) near = min_near;
    return true;
}

// ------------------------------------------------------------------ march_rays_train
// Pass structure (replaces the reference's count -> 2 global atomics -> write in one thread):
//   k_march_count : num_steps[n]  (written to rays[n][2]); the wave-per-ray variant also logs its emitting chunks
//   k_march_scan  : single workgroup exclusive scan -> rays[n] = (n, base + excl, num_steps); counter update
//   k_march_write : emit samples at the scanned offset (wave-per-ray: replay of the chunk log; thread-per-ray: re-march)
// so offsets are those of sequential execution and reproducible run to run.
template <bool WRITE>
__device__ __forceinline__ uint32_t march_one_ray(const RayCtx& c, float t0, float far, uint32_t limit, float* xyzs,
                                                  float* dirs, float* deltas) {
    float t = t0, last_t = t0, x, y, z, dt, ts = 0.0f;
    uint32_t step = 0;
    while (t < far && step < limit) {
        if (eval_step(c, t, x, y, z, dt, ts)) {
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
            t = ts;
        }
    }
    return step;
}

// ------------------------------------------------------------------ wave-per-ray "lattice" marcher (dt_gamma == 0)
// With dt_gamma == 0 every update of t in the reference's loop -- the occupied step and each iteration of the
// empty-voxel skip loop alike -- is t += dt_min, so the values t can take form a fixed lattice t_0, t_1 = fl(t_0 + dt),
// ... that does not depend on occupancy; occupancy only decides which lattice points are evaluated and which are
// emitted.  Inside one binade of t, fl(t + dt) - t is the same multiple of ulp(t) for every t (ties excepted, which
// are detected), so 64 consecutive lattice points are t_base + lane * delta EXACTLY and all 64 lanes of a wavefront
// evaluate their cells in parallel.  The sequential control flow (emit on occupied, jump to the first lattice point
// >= the voxel exit on empty) is then replayed on wave-uniform bit masks: runs of occupied lanes are emitted whole,
// empty lanes jump through a precomputed per-lane "next" index.  The emitted samples, their count and their
// positions are bit-identical to the one-thread-per-ray loop above; the work per ray is spread over 64 lanes
// instead of one latency-bound thread.
// Chunk log (count pass -> write pass): the chunks of a ray that emitted anything, as (first lattice point, emit mask).
// Everything the write pass stores -- positions, dt, real delta-t -- is a function of those two and of the ray, so it
// replays the log instead of marching again: no bitfield reads, no serial control-flow replay.
struct ChunkEntry {
    float base;
    uint32_t pad;
    unsigned long long emit;
};
constexpr uint32_t kLogCap = 64;                 // entries per ray; a ray that needs more is re-marched by the write pass
constexpr uint32_t kLogOverflow = 0xffffffffu;

template <bool WRITE, bool LOG = false, bool FAST = false>
__device__ __forceinline__ uint32_t lattice_march(const RayCtx& c, float t0, float far, uint32_t limit, float* xyzs,
                                                  float* dirs, float* deltas, ChunkEntry* log = nullptr,
                                                  uint32_t* nlog = nullptr, const MarchTabs* tabs = nullptr) {
    uint32_t logged = 0;
    RayFixed rf;
    if (FAST) ray_fixed_init(rf, c);
    const int lane = lane_id();
    const float dt = c.dt_min;
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;   // lanes strictly below mine
    float base = t0;                              // wave-uniform: first lattice point of the current chunk
    float tt_pending = -__builtin_huge_valf();    // exit parameter of a skip that ran past the previous chunk
    float last_t = t0;                            // t after the previously emitted sample's step
    uint32_t count = 0;
    while (base < far && count < limit) {
        // increment of the arithmetic progression starting at `base`, and how far it is valid
        const float delta = (base + dt) - base;
        const float delta2 = ((base + delta) + dt) - (base + delta);
        int e;
        (void)frexpf(base, &e);
        const float bin_top = ldexpf(1.0f, e);                                  // base in [bin_top/2, bin_top)
        const bool progression = base >= 2.0f * dt && delta2 == delta;
        const float ti = progression ? fmaf((float)lane, delta, base) : base;   // exact within the binade
        const bool ok = lane == 0 || (progression && ti < bin_top);
        const unsigned long long okm = __ballot(ok && ti < far);
        const int nvalid = okm == ~0ull ? 64 : __builtin_ctzll(~okm);           // leading run of usable lanes (>= 1)
        const unsigned long long vmask = nvalid == 64 ? ~0ull : ((1ull << nvalid) - 1ull);

        float x, y, z, dts, tt;
        bool occ;
        if (FAST) {
            occ = eval_cell_fixed(c, rf, *tabs, ti, x, y, z, tt);
            dts = dt;
        } else {
            occ = eval_cell(c, ti, x, y, z, dts, tt);
        }
        const float t_next = ti + dt;                                           // the true next lattice value
        // per-lane jump target for an empty cell: first lattice index j > lane with !(t_j < tt).  The first guess
        // may be off (FAST: reciprocal instead of a division); the two loops below make it exact either way.
        int nxt = lane + 1;
        if (progression && ti < tt) {
            const float steps = FAST ? (tt - ti) * __builtin_amdgcn_rcpf(delta) : (tt - ti) / delta;
            int j = lane + (int)fminf(fmaxf(ceilf(steps), 1.0f), 64.0f);
            while (j - 1 > lane && !(fmaf((float)(j - 1), delta, base) < tt)) j--;
            while (j < 64 && fmaf((float)j, delta, base) < tt) j++;
            nxt = j;
        }
        const unsigned long long occm = __ballot(occ) & vmask;
        // first lattice point of this chunk not skipped by a pending voxel exit
        const unsigned long long reach = __ballot(!(ti < tt_pending)) & vmask;
        int cur = reach ? __builtin_ctzll(reach) : nvalid;
        if (cur < nvalid) tt_pending = -__builtin_huge_valf();
        unsigned long long emit = 0ull;
        uint32_t room = limit - count;
        while (cur < nvalid && room > 0) {
            if ((occm >> cur) & 1ull) {
                const unsigned long long rest = ~(occm >> cur);
                uint32_t run = rest ? (uint32_t)__builtin_ctzll(rest) : (uint32_t)(64 - cur);
                if (run > room) run = room;
                emit |= (run == 64 ? ~0ull : ((1ull << run) - 1ull)) << cur;
                cur += (int)run;
                room -= run;
            } else {
                const int j = __builtin_amdgcn_readlane(nxt, cur);
                if (j >= nvalid) {
                    tt_pending = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tt), cur));
                    cur = nvalid;
                } else {
                    cur = j;
                }
            }
        }
        const uint32_t nemit = (uint32_t)__popcll(emit);
        if (nemit) {
            const int top = 63 - __builtin_clzll(emit);
            if (WRITE) {
                const unsigned long long before = emit & below;
                const int prev = before ? 63 - __builtin_clzll(before) : 0;
                const float prev_next = __shfl(t_next, prev, 64);
                if ((emit >> lane) & 1ull) {
                    const size_t k = (size_t)count + (uint32_t)__popcll(before);
                    xyzs[k * 3] = x; xyzs[k * 3 + 1] = y; xyzs[k * 3 + 2] = z;
                    dirs[k * 3] = c.dx; dirs[k * 3 + 1] = c.dy; dirs[k * 3 + 2] = c.dz;
                    deltas[k * 2] = dts;
                    deltas[k * 2 + 1] = t_next - (before ? prev_next : last_t);
                }
            }
            last_t = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t_next), top));
            count += nemit;
            if (LOG) {
                if (logged < kLogCap && lane == 0) {
                    log[logged].base = base;
                    log[logged].emit = emit;
                }
                logged++;
            }
        }
        base = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t_next), nvalid - 1));
    }
    if (LOG && lane == 0) *nlog = logged <= kLogCap ? logged : kLogOverflow;
    return count;
}

// The same marcher for the fixed-step, table-friendly case (FAST above), restructured around what actually bounds it.
// Counters on the plain version: 2.2 scalar instructions per vector instruction and 40 % of the wave cycles waiting --
// a wave spends its time in (a) the latency of the occupancy-byte load each chunk starts with and (b) the serial
// replay, one scalar round trip (readlane, compare, branch) per voxel crossed.  Hence:
//   * software pipelining: the lattice does not depend on occupancy, so the next chunk's geometry is computed and its
//     occupancy bytes are requested BEFORE the current chunk is replayed;
//   * pointer doubling: every lane learns in 6 shuffle rounds where the chain of empty-voxel jumps starting at it ends
//     (first occupied lane reached, or the lane whose jump leaves the chunk), so the scalar replay takes one step per
//     occupied run instead of one per voxel.
// Same visited set, same emitted samples, bit for bit.
struct ChunkGeo {
    float base, delta, ti, x, y, z, tt, t_next;
    int nvalid, nxt;
    unsigned long long vmask;
    uint32_t bit, byte;
};
__device__ __forceinline__ ChunkGeo chunk_geometry(const RayCtx& c, const RayFixed& f, const MarchTabs& tb, float base,
                                                   float far, int lane) {
    ChunkGeo g;
    const float dt = c.dt_min;
    g.base = base;
    g.delta = (base + dt) - base;
    const float delta2 = ((base + g.delta) + dt) - (base + g.delta);
    int e;
    (void)frexpf(base, &e);
    const float bin_top = ldexpf(1.0f, e);                                  // base in [bin_top/2, bin_top)
    const bool progression = base >= 2.0f * dt && delta2 == g.delta;
    g.ti = progression ? fmaf((float)lane, g.delta, base) : base;           // exact within the binade
    const bool ok = lane == 0 || (progression && g.ti < bin_top);
    const unsigned long long okm = __ballot(ok && g.ti < far);
    g.nvalid = okm == ~0ull ? 64 : __builtin_ctzll(~okm);                   // leading run of usable lanes (>= 1)
    g.vmask = g.nvalid == 64 ? ~0ull : ((1ull << g.nvalid) - 1ull);
    g.t_next = g.ti + dt;

    // eval_cell_fixed, with the occupancy byte only requested here
    const float bound = c.bound;
    const uint32_t H = c.H;
    g.x = clampf_(fmaf(g.ti, c.dx, c.ox), -bound, bound);
    g.y = clampf_(fmaf(g.ti, c.dy, c.oy), -bound, bound);
    g.z = clampf_(fmaf(g.ti, c.dz, c.oz), -bound, bound);
    const int lp = mip_exponent(fmaxf(fabsf(g.x), fmaxf(fabsf(g.y), fabsf(g.z))), c.C);
    const int level = lp > f.ld ? lp : f.ld;
    const float pw = (float)(1 << level);
    const bool clamped = pw > bound;
    const float mip_bound = clamped ? bound : pw;
    const float mip_rbound = clamped ? f.rbound : __int_as_float((127 - level) << 23);
    const float hm1 = (float)(H - 1);
    const int nx = (int)clampf_(fmaf(g.x, mip_rbound, 1.0f) * f.half_h, 0.0f, hm1);
    const int ny = (int)clampf_(fmaf(g.y, mip_rbound, 1.0f) * f.half_h, 0.0f, hm1);
    const int nz = (int)clampf_(fmaf(g.z, mip_rbound, 1.0f) * f.half_h, 0.0f, hm1);
    const uint32_t index = (uint32_t)level * H * H * H + (tb.expand[nx] | (tb.expand[ny] << 1) | (tb.expand[nz] << 2));
    g.bit = index & 7u;
    g.byte = c.grid[index >> 3];
    const float tx = fmaf(tb.face[nx + f.sx], mip_bound, -g.x) * c.rdx;
    const float ty = fmaf(tb.face[ny + f.sy], mip_bound, -g.y) * c.rdy;
    const float tz = fmaf(tb.face[nz + f.sz], mip_bound, -g.z) * c.rdz;
    g.tt = g.ti + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    // jump target of an empty cell: first lattice index j > lane with !(t_j < tt) (the loops make any guess exact)
    g.nxt = lane + 1;
    if (progression && g.ti < g.tt) {
        int j = lane + (int)fminf(fmaxf(ceilf((g.tt - g.ti) * __builtin_amdgcn_rcpf(g.delta)), 1.0f), 64.0f);
        while (j - 1 > lane && !(fmaf((float)(j - 1), g.delta, base) < g.tt)) j--;
        while (j < 64 && fmaf((float)j, g.delta, base) < g.tt) j++;
        g.nxt = j;
    }
    return g;
}

template <bool WRITE, bool LOG>
__device__ __forceinline__ uint32_t lattice_march_fast(const RayCtx& c, const MarchTabs& tabs, float t0, float far,
                                                       uint32_t limit, float* xyzs, float* dirs, float* deltas,
                                                       ChunkEntry* log, uint32_t* nlog) {
    RayFixed rf;
    ray_fixed_init(rf, c);
    const int lane = lane_id();
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    float tt_pending = -__builtin_huge_valf();
    float last_t = t0;
    uint32_t count = 0, logged = 0;
    if (t0 < far && limit > 0) {
        ChunkGeo A = chunk_geometry(c, rf, tabs, t0, far, lane);
        while (true) {
            const float base_next = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(A.t_next), A.nvalid - 1));
            const bool have_next = base_next < far;
            ChunkGeo Bn = A;
            if (have_next) Bn = chunk_geometry(c, rf, tabs, base_next, far, lane);     // its loads fly during the replay

            const bool occ = ((A.byte >> A.bit) & 1u) != 0;
            const unsigned long long occm = __ballot(occ) & A.vmask;
            // pointer doubling over the empty lanes: ps = (lane reached << 8) | lane whose jump got there
            int ps = (occ || lane >= A.nvalid) ? ((lane << 8) | lane) : (((A.nxt < 64 ? A.nxt : 64) << 8) | lane);
#pragma unroll
            for (int r = 0; r < 6; r++) {
                const int P = ps >> 8;
                const int q = __shfl(ps, P < 64 ? P : 63, 64);
                if (P < A.nvalid && (q >> 8) != P) ps = q;       // landed on an empty lane inside the chunk: keep going
            }
            const unsigned long long reach = __ballot(!(A.ti < tt_pending)) & A.vmask;
            int cur = reach ? __builtin_ctzll(reach) : A.nvalid;
            if (cur < A.nvalid) tt_pending = -__builtin_huge_valf();
            unsigned long long emit = 0ull;
            uint32_t room = limit - count;
            while (cur < A.nvalid && room > 0) {
                if ((occm >> cur) & 1ull) {
                    const unsigned long long rest = ~(occm >> cur);
                    uint32_t run = rest ? (uint32_t)__builtin_ctzll(rest) : (uint32_t)(64 - cur);
                    if (run > room) run = room;
                    emit |= (run == 64 ? ~0ull : ((1ull << run) - 1ull)) << cur;
                    cur += (int)run;
                    room -= run;
                } else {
                    const int e = __builtin_amdgcn_readlane(ps, cur);
                    if ((e >> 8) >= A.nvalid) {
                        tt_pending = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(A.tt), e & 0xff));
                        cur = A.nvalid;
                    } else {
                        cur = e >> 8;
                    }
                }
            }
            const uint32_t nemit = (uint32_t)__popcll(emit);
            if (nemit) {
                const int top = 63 - __builtin_clzll(emit);
                if (WRITE) {
                    const unsigned long long before = emit & below;
                    const int prev = before ? 63 - __builtin_clzll(before) : 0;
                    const float prev_next = __shfl(A.t_next, prev, 64);
                    if ((emit >> lane) & 1ull) {
                        const size_t k = (size_t)count + (uint32_t)__popcll(before);
                        xyzs[k * 3] = A.x; xyzs[k * 3 + 1] = A.y; xyzs[k * 3 + 2] = A.z;
                        dirs[k * 3] = c.dx; dirs[k * 3 + 1] = c.dy; dirs[k * 3 + 2] = c.dz;
                        deltas[k * 2] = c.dt_min;
                        deltas[k * 2 + 1] = A.t_next - (before ? prev_next : last_t);
                    }
                }
                last_t = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(A.t_next), top));
                count += nemit;
                if (LOG) {
                    if (logged < kLogCap && lane == 0) {
                        log[logged].base = A.base;
                        log[logged].emit = emit;
                    }
                    logged++;
                }
            }
            if (!have_next || count >= limit) break;
            A = Bn;
        }
    }
    if (LOG && lane == 0) *nlog = logged <= kLogCap ? logged : kLogOverflow;
    return count;
}

// Write pass of the wave-per-ray marcher: replay a ray's chunk log.
__device__ __forceinline__ void lattice_replay(const RayCtx& c, float t0, const ChunkEntry* log, uint32_t nlog,
                                               float* xyzs, float* dirs, float* deltas) {
    const int lane = lane_id();
    const float dt = c.dt_min;
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    float last_t = t0;
    uint32_t count = 0;
    for (uint32_t e = 0; e < nlog; e++) {
        const float base = log[e].base;
        const unsigned long long emit = log[e].emit;
        const float delta = (base + dt) - base;
        const float delta2 = ((base + delta) + dt) - (base + delta);
        const bool progression = base >= 2.0f * dt && delta2 == delta;
        const float ti = progression ? fmaf((float)lane, delta, base) : base;
        const float t_next = ti + dt;
        const unsigned long long before = emit & below;
        const int prev = before ? 63 - __builtin_clzll(before) : 0;
        const float prev_next = __shfl(t_next, prev, 64);
        if ((emit >> lane) & 1ull) {
            const size_t k = (size_t)count + (uint32_t)__popcll(before);
            xyzs[k * 3] = clampf_(fmaf(ti, c.dx, c.ox), -c.bound, c.bound);
            xyzs[k * 3 + 1] = clampf_(fmaf(ti, c.dy, c.oy), -c.bound, c.bound);
            xyzs[k * 3 + 2] = clampf_(fmaf(ti, c.dz, c.oz), -c.bound, c.bound);
            dirs[k * 3] = c.dx; dirs[k * 3 + 1] = c.dy; dirs[k * 3 + 2] = c.dz;
            deltas[k * 2] = clampf_(ti * c.dt_gamma, c.dt_min, c.dt_max);
            deltas[k * 2 + 1] = t_next - (before ? prev_next : last_t);
        }
        const int top = 63 - __builtin_clzll(emit);
        last_t = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t_next), top));
        count += (uint32_t)__popcll(emit);
    }
}

// ---- bounding box of the occupied cells ------------------------------------------------------------------------------
// A sample can only be emitted at a position whose cell bit is set, and a cell of cascade level L is only ever consulted
// for positions inside that cell's world-space box (the level is at least the position's own mip level, so the position
// lies within the level's cube and maps to the cell that contains it).  Rays that miss the union of those boxes emit
// nothing, and a ray emits nothing once it has left it -- exactly, whatever the visiting order of the lattice was up to
// there.  The count pass tests each ray against the box (enlarged by two coarsest-level cells) before it marches and stops
// at the box's far side instead of the volume's: on the training cameras of the synthetic scene half of the rays are
// settled by the test and the rest march a third of their chord.
// Keys: order-preserving int image of a float; six running minima (lo.xyz, -hi.xyz), initialised by memset(0x7f).
__device__ __forceinline__ int aabb_key(float v) {
    const int b = __float_as_int(v);
    return b >= 0 ? b : b ^ 0x7fffffff;
}
__device__ __forceinline__ float aabb_unkey(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

// Ray against the enlarged occupied box: false = the ray cannot emit a sample; otherwise `far` is lowered to where the
// ray leaves the box (plus a margin).  Axis-parallel rays: a zero direction component constrains only the origin.
__device__ __forceinline__ bool clip_to_occupied(const RayCtx& c, const int* __restrict__ keys, float& far) {
    const float pad = 4.0f * fminf((float)(1u << (c.C - 1)), c.bound) / (float)c.H;     // two cells of the coarsest level
    const float o[3] = {c.ox, c.oy, c.oz}, d[3] = {c.dx, c.dy, c.dz};
    float t_in = -3.0e38f, t_out = 3.0e38f;
    for (int a = 0; a < 3; a++) {
        const float lo = aabb_unkey(keys[a]) - pad, hi = -aabb_unkey(keys[3 + a]) + pad;
        if (!(lo <= hi)) return false;                                      // no occupied cell at all
        if (fabsf(d[a]) < 1e-12f) {
            if (o[a] < lo || o[a] > hi) return false;
            continue;
        }
        const float r = 1.0f / d[a];
        const float t1 = (lo - o[a]) * r, t2 = (hi - o[a]) * r;
        t_in = fmaxf(t_in, fminf(t1, t2));
        t_out = fminf(t_out, fmaxf(t1, t2));
    }
    const float slack = 1e-3f * c.bound + 1e-5f * fabsf(t_out);
    if (t_in > t_out + slack) return false;
    far = fminf(far, t_out + slack);
    return true;
}

// The count pass of the wave-per-ray lattice marcher, as workgroup `bid` of `nblocks` workgroups of 256 threads (one ray per
// wavefront and pass; with fewer wavefronts than rays each walks its share): rays[n][2] = samples of ray n, chunk log filled.
// `s_face` / `s_expand`: kTabH + 1 floats / kTabH words of the caller's LDS.  (`a`: common.h MarchCountJob)
__device__ __forceinline__ void march_count_block(const enerf::MarchCountJob& a, uint32_t bid, uint32_t nblocks, float* s_face,
                                                  uint32_t* s_expand) {
    const float* __restrict__ rays_o = a.rays_o;
    const float* __restrict__ rays_d = a.rays_d;
    const uint8_t* __restrict__ grid = a.grid;
    const float bound = a.bound;
    const uint32_t max_steps = a.max_steps, N = a.N, C = a.C, H = a.H, perturb = a.perturb;
    const float* nears = a.nears;
    const float* fars = a.fars;
    int32_t* rays = a.rays;
    ChunkEntry* __restrict__ log = static_cast<ChunkEntry*>(a.log);
    uint32_t* __restrict__ nlog = a.nlog;
    const int* __restrict__ occ_keys = a.occ_keys;
    const float* __restrict__ nf_aabb = a.nf_aabb;
    const float nf_min_near = a.nf_min_near;
    float* nf_nears = a.nf_nears;
    float* nf_fars = a.nf_fars;

    const bool fast = march_fast_ok(H);
    if (fast) build_march_tabs(s_face, s_expand, H);
    const MarchTabs tabs = {s_face, s_expand};
    // one ray per wavefront and pass; a launch with fewer wavefronts than rays (background mode) walks the rest
    const uint32_t nw = nblocks * (blockDim.x >> 6);
    for (uint32_t n = __builtin_amdgcn_readfirstlane(bid * (blockDim.x >> 6) + (threadIdx.x >> 6)); n < N;
         n += nw) {
        RayCtx c;
        ray_ctx_init(c, rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, grid, bound, 0.0f, max_steps, C, H);
        float t0, far;
        if (nf_aabb) {
            // near_far_from_aabb of this ray, here (enerf_march_fuse_near_far): k_near_far's own arithmetic, every lane the same
            (void)near_far_of(c.ox, c.oy, c.oz, c.dx, c.dy, c.dz, nf_aabb, nf_min_near, t0, far);
            if (lane_id() == 0) { nf_nears[n] = t0; nf_fars[n] = far; }
        } else {
            t0 = nears[n];
            far = fars[n];
        }
        if (perturb) t0 = fmaf(c.dt_min, pcg_first_float((uint64_t)n, 1u), t0);   // contracted by the reference's compiler (:351)
        uint32_t cnt = 0;
        if (occ_keys && !clip_to_occupied(c, occ_keys, far)) {
            if (lane_id() == 0) nlog[n] = 0;                                   // nothing to replay
        } else {
            cnt = fast ? lattice_march_fast<false, true>(c, tabs, t0, far, max_steps, nullptr, nullptr, nullptr,
                                                         log + (size_t)n * kLogCap, nlog + n)
                       : lattice_march<false, true, false>(c, t0, far, max_steps, nullptr, nullptr, nullptr,
                                                           log + (size_t)n * kLogCap, nlog + n);
        }
        if (lane_id() == 0) rays[(size_t)n * 3 + 2] = (int32_t)cnt;
    }
}
