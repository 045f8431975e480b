// gridencoder.hip -- multiresolution hash / tiled grid encoder for gfx950.
//
// Replaces gridencoder/src/gridencoder.cu of the reference (grid_encode_forward / grid_encode_backward).
//
// MI355X mapping
//  * One thread per (point, level), 256 points per workgroup.  The forward kernel decodes the flat workgroup id so
//    that each pair of XCDs (observed placement: id % 8) owns a quarter of the levels -- dealt in snake order from the
//    finest level down, so every pair gets the same mix of expensive fine and cheap coarse levels -- and walks them
//    one at a time, the two XCDs splitting each level's points.  An XCD's private 4 MiB L2 then holds one level's
//    table at a time (hashed levels are exactly 4 MiB of fp32 pairs) while the 256 MiB Infinity Cache holds the
//    whole 52 MB table.  A different placement changes speed only.
//  * The gather is bound by the number of distinct cache lines a wavefront touches, not by bytes: the two corners
//    that differ in x sit in adjacent rows on every dense level and, on hashed levels, whenever x is even (the x
//    prime is 1, so the row index just flips bit 0) -- those lanes fetch both with one double-width load.
//  * Per-level scale / resolution are computed on the host (glibc exp2f, ceil) and passed by value, so the
//    uint32 index arithmetic is identical to the CPU oracle's; the kernel never calls exp2f.
//  * A corner's C features are fetched with one C*sizeof(T)-byte load (8 B for the fp32 C=2 configuration),
//    an adjacent corner pair with one 2*C*sizeof(T)-byte load.
//  * Output layouts: [L,B,C] (the reference's), [B,L*C] (what its Python wrapper permutes to) and [L,Bp,C] with
//    Bp = B rounded up to 32 and zeroed pad rows, which the fused MLP (mlp32.hip) consumes directly.
//  * Backward scatters with hardware float atomics (global_atomic_add_f32, -munsafe-fp-atomics) or packed
//    half2 atomics for fp16 tables.
//
// Compiled with -ffp-contract=off; fused multiply-adds are explicit.
#include <hip/hip_ext.h>
#include <hip/hip_fp16.h>
#include <math.h>

#include <type_traits>

#include "common.h"
#include "sweep_points.h"

using namespace enerf;

namespace {

#ifdef ENERF_TA_TIMING
// development aid (tools/dev/ta_tiles.py): per tile of the last k_grid_tile_adam launch -- level, records, begin, end (100 MHz)
__device__ uint32_t g_ta_log[4 * 4096];
__device__ uint32_t g_ta_wg[2 * 2048];          // per workgroup: entry, exit
__device__ unsigned long long g_ta_marks[8];    // last launch of grid_fwd / bin / tile_adam: first entry, last exit (100 MHz)
#define TA_MARK_IN(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_ta_marks[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define TA_MARK_OUT(k) do { if (threadIdx.x == 0) atomicMax(&g_ta_marks[k], (unsigned long long)__builtin_amdgcn_s_memrealtime()); } while (0)
__device__ unsigned long long g_bin_ph[32 * 8];      // binning pass: 100 MHz ticks per phase, summed over workgroups; [7] = workgroups
#define BIN_PH(k) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); atomicAdd(&g_bin_ph[bin_lv * 8 + (k)], t_ - bin_t); bin_t = t_; } } while (0)
#else
#define TA_MARK_IN(k)
#define TA_MARK_OUT(k)
#define BIN_PH(k)
#endif

constexpr int kMaxLevels = 32;
constexpr int kPtsPerBlock = 256;
// points per workgroup of the backward's binning pass (its LDS staging area grows with PTS * 2^D * (1 + C))
// measured at C = 2, 133 k samples: 256 -> 66 us, 512 -> 61 us, 1024 -> 79 us (round 2, three workgroups per CU: 384 -> +9 us)
constexpr int bin_pts(int C) { return C <= 4 ? 512 : 256; }

struct LevelTab {
    float scale[kMaxLevels];
    uint32_t resolution[kMaxLevels];
    uint32_t level_mask;   // profiling aid (enerf_debug_grid_level_mask): levels whose bit is clear are skipped
    float in_add, in_mul;  // inputs are read as (x + in_add) * in_mul: (0, 1) = as given; (bound, 1/(2 bound)) folds
                           // the wrapper's normalisation (grid.py:150) into the kernels, rounded exactly as torch's
                           // two elementwise kernels round it
    // enerf::grid_valid_rows (the training step): the batch is a budget of B rows of which the marcher filled
    // base + min(*valid_rows, cap) (cap == 0: *valid_rows) -- enerf_mlp32_valid_rows' convention, the MLP kernels between
    // this file's forward and backward skip the same rows.  Rows from that count rounded up to the MLP kernels' 32-row
    // tile onwards are neither encoded (nobody reads their features) nor binned (their gradient is zero).
    const int32_t* valid_rows;
    uint32_t valid_base, valid_cap;
};
__device__ __forceinline__ uint32_t grid_row_limit(const LevelTab& tab, uint32_t B) {
    if (!tab.valid_rows) return B;
    int32_t v = tab.valid_rows[0];
    if (tab.valid_cap) v = (int32_t)tab.valid_base + (v <= 0 ? 0 : (v < (int32_t)tab.valid_cap ? v : (int32_t)tab.valid_cap));
    const uint32_t rows = v <= 0 ? 0u : ((uint32_t)v < B ? (uint32_t)v : B);
    const uint32_t up = (rows + 31u) & ~31u;
    return up < B ? up : B;
}

uint32_t g_level_mask = 0xffffffffu;
uint32_t g_binned_min_batch = 16384;     // enerf_debug_grid_bwd_binned
uint32_t g_binned_min_tiles = 8;

uint32_t* g_bin_cursors = nullptr;

uint32_t* bin_cursors();


template <typename T, int C>
struct alignas((sizeof(T) * C) > 16 ? 16 : (sizeof(T) * C)) Feat {
    T v[C];
};

// two adjacent rows; only the row alignment is guaranteed (wide loads tolerate it on gfx950)
template <typename T, int C>
struct __attribute__((packed, aligned((sizeof(T) * C) > 16 ? 16 : (sizeof(T) * C)))) FeatPair {
    Feat<T, C> a, b;
};

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(__half v) { return __half2float(v); }
template <typename T>
__device__ __forceinline__ T from_f(float v);
template <>
__device__ __forceinline__ float from_f<float>(float v) { return v; }
template <>
__device__ __forceinline__ __half from_f<__half>(float v) { return __float2half(v); }
// results[ch] += w * grid[i + ch] and results_grad[ch] += w * (grid[r + ch] - grid[l + ch]) as the reference's types make
// them (gridencoder.cu:136,163,186,210: `scalar_t results[C]`, float w).  fp32 table: one contracted multiply-add.
// at::Half table: the float product is converted to Half for Half's operator+=, which adds in float and rounds again;
// the difference of two table entries is a Half.  `acc` holds the running Half as a float.  Bit for bit the
// reference's kernel (tests/test_gpu_ref_gridencoder.py).
__device__ __forceinline__ float acc_feat(float acc, float w, float g) { return fmaf(w, g, acc); }
__device__ __forceinline__ float acc_feat(float acc, float w, __half g) {
    // (the float product must exist: left alone, the compiler folds multiply + conversion into v_fma_mixlo_f16, which
    //  rounds the exact product to half ONCE -- one entry in 8000 then differs from the reference's by a half ulp)
    float p = w * __half2float(g);
    asm volatile("" : "+v"(p));
    return __half2float(__float2half(acc + __half2float(__float2half(p))));
}
__device__ __forceinline__ float acc_diff(float acc, float w, float r, float l) { return fmaf(w, r - l, acc); }
__device__ __forceinline__ float acc_diff(float acc, float w, __half r, __half l) {
    return acc_feat(acc, w, __float2half(__half2float(r) - __half2float(l)));
}

template <int D>
__device__ __forceinline__ uint32_t fast_hash(const uint32_t (&p)[D]) {
    constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) r ^= p[i] * primes[i];
    return r;
}

// Row index (not yet multiplied by C) of a grid vertex, gridencoder.cu:60-84: dimensions are folded in while the
// running uint32 stride (which may wrap, as in the reference) is <= the level size; a hash level (stride ran past the
// size) uses the xor-prime hash instead; the result is reduced modulo the size.  Everything that depends only on the
// level is resolved once per thread into a LevelGeom (wave-uniform), so the per-vertex work is a few integer ops:
// dense levels need no reduction at all (largest index < size), power-of-two sizes (every hashed level) reduce with
// a mask, and only the remaining cases (tiled grids that overflow) pay for an integer modulo.
template <int D>
struct LevelGeom {
    uint32_t size, mask;
    uint32_t stride[D];     // 0 for dimensions the reference's loop skips
    bool hash;
    int wrap;               // 0: none needed, 1: & mask, 2: % size
};
template <int D>
__device__ __forceinline__ LevelGeom<D> make_geom(uint32_t gridtype, uint32_t size, uint32_t resolution) {
    LevelGeom<D> g;
    g.size = size;
    g.mask = size - 1u;
    uint32_t stride = 1;
    unsigned long long max_index = 0;
#pragma unroll
    for (int d = 0; d < D; d++) {
        if (stride <= size) {
            g.stride[d] = stride;
            max_index += (unsigned long long)resolution * stride;     // vertex coordinates are <= resolution
            stride *= (resolution + 1);
        } else {
            g.stride[d] = 0;
        }
    }
    g.hash = gridtype == 0 && stride > size;
    g.wrap = (size & (size - 1u)) == 0u ? 1 : ((!g.hash && max_index < size) ? 0 : 2);
    return g;
}
template <int D>
__device__ __forceinline__ uint32_t grid_row(const LevelGeom<D>& g, const uint32_t (&p)[D]) {
    uint32_t index;
    if (g.hash) {
        index = fast_hash<D>(p);
    } else {
        index = 0;
#pragma unroll
        for (int d = 0; d < D; d++) index += p[d] * g.stride[d];
    }
    return g.wrap == 0 ? index : (g.wrap == 1 ? (index & g.mask) : index % g.size);
}

// Rows of all 2^D corners of a cell (corner idx adds bit d of idx to coordinate d): the level-mode branches are taken
// once per cell, hashed corners share the per-dimension products (p+1)*prime = p*prime + prime, dense corners are the
// base row plus wave-uniform offsets.  Same values as grid_row on each corner.
template <int D>
__device__ __forceinline__ void corner_rows(const LevelGeom<D>& g, const uint32_t (&p)[D], uint32_t (&rows)[1 << D]) {
    if (g.hash) {
        constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
        uint32_t h[D][2];
#pragma unroll
        for (int d = 0; d < D; d++) {
            h[d][0] = p[d] * primes[d];
            h[d][1] = h[d][0] + primes[d];
        }
#pragma unroll
        for (int idx = 0; idx < (1 << D); idx++) {
            uint32_t r = 0;
#pragma unroll
            for (int d = 0; d < D; d++) r ^= h[d][(idx >> d) & 1];
            rows[idx] = r;
        }
    } else {
        uint32_t base = 0;
#pragma unroll
        for (int d = 0; d < D; d++) base += p[d] * g.stride[d];
#pragma unroll
        for (int idx = 0; idx < (1 << D); idx++) {
            uint32_t off = 0;
#pragma unroll
            for (int d = 0; d < D; d++) off += ((idx >> d) & 1) ? g.stride[d] : 0u;
            rows[idx] = base + off;
        }
    }
    if (g.wrap == 1) {
#pragma unroll
        for (int idx = 0; idx < (1 << D); idx++) rows[idx] &= g.mask;
    } else if (g.wrap == 2) {
#pragma unroll
        for (int idx = 0; idx < (1 << D); idx++) rows[idx] %= g.size;
    }
}

// forward: XCD-pair groups.  Group g = XCDs (2g, 2g+1); round r deals levels L-1-4r .. L-4-4r to the groups in
// alternating direction; the two members take alternate chunks of every level.
// (Rounds walked in pairs -- a group's workgroups alternating between a gather-bound fine level and an instruction-bound
//  coarse one -- are slower, not faster: profiles/r05_fwd_paired_rounds_rejected.txt.)
constexpr uint32_t kGroupXcds = 2, kGroups = 8 / kGroupXcds;
__device__ __forceinline__ bool decode_block_fwd(uint32_t bid, uint32_t nchunks, uint32_t L, uint32_t& level,
                                                 uint32_t& chunk) {
    const uint32_t xcd = bid & 7u, j = bid >> 3;
    const uint32_t group = xcd / kGroupXcds, member = xcd % kGroupXcds;
    const uint32_t per = div_up(nchunks, kGroupXcds);
    const uint32_t round = j / per;
    chunk = (j % per) * kGroupXcds + member;
    const uint32_t dealt = round * kGroups + ((round & 1u) ? (kGroups - 1 - group) : group);
    level = L - 1 - dealt;
    return dealt < L && chunk < nchunks;
}
inline uint32_t fwd_blocks(uint32_t nchunks, uint32_t L) { return 8u * div_up(nchunks, kGroupXcds) * div_up(L, kGroups); }

// backward: XCD k takes level k, then k+8.  (The binning pass too: its lists' partial lines meet in one XCD's L2 --
// dealt over all eight XCDs it takes 75 us instead of 66; pairing the k-th finest with the k-th coarsest level, to
// even out the 23..37 us a level costs, changed nothing in the training step.)
__device__ __forceinline__ bool decode_block(uint32_t nchunks, uint32_t L, uint32_t& level, uint32_t& chunk) {
    const uint32_t bid = blockIdx.x;
    const uint32_t xcd = bid & 7u;
    const uint32_t j = bid >> 3;
    level = xcd + 8u * (j / nchunks);
    chunk = j % nchunks;
    return level < L;
}
__device__ __forceinline__ bool level_enabled(const LevelTab& tab, uint32_t level) {
    return (tab.level_mask >> level) & 1u;
}

// (`bid`: the workgroup's index among the forward's own -- blockIdx.x, less the carried job's workgroups in front)
template <typename T, int D, int C>
__device__ __forceinline__ void grid_fwd_block(uint32_t bid, const float* __restrict__ inputs, const T* __restrict__ grid,
                                               const int32_t* __restrict__ offsets, T* __restrict__ outputs, uint32_t B,
                                               uint32_t L, const LevelTab& tab, bool calc_grad_inputs,
                                               T* __restrict__ dy_dx, uint32_t gridtype, int out_layout, uint32_t nchunks,
                                               const SweepGen& gen) {
    TA_MARK_IN(0);
    uint32_t level, chunk;
    if (!decode_block_fwd(bid, nchunks, L, level, chunk)) return;
    if (!level_enabled(tab, level)) return;
    const uint32_t b = chunk * kPtsPerBlock + threadIdx.x;
    const uint32_t Bp = (B + 31u) & ~31u;
    if (tab.valid_rows && b >= grid_row_limit(tab, B)) return;      // (the budget's unfilled rows: nobody reads them)
    if (b >= B) {
        if (out_layout == 2 && b < Bp) {     // zeroed pad rows of the [L,Bp,C] layout
            Feat<T, C> z;
#pragma unroll
            for (int c = 0; c < C; c++) z.v[c] = from_f<T>(0.0f);
            reinterpret_cast<Feat<T, C>*>(outputs)[(size_t)level * Bp + b] = z;
        }
        return;
    }

    const uint32_t off0 = (uint32_t)offsets[level];
    const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off0;
    const float scale = tab.scale[level];
    const uint32_t resolution = tab.resolution[level];
    const Feat<T, C>* __restrict__ rows = reinterpret_cast<const Feat<T, C>*>(grid) + off0;
    const LevelGeom<D> geom = make_geom<D>(gridtype, hashmap_size, resolution);

    float in[D];
    bool oob = false;
    float raw[D];
    if (D == 3 && gen.enabled) {
        // a full density-grid sweep: point b is generated (sweep_points.h), there is no input array -- every level's
        // workgroups would otherwise read the same 12 bytes per point again
        float q[3];
        sweep_point(gen, b, q);
#pragma unroll
        for (int d = 0; d < D; d++) raw[d] = q[d < 3 ? d : 0];
    } else {
#pragma unroll
        for (int d = 0; d < D; d++) raw[d] = inputs[(size_t)b * D + d];
    }
#pragma unroll
    for (int d = 0; d < D; d++) {
        in[d] = (raw[d] + tab.in_add) * tab.in_mul;
        oob |= (in[d] < 0 || in[d] > 1);
    }
    Feat<T, C>* out = reinterpret_cast<Feat<T, C>*>(outputs) +
                      (out_layout == 0 ? (size_t)level * B + b
                                       : out_layout == 1 ? (size_t)b * L + level : (size_t)level * Bp + b);
    T* jac = calc_grad_inputs ? dy_dx + ((size_t)b * L + level) * D * C : nullptr;
    if (oob) {
        Feat<T, C> z;
#pragma unroll
        for (int c = 0; c < C; c++) z.v[c] = from_f<T>(0.0f);
        *out = z;
        if (jac) {
#pragma unroll
            for (int i = 0; i < D * C; i++) jac[i] = from_f<T>(0.0f);
        }
        return;
    }

    float pos[D];
    uint32_t pos_grid[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        pos[d] = fmaf(in[d], scale, 0.5f);
        const float fl = floorf(pos[d]);
        pos_grid[d] = (uint32_t)fl;
        pos[d] -= (float)pos_grid[d];
    }

    // issue all gathers before using any of them
    Feat<T, C> f[1 << D];
    float w[1 << D];
#pragma unroll
    for (int idx = 0; idx < (1 << D); idx++) {
        float wi = 1;
#pragma unroll
        for (int d = 0; d < D; d++) wi *= (idx & (1 << d)) ? pos[d] : 1 - pos[d];
        w[idx] = wi;
    }
    if constexpr (2 * sizeof(Feat<T, C>) <= 16) {
        // corners 2k (x) and 2k+1 (x+1): one double-width load when their rows are adjacent, else the aligned pair
        // holding the first row plus a single-row load of the second (hashed levels, odd x).  hashmap_size is a
        // multiple of 8, so the aligned pair never leaves the level.
        constexpr int H = 1 << (D - 1);
        uint32_t cr[1 << D], r0[H], r1[H];
        corner_rows<D>(geom, pos_grid, cr);
        bool adj[H];
        FeatPair<T, C> q[H];
#pragma unroll
        for (int k = 0; k < H; k++) {
            r0[k] = cr[2 * k];
            r1[k] = cr[2 * k + 1];
            adj[k] = ((r0[k] ^ r1[k]) == 1u) || (r1[k] == r0[k] + 1u);
            const uint32_t lo = r0[k] < r1[k] ? r0[k] : r1[k];
            q[k] = *reinterpret_cast<const FeatPair<T, C>*>(rows + (adj[k] ? lo : (r0[k] & ~1u)));
        }
        Feat<T, C> e[H];
#pragma unroll
        for (int k = 0; k < H; k++) {
#pragma unroll
            for (int c = 0; c < C; c++) e[k].v[c] = from_f<T>(0.0f);
            if (!adj[k]) e[k] = rows[r1[k]];
        }
#pragma unroll
        for (int k = 0; k < H; k++) {
            const bool first = adj[k] ? (r0[k] < r1[k]) : ((r0[k] & 1u) == 0u);
#pragma unroll
            for (int c = 0; c < C; c++) {
                f[2 * k].v[c] = first ? q[k].a.v[c] : q[k].b.v[c];
                f[2 * k + 1].v[c] = adj[k] ? (first ? q[k].b.v[c] : q[k].a.v[c]) : e[k].v[c];
            }
        }
    } else {
        uint32_t cr[1 << D];
        corner_rows<D>(geom, pos_grid, cr);
#pragma unroll
        for (int idx = 0; idx < (1 << D); idx++) f[idx] = rows[cr[idx]];
    }
    float res[C];
#pragma unroll
    for (int c = 0; c < C; c++) res[c] = 0;
#pragma unroll
    for (int idx = 0; idx < (1 << D); idx++) {
#pragma unroll
        for (int c = 0; c < C; c++) res[c] = acc_feat(res[c], w[idx], f[idx].v[c]);
    }
    Feat<T, C> o;
#pragma unroll
    for (int c = 0; c < C; c++) o.v[c] = from_f<T>(res[c]);
    *out = o;
    TA_MARK_OUT(1);

    if (jac) {
#pragma unroll
        for (int gd = 0; gd < D; gd++) {
            float rg[C];
#pragma unroll
            for (int c = 0; c < C; c++) rg[c] = 0;
#pragma unroll
            for (int idx = 0; idx < (1 << (D - 1)); idx++) {
                float wi = scale;
                uint32_t pgl[D];
#pragma unroll
                for (int nd = 0; nd < D - 1; nd++) {
                    const int d = (nd >= gd) ? (nd + 1) : nd;
                    if ((idx & (1 << nd)) == 0) {
                        wi *= 1 - pos[d];
                        pgl[d] = pos_grid[d];
                    } else {
                        wi *= pos[d];
                        pgl[d] = pos_grid[d] + 1;
                    }
                }
                pgl[gd] = pos_grid[gd];
                const Feat<T, C> fl_ = rows[grid_row<D>(geom, pgl)];
                pgl[gd] = pos_grid[gd] + 1;
                const Feat<T, C> fr_ = rows[grid_row<D>(geom, pgl)];
#pragma unroll
                for (int c = 0; c < C; c++) rg[c] = acc_diff(rg[c], wi, fr_.v[c], fl_.v[c]);
            }
#pragma unroll
            for (int c = 0; c < C; c++) jac[gd * C + c] = from_f<T>(rg[c]);
        }
    }
}

// common.h SplitJob: one workgroup's share (kPtsPerBlock threads of 8 values)
typedef float cj_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 cj_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_job_block(const float* s0, const float* s1, const float* s2, const float* s3,
                                                const float* s4, const uint32_t* __restrict__ map,
                                                uint32_t* __restrict__ out_words, uint32_t threads, uint32_t block) {
    const uint32_t t = block * kPtsPerBlock + threadIdx.x;
    if (t >= threads) return;
    const uint4 m0 = reinterpret_cast<const uint4*>(map)[2 * t], m1 = reinterpret_cast<const uint4*>(map)[2 * t + 1];
    const uint32_t m[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const uint32_t k = m[e] >> 16;
        const float* src = k == 0 ? s0 : k == 1 ? s1 : k == 2 ? s2 : k == 3 ? s3 : s4;
        v[e] = m[e] == 0xffffffffu ? 0.0f : src[m[e] & 0xffffu];
    }
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const cj_f32x2 f = {v[2 * p], v[2 * p + 1]};
        const cj_bf16x2 h2 = __builtin_convertvector(f, cj_bf16x2);               // round to nearest even
        const cj_f32x2 rest = f - __builtin_convertvector(h2, cj_f32x2);          // exact in fp32
        hi[p] = __builtin_bit_cast(uint32_t, h2);
        lo[p] = __builtin_bit_cast(uint32_t, __builtin_convertvector(rest, cj_bf16x2));
    }
    uint4* dst = reinterpret_cast<uint4*>(out_words + (size_t)(t >> 6) * 512u) + (t & 63u);
    dst[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    dst[64] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}
// (carry_blocks: the launch's first workgroups -- a multiple of 8, so that the forward's own keep their XCDs -- do `job`
//  instead, enerf::grid_fwd_carry; 0 everywhere but in the training step's fp32 D = 3, C = 2 launch)
constexpr uint32_t kCarryBlocks = 16;
template <typename T, int D, int C>
__global__ void __launch_bounds__(kPtsPerBlock) k_grid_fwd(const float* __restrict__ inputs, const T* __restrict__ grid,
                                                           const int32_t* __restrict__ offsets, T* __restrict__ outputs,
                                                           uint32_t B, uint32_t L, LevelTab tab, bool calc_grad_inputs,
                                                           T* __restrict__ dy_dx, uint32_t gridtype, int out_layout,
                                                           uint32_t nchunks, SweepGen gen, SplitJob job,
                                                           uint32_t carry_blocks) {
    if constexpr (std::is_same<T, float>::value && D == 3 && C == 2) {
        if (blockIdx.x < carry_blocks) {
            split_job_block(job.src[0], job.src[1], job.src[2], job.src[3], job.src[4], job.map, job.out, job.threads,
                            blockIdx.x);
            return;
        }
    }
    grid_fwd_block<T, D, C>(blockIdx.x - carry_blocks, inputs, grid, offsets, outputs, B, L, tab, calc_grad_inputs, dy_dx,
                            gridtype, out_layout, nchunks, gen);
}

template <int C>
__device__ __forceinline__ void scatter_add(float* row, float w, const float (&g)[C]) {
#pragma unroll
    for (int c = 0; c < C; c++) atomicAdd(row + c, w * g[c]);
}
template <int C>
__device__ __forceinline__ void scatter_add(__half* row, float w, const float (&g)[C]) {
    if constexpr (C % 2 == 0) {
#pragma unroll
        for (int c = 0; c < C; c += 2) {
            const __half2 v = __halves2half2(__float2half(w * g[c]), __float2half(w * g[c + 1]));
            unsafeAtomicAdd(reinterpret_cast<__half2*>(row + c), v);
        }
    } else {
        // C == 1 with a half table: 16-bit CAS on the containing dword (slow path; the reference warns about it too)
#pragma unroll
        for (int c = 0; c < C; c++) {
            __half* addr = row + c;
            unsigned int* base = reinterpret_cast<unsigned int*>(reinterpret_cast<uintptr_t>(addr) & ~uintptr_t(3));
            const bool hi = (reinterpret_cast<uintptr_t>(addr) & 2) != 0;
            unsigned int old = *base, assumed;
            do {
                assumed = old;
                const unsigned short cur = hi ? (unsigned short)(assumed >> 16) : (unsigned short)(assumed & 0xffffu);
                const __half sum = __float2half(__half2float(__ushort_as_half(cur)) + w * g[c]);
                const unsigned int s = (unsigned int)__half_as_ushort(sum);
                const unsigned int repl = hi ? ((assumed & 0x0000ffffu) | (s << 16)) : ((assumed & 0xffff0000u) | s);
                old = atomicCAS(base, assumed, repl);
            } while (old != assumed);
        }
    }
}

// Backward scatter.  Device-scope float atomics execute memory-side on MI355X (every one is a 32-byte fabric
// transaction) and top out at ~21 G/s whatever the footprint or XCD affinity (tools/atomic_rate.hip), so both
// backward kernels work at issuing few of them.  Lanes of a wavefront hold consecutive samples, i.e. consecutive
// points along a ray: at every level whose cell is larger than the marching step they fall into the same cell in
// runs.  A run's 2^D corner contributions are summed in-wave (segmented suffix sum keyed on the cell coordinates)
// and only the head lane of the run scatters.  Levels with no adjacent sharing (wave-uniform ballot test) skip the
// reduction.
//
// Building blocks shared by the two backward kernels (all 64 lanes of a wave call them together):
//   load_sample     position + upstream gradient of sample b at one level; returns whether it contributes
//   corner_contrib  cell coordinates and the 2^D * C per-corner contributions v[idx*C+c] = w_idx * g[c]
//   aggregate_runs  in-wave run aggregation; returns whether this lane is the head of its run (only heads scatter)
// A sample's position and gradient: requested (every load issued before anything is tested -- tested one coordinate at a
// time, each load waited for the previous one: four memory latencies in a row at the head of both backward kernels) and,
// separately, turned into the kernel's values, so that a persistent workgroup can request its next item's while it works
template <typename T, int D, int C>
struct RawSample {
    float raw[D];
    Feat<T, C> gv;
};
template <typename T, int D, int C>
__device__ __forceinline__ RawSample<T, D, C> request_sample(uint32_t b, const T* __restrict__ grad,
                                                             const float* __restrict__ inputs, uint32_t level, uint32_t B,
                                                             uint32_t L, int grad_layout) {
    RawSample<T, D, C> r;
    const uint32_t bc = b < B ? b : B - 1u;
    const uint32_t Bp = (B + 31u) & ~31u;
#pragma unroll
    for (int d = 0; d < D; d++) r.raw[d] = inputs[(size_t)bc * D + d];
    r.gv = reinterpret_cast<const Feat<T, C>*>(
        grad)[grad_layout == 0 ? (size_t)level * B + bc : grad_layout == 1 ? (size_t)bc * L + level : (size_t)level * Bp + bc];
    return r;
}
template <typename T, int D, int C>
__device__ __forceinline__ bool finish_sample(const RawSample<T, D, C>& r, bool valid, float in_add, float in_mul,
                                              float (&in)[D], float (&g)[C]) {
#pragma unroll
    for (int d = 0; d < D; d++) {
        in[d] = valid ? (r.raw[d] + in_add) * in_mul : 0.0f;
        valid = valid && !(in[d] < 0 || in[d] > 1);   // out-of-range points contribute nothing
    }
#pragma unroll
    for (int c = 0; c < C; c++) g[c] = valid ? to_f(r.gv.v[c]) : 0.0f;
    return valid;
}
template <typename T, int D, int C>
__device__ __forceinline__ bool load_sample(uint32_t b, bool valid, const T* __restrict__ grad, const float* __restrict__ inputs,
                                            uint32_t level, uint32_t B, uint32_t L, int grad_layout, float in_add,
                                            float in_mul, float (&in)[D], float (&g)[C]) {
    const RawSample<T, D, C> r = request_sample<T, D, C>(b, grad, inputs, level, B, L, grad_layout);
    return finish_sample<T, D, C>(r, valid, in_add, in_mul, in, g);
}

template <int D>
__device__ __forceinline__ void cell_of(const float (&in)[D], float scale, uint32_t (&pos_grid)[D], float (&pos)[D]) {
#pragma unroll
    for (int d = 0; d < D; d++) {
        pos[d] = fmaf(in[d], scale, 0.5f);
        pos_grid[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pos_grid[d];
    }
}

template <int D, int C>
__device__ __forceinline__ void corner_contrib(const float (&pos)[D], const float (&g)[C], float (&v)[(1 << D) * C]) {
#pragma unroll
    for (int idx = 0; idx < (1 << D); idx++) {
        float wi = 1;
#pragma unroll
        for (int d = 0; d < D; d++) wi *= ((idx >> d) & 1) ? pos[d] : 1 - pos[d];
#pragma unroll
        for (int c = 0; c < C; c++) v[idx * C + c] = wi * g[c];
    }
}

// lane i reads lane i + O of its own 16-lane row (DPP row_shl; 0.0 where that lane lies beyond the row)
template <int O>
__device__ __forceinline__ float row_down(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x100 | O, 0xf, 0xf, true));
}
// one step of the in-row suffix sums: v[k] += (take ? v[k] of lane + O : 0) -- as v + take * t: a DPP move per value
// and one packed multiply-add per pair of values (v_pk_fma_f32), no select
template <int O, int N>
__device__ __forceinline__ void run_step(float (&v)[N], float takef) {
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = fmaf(takef, row_down<O>(v[k]), v[k]);
}

template <int D, int C>
__device__ __forceinline__ bool aggregate_runs(bool valid, int lane, const uint32_t (&pos_grid)[D],
                                               float (&v)[(1 << D) * C]) {
    // run detection: same cell as the previous lane
    bool same = lane > 0 && valid;
#pragma unroll
    for (int d = 0; d < D; d++) {
        const uint32_t prev = (uint32_t)dpp_take<0x138>(0, (int)pos_grid[d]);      // lane - 1 (wave_shr:1)
        same = same && (prev == pos_grid[d]);
    }
    const bool prev_valid = dpp_take<0x138>(0, (int)valid) != 0;
    same = same && prev_valid;
    const unsigned long long same_mask = __ballot(same);
    bool head = valid;
    if (same_mask != 0ull) {
        head = valid && !same;
        // a run ends right before the next lane that is not a continuation (head or invalid)
        const unsigned long long cont = same_mask >> 1;                 // bit i: lane i+1 continues lane i's run
        const unsigned long long stop = ~cont >> lane;                  // first zero of cont at or above my lane
        const int end = lane + (stop ? __builtin_ctzll(stop) : 64 - lane) + 1;   // one past my run's last lane
        // The run's sum lands in its head lane in two stages, neither of which goes through the LDS crossbar
        // (16 ds_bpermute per doubling step kept the CU's one LDS pipe busy for a third of the binning pass):
        // 1. suffix sums inside each 16-lane row with DPP row shifts: lane i ends up with the sum over
        //    [i, min(run end, row end));
        // 2. a run that continues into the next rows picks up those rows' first lanes (which hold the run's share of
        //    their row), read with v_readlane -- three rows at most.
        constexpr int N = (1 << D) * C;
        if (__ballot(lane + 1 < end) != 0ull) {
            run_step<1, N>(v, lane + 1 < end ? 1.0f : 0.0f);
            if (__ballot(lane + 2 < end) != 0ull) {
                run_step<2, N>(v, lane + 2 < end ? 1.0f : 0.0f);
                if (__ballot(lane + 4 < end) != 0ull) {
                    run_step<4, N>(v, lane + 4 < end ? 1.0f : 0.0f);
                    if (__ballot(lane + 8 < end) != 0ull) run_step<8, N>(v, lane + 8 < end ? 1.0f : 0.0f);
                }
            }
        }
#pragma unroll
        for (int r = 1; r < 4; r++) {
            if ((same_mask >> (16 * r)) & 1ull) {                      // lane 16 r continues a run of the row before
                const float addf = (lane < 16 * r && 16 * r < end) ? 1.0f : 0.0f;
#pragma unroll
                for (int k = 0; k < N; k++) {
                    const float part = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[k]), 16 * r));
                    v[k] = fmaf(addf, part, v[k]);
                }
            }
        }
    }
    return head;
}

// Large fp32 batches take the *binned* backward: pass A (k_grid_bwd_bin) computes each run head's 2^D corner
// contributions once and appends them, as (row-in-tile, values) records, to the record list of the table tile they
// fall in; pass B (k_grid_bwd_tile) gives every list to one workgroup that sums its records into an LDS-resident tile
// (kTileElems / C rows of fp64 accumulators, 128 KiB) and adds the tile into the table with plain coalesced
// read-modify-writes -- a counting sort by tile in place of ~100 memory-side float atomics per sample.  The LDS
// accumulators are fp64 because ds_add_f64 runs at ~0.45 cycles per lane on gfx950 while ds_add_f32 takes ~3.1
// (tools/lds_atomic_rate.hip); the sums are rounded to fp32 once, when the tile is added into the table.  A level smaller than `min_tiles`
// tiles keeps several replica lists per tile (filled by different workgroups, flushed with a few atomics) so that
// its records do not pile up on one CU.  Everything that depends on the table geometry is derived from `offsets`
// on the device.  Levels with more than kMaxBins lists (tables beyond 2^20 rows per level) stay with the atomic kernel.
// 64-KiB tiles (8192 fp64 accumulators), two tile workgroups per CU: against 128-KiB tiles with one workgroup per CU the
// tile kernels gain ~8 % (finer-grained last round, two workgroups' phases interleave) at no cost to the binning pass
#ifndef ENERF_TILE_ELEMS
#define ENERF_TILE_ELEMS 4096
#endif
// how k_grid_bwd_bin ranks a head's records within their lists: 0 = one returning LDS atomic per record (shipped), 1 = one
// per corner pair / cell when the records share a list, 2 = 1 + one per wavefront and list on the coarse levels (ballot).
// Both alternatives were measured SLOWER (profiles/r06_bin_ranks_rejected.txt) and are kept for the record only.
#ifndef ENERF_BIN_RANKS
#define ENERF_BIN_RANKS 0
#endif
#ifndef ENERF_TA_PAIRS
#define ENERF_TA_PAIRS 10
#endif
#ifdef ENERF_TA_CAP
#define ENERF_TA_REGS __attribute__((amdgpu_waves_per_eu(5, 8)))
#else
#define ENERF_TA_REGS __attribute__((amdgpu_waves_per_eu(4, 8)))
#endif
struct __attribute__((aligned(8))) f32x2g { float x, y; };
// k_grid_tile_adam's tiles are handed out on request: [0] the next tile nobody has taken, [1] workgroups that have left -- the
// last one to leave clears [0] for the next launch ([1] wraps by itself); launches follow one another on one stream
__device__ uint32_t g_ta_next[2];
// k_grid_tile_adam's m / v stream is marked non-temporal (every element is touched once per step; p is read again by the next
// forward and keeps its place in the caches): the table's backward + Adam 120 -> 111 us stand-alone, 1 - 2 us in the step.
// (Stores written through -- sc0 sc1 -- so that no dirty line waits for the end of the kernel: no change, 85.0 vs 85.7 us.)
// (-DENERF_TA_TEMPORAL: plain loads and stores)
typedef float ta_f4 __attribute__((ext_vector_type(4)));
#ifndef ENERF_TA_TEMPORAL
__device__ __forceinline__ float4 ta_load4(const float* p) {
    const ta_f4 v = __builtin_nontemporal_load(reinterpret_cast<const ta_f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void ta_store4(float* p, const float4& v) {
    ta_f4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<ta_f4*>(p));
}
#define TA_LOAD4(p) ta_load4(p)
#define TA_STORE4(p, v) ta_store4(p, v)
#else
#define TA_LOAD4(p) (*reinterpret_cast<const float4*>(p))
#define TA_STORE4(p, v) (*reinterpret_cast<float4*>(p) = (v))
#endif
constexpr uint32_t kTileElems = ENERF_TILE_ELEMS;
constexpr uint32_t kTileThreads = ENERF_TILE_ELEMS / 16;
constexpr uint32_t kTileAccBytes = ENERF_TILE_ELEMS * 8;         // k_grid_tile_adam's fp64 accumulators (dynamic LDS)
constexpr uint32_t kTilesPerCu = 16384 / ENERF_TILE_ELEMS;     // resident tile workgroups per CU
// record lists per level: a 2^19-row level cut into tiles of R rows, between 256 (C = 2: the binning pass's three LDS tables
// of that many words leave room for 3 workgroups per CU) and kMaxBins (the stride of the cursor array)
constexpr uint32_t kMaxBins = 512;
__host__ __device__ constexpr uint32_t bins_limit(uint32_t R) {
    return (1u << 19) / R <= 256u ? 256u : ((1u << 19) / R < kMaxBins ? (1u << 19) / R : kMaxBins);
}
struct BinPlan {
    uint32_t tiles, replicas, bins;          // bins = tiles * replicas (0: level not binned)
};
__device__ __forceinline__ BinPlan bin_plan(const int32_t* __restrict__ offsets, uint32_t level, uint32_t rows_per_tile,
                                            uint32_t min_tiles) {
    BinPlan p;
    p.tiles = div_up((uint32_t)(offsets[level + 1] - offsets[level]), rows_per_tile);
    p.replicas = p.tiles >= min_tiles ? 1u : div_up(min_tiles, p.tiles);
    p.bins = p.tiles * p.replicas <= bins_limit(rows_per_tile) ? p.tiles * p.replicas : 0u;
    return p;
}

// records a list can hold: the level's region cut evenly, kept even so that two neighbouring records (one 4-byte pair of
// keys, one 8-byte pair per value plane) can be fetched with one load each (k_grid_tile_adam)
__device__ __forceinline__ uint32_t bin_cap(uint32_t region, uint32_t bins) { return (region / bins) & ~1u; }

// One thread per (sample, level), global atomics from run heads.  binned_min_tiles != 0: skip the binned levels.
template <typename T, int D, int C>
__global__ void __launch_bounds__(kPtsPerBlock) k_grid_bwd(const T* __restrict__ grad, const float* __restrict__ inputs,
                                                           const int32_t* __restrict__ offsets, T* __restrict__ grad_grid,
                                                           uint32_t B, uint32_t L, LevelTab tab, uint32_t gridtype,
                                                           int grad_layout, uint32_t nchunks, uint32_t binned_min_tiles) {
    uint32_t level, chunk;
    if (!decode_block(nchunks, L, level, chunk)) return;
    if (!level_enabled(tab, level)) return;
    if (binned_min_tiles != 0 && bin_plan(offsets, level, kTileElems / C, binned_min_tiles).bins != 0) return;
    const uint32_t b = chunk * kPtsPerBlock + threadIdx.x;
    const int lane = lane_id();

    const uint32_t off0 = (uint32_t)offsets[level];
    const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off0;
    const LevelGeom<D> geom = make_geom<D>(gridtype, hashmap_size, tab.resolution[level]);
    T* rows = grad_grid + (size_t)off0 * C;

    float in[D], g[C], pos[D], v[(1 << D) * C];
    uint32_t pos_grid[D];
    const bool valid = load_sample<T, D, C>(b, b < grid_row_limit(tab, B), grad, inputs, level, B, L, grad_layout, tab.in_add, tab.in_mul, in, g);
    cell_of<D>(in, tab.scale[level], pos_grid, pos);
    corner_contrib<D, C>(pos, g, v);
    if (!aggregate_runs<D, C>(valid, lane, pos_grid, v)) return;

    uint32_t cr[1 << D];
    corner_rows<D>(geom, pos_grid, cr);
#pragma unroll
    for (int idx = 0; idx < (1 << D); idx++) {
        float gg[C];
#pragma unroll
        for (int c = 0; c < C; c++) gg[c] = v[idx * C + c];
        scatter_add<C>(rows + (size_t)cr[idx] * C, 1.0f, gg);
    }
}

// Pass A of the binned path: 256 samples of one level per workgroup.  Level l owns `region` records of `recs`, cut
// evenly between its lists (capacity = region / bins, at least twice the expected load); a list stores its records as
// 1 + C arrays of `capacity` dwords (key = row within the tile, then the C values); `cursors[l][list]` counts the
// records appended.  The workgroup ranks its records per list in LDS, stages them sorted by list, reserves a range
// in every list it feeds with one integer atomic, and copies the staged records out in runs of consecutive slots.
// A record that does not fit (pathologically skewed input) is added to the table with float atomics right here.
// PTS points per workgroup: the longer a workgroup's run in each list, the fewer partial cache lines it writes
template <int D, int C, int PTS>
__global__ void __launch_bounds__(PTS) k_grid_bwd_bin(const float* __restrict__ grad,
                                                               const float* __restrict__ inputs,
                                                               const int32_t* __restrict__ offsets,
                                                               float* __restrict__ grad_grid, uint32_t B, uint32_t L,
                                                               LevelTab tab, uint32_t gridtype, int grad_layout,
                                                               uint32_t nchunks, uint32_t min_tiles,
                                                               uint32_t* __restrict__ recs, uint32_t* __restrict__ cursors,
                                                               uint32_t region, uint32_t* __restrict__ overflow) {
    constexpr uint32_t R = kTileElems / C;
    constexpr uint32_t NREC = PTS << D;
    constexpr uint32_t kBins = bins_limit(kTileElems / C);
    __shared__ uint32_t s_ofs[kBins];            // records per list, then exclusive offset of the list in the staging area
    // where the staging area's record j of list t goes, in words from the level's first record: s_at[t] + j, valid
    // below s_end[t] (= the list's first word + its capacity); both are made once per list, so that the copy-out does
    // no multiplication and no 64-bit arithmetic per record
    __shared__ uint32_t s_at[kBins];
    __shared__ uint32_t s_end[kBins];
    __shared__ uint32_t s_key[NREC];             // row within tile | list << 16
    __shared__ float s_val[C][NREC];
    __shared__ uint32_t s_wave[PTS / 64 + 1];
    TA_MARK_IN(2);
    uint32_t level, chunk;
    if (!decode_block(nchunks, L, level, chunk)) return;
    if (!level_enabled(tab, level)) return;
    const BinPlan plan = bin_plan(offsets, level, R, min_tiles);
    const uint32_t b = chunk * PTS + threadIdx.x;
    const int lane = lane_id();
    if (plan.bins == 0) {
        // a level too small to bin (less than one tile): scatter it with atomics right here (block-uniform branch)
        // instead of launching the atomic kernel for it
        const uint32_t off0 = (uint32_t)offsets[level];
        const LevelGeom<D> geom = make_geom<D>(gridtype, (uint32_t)offsets[level + 1] - off0, tab.resolution[level]);
        float* rows = grad_grid + (size_t)off0 * C;
        float in[D], g[C], pos[D], v[(1 << D) * C];
        uint32_t pos_grid[D], cr[1 << D];
        const bool valid = load_sample<float, D, C>(b, b < grid_row_limit(tab, B), grad, inputs, level, B, L, grad_layout, tab.in_add,
                                                    tab.in_mul, in, g);
        cell_of<D>(in, tab.scale[level], pos_grid, pos);
        corner_contrib<D, C>(pos, g, v);
        if (!aggregate_runs<D, C>(valid, lane, pos_grid, v)) return;
        corner_rows<D>(geom, pos_grid, cr);
#pragma unroll
        for (int idx = 0; idx < (1 << D); idx++) {
            float gg[C];
#pragma unroll
            for (int c = 0; c < C; c++) gg[c] = v[idx * C + c];
            scatter_add<C>(rows + (size_t)cr[idx] * C, 1.0f, gg);
        }
        return;
    }
    const uint32_t cap = bin_cap(region, plan.bins);
    const uint32_t replica = chunk % plan.replicas;
#ifdef ENERF_TA_TIMING
    unsigned long long bin_t = __builtin_amdgcn_s_memrealtime();
    const uint32_t bin_lv = level;
    if (threadIdx.x == 0) atomicAdd(&g_bin_ph[bin_lv * 8 + 7], 1ull);
#endif

    for (uint32_t t = threadIdx.x; t < plan.bins; t += PTS) s_ofs[t] = 0;
    __syncthreads();
    BIN_PH(0);

    const uint32_t off0 = (uint32_t)offsets[level];
    const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off0;
    const LevelGeom<D> geom = make_geom<D>(gridtype, hashmap_size, tab.resolution[level]);

    float in[D], g[C], pos[D], v[(1 << D) * C];
    uint32_t pos_grid[D], cr[1 << D], bin[1 << D], rank[1 << D];
    const bool valid = load_sample<float, D, C>(b, b < grid_row_limit(tab, B), grad, inputs, level, B, L, grad_layout, tab.in_add, tab.in_mul, in, g);
    cell_of<D>(in, tab.scale[level], pos_grid, pos);
    corner_contrib<D, C>(pos, g, v);
    const bool head = aggregate_runs<D, C>(valid, lane, pos_grid, v);
    corner_rows<D>(geom, pos_grid, cr);
    // Ranks of the records within their lists.  A returning LDS atomic per record (eight per run head) was 2-4 of this
    // workgroup's ~7 us; records that share a list are ranked together instead:
    //  * the two x-neighbours of a corner pair sit in adjacent rows (dense levels; hashed levels with x even) or in rows that
    //    differ in a few low bits (x odd): one list almost always -> one atomic of 2;
    //  * on the coarse levels all eight corners of a cell -- and those of every head of the wavefront -- fall into one or
    //    two lists: the wavefront's heads are counted with a ballot and ONE lane reserves for all of them (what serialised
    //    64 same-address atomics per instruction before).
    // The order of a list's records changes with this, nothing else: k_grid_tile_adam sums them in fp64 and rounds once.
#if ENERF_BIN_RANKS >= 1
    bool all8 = head, ranked = false;
#pragma unroll
    for (int idx = 0; idx < (1 << D); idx++) {
        bin[idx] = __umul24(cr[idx] / R, plan.replicas) + replica;          // (R is a power of two; both factors < 2^24)
        rank[idx] = 0;
        all8 = all8 && bin[idx] == bin[0];
    }
    {
        unsigned long long m = ENERF_BIN_RANKS >= 2 ? __ballot(all8) : 0ull;
        if (__popcll(m) >= 8) {                                              // (wave-uniform)
            for (int round = 0; round < 4 && m != 0ull; round++) {
                const int leader = __ffsll((long long)m) - 1;
                const uint32_t bl = (uint32_t)__builtin_amdgcn_readlane((int)bin[0], leader);
                const bool in = all8 && !ranked && bin[0] == bl;
                const unsigned long long mm = __ballot(in);
                uint32_t base = 0;
                if (lane == leader) base = atomicAdd(&s_ofs[bl], (uint32_t)__popcll(mm) << D);
                base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
                if (in) {
                    const uint32_t before_me = __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32),
                                                                         __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
#pragma unroll
                    for (int idx = 0; idx < (1 << D); idx++) rank[idx] = base + (before_me << D) + (uint32_t)idx;
                    ranked = true;
                }
                m &= ~mm;
            }
        }
    }
    if (head && all8 && !ranked) {                                           // one list, not ranked by the wavefront
        const uint32_t r = atomicAdd(&s_ofs[bin[0]], 1u << D);
#pragma unroll
        for (int idx = 0; idx < (1 << D); idx++) rank[idx] = r + (uint32_t)idx;
    } else if (head && !ranked) {
#pragma unroll
        for (int k = 0; k < (1 << (D - 1)); k++) {
            if (bin[2 * k] == bin[2 * k + 1]) {
                const uint32_t r = atomicAdd(&s_ofs[bin[2 * k]], 2u);
                rank[2 * k] = r;
                rank[2 * k + 1] = r + 1u;
            } else {
                rank[2 * k] = atomicAdd(&s_ofs[bin[2 * k]], 1u);
                rank[2 * k + 1] = atomicAdd(&s_ofs[bin[2 * k + 1]], 1u);
            }
        }
    }
#else
    if (head) {
#pragma unroll
        for (int idx = 0; idx < (1 << D); idx++) {
            bin[idx] = __umul24(cr[idx] / R, plan.replicas) + replica;      // (R is a power of two; both factors < 2^24)
            rank[idx] = atomicAdd(&s_ofs[bin[idx]], 1u);
        }
    }
#endif
    __syncthreads();
    BIN_PH(1);

    // exclusive scan of the per-list counts (thread t owns `per` consecutive lists) + global reservation
    const uint32_t per = div_up(plan.bins, PTS);
    uint32_t mine = 0;
    for (uint32_t k = 0; k < per; k++) {
        const uint32_t t = threadIdx.x * per + k;
        if (t < plan.bins) mine += s_ofs[t];
    }
    const uint32_t incl = wave_incl_scan_add_u32(mine, lane);
    if (lane == 63) s_wave[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t before = incl - mine;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) before += s_wave[w];
    if (threadIdx.x == PTS - 1) s_wave[PTS / 64] = before + mine;      // total records
    // the reservation in the global lists (one returning atomic per list fed) is requested here and its answer is used
    // only after the records have been staged: the round trip to the cursor runs beside the barrier and the staging
    constexpr uint32_t PER_MAX = (kBins + PTS - 1) / PTS;
    uint32_t got_[PER_MAX], bef_[PER_MAX];
#pragma unroll
    for (uint32_t k = 0; k < PER_MAX; k++) {
        const uint32_t t = threadIdx.x * per + k;
        got_[k] = 0; bef_[k] = 0;
        if (k < per && t < plan.bins) {
            const uint32_t n = s_ofs[t];
            s_ofs[t] = before;
            bef_[k] = before;
            got_[k] = n ? atomicAdd(&cursors[level * kMaxBins + t], n) : 0u;
            before += n;
        }
    }
    __syncthreads();
    BIN_PH(2);

    if (head) {
#pragma unroll
        for (int idx = 0; idx < (1 << D); idx++) {
            const uint32_t p = s_ofs[bin[idx]] + rank[idx];
            s_key[p] = (cr[idx] & (R - 1u)) | (bin[idx] << 16);
#pragma unroll
            for (int c = 0; c < C; c++) s_val[c][p] = v[idx * C + c];
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < PER_MAX; k++) {
        const uint32_t t = threadIdx.x * per + k;
        if (k < per && t < plan.bins) {
            const uint32_t first = t * cap * (1u + C);        // (< 2^32 words: a level's region is far smaller)
            s_at[t] = first + got_[k] - bef_[k];
            s_end[t] = first + cap;
        }
    }
    __syncthreads();
    BIN_PH(3);

    const uint32_t total = s_wave[PTS / 64];
    uint32_t* lrecs = recs + (size_t)level * region * (1 + C);
    for (uint32_t j = threadIdx.x; j < total; j += PTS) {
        const uint32_t key = s_key[j], list = key >> 16, loc = key & 0xffffu;
        const uint32_t w = s_at[list] + j, e = s_end[list];   // word of the record's slot in plane 0; the list's limit
        if (w < e) {
            // record = 16-bit row within the tile (first half of the list's index plane) + C value planes:
            // list word f = e - cap, slot = w - f -> index plane entry 2 f + slot = f + w, value plane c at w + (1 + c) cap
            reinterpret_cast<uint16_t*>(lrecs)[(e - cap) + w] = (uint16_t)loc;
#pragma unroll
            for (int c = 0; c < C; c++) lrecs[w + (1u + c) * cap] = __float_as_uint(s_val[c][j]);
        } else {
            float gg[C];
#pragma unroll
            for (int c = 0; c < C; c++) gg[c] = s_val[c][j];
            scatter_add<C>(grad_grid + ((size_t)off0 + (size_t)(list / plan.replicas) * R + loc) * C, 1.0f, gg);
            if (overflow) atomicAdd(overflow, 1u);          // (a deferred flush must then also read the dense gradient)
        }
    }
#ifdef ENERF_TA_TIMING
    __syncthreads();
#endif
    BIN_PH(4);
    TA_MARK_OUT(3);
}

// Pass B: persistent workgroups (one per CU: the tile takes 128 KiB of LDS) walk the record lists, finest level
// first.  Sum the list's records into the LDS tile, add the non-zero part into the table (plain read-modify-write
// when the tile has a single list, float atomics for replica lists), reset the list's cursor for the next call.
// Owner range (data parallel, sharded tail; enerf_grid_owner_range): elements [own_lo, own_hi) of the flat table belong to
// this rank.  Lists of tiles that lie wholly inside it are left alone -- they wait for k_grid_tile_adam, which sums them
// in LDS as on one GPU -- and only the others are flushed into the dense gradient, for the reduce-scatter to carry away.
template <int C>
__global__ void __launch_bounds__(kTileThreads) k_grid_bwd_tile(const int32_t* __restrict__ offsets,
                                                                float* __restrict__ grad_grid, uint32_t L, LevelTab tab,
                                                                uint32_t min_tiles, const uint32_t* __restrict__ recs,
                                                                uint32_t* __restrict__ cursors, uint32_t region,
                                                                size_t own_lo = 0, size_t own_hi = 0) {
    __shared__ __attribute__((aligned(16))) double acc[kTileElems];
    __shared__ uint32_t s_n;
    constexpr uint32_t R = kTileElems / C;
    constexpr int U = 4;                         // records in flight per thread
    uint32_t total = 0;
    for (uint32_t lv = 0; lv < L; lv++) total += bin_plan(offsets, lv, R, min_tiles).bins;
    for (uint32_t item = blockIdx.x; item < total; item += gridDim.x) {
        uint32_t level = 0, list = 0, rem = item;
        BinPlan plan;
        for (uint32_t lv = L; lv-- > 0;) {
            plan = bin_plan(offsets, lv, R, min_tiles);
            if (rem < plan.bins) {
                level = lv;
                list = rem;
                break;
            }
            rem -= plan.bins;
        }
        const uint32_t cap = bin_cap(region, plan.bins);
        if (own_hi > own_lo) {
            const uint32_t o0 = (uint32_t)offsets[level], hs = (uint32_t)offsets[level + 1] - o0;
            const uint32_t r0 = (list / plan.replicas) * R, nr = hs - r0 < R ? hs - r0 : R;
            const size_t t_lo = ((size_t)o0 + r0) * C, t_hi = t_lo + (size_t)nr * C;
            if (t_lo >= own_lo && t_hi <= own_hi) continue;             // mine: the optimizer pass consumes the list
        }
        if (threadIdx.x == 0) {
            const uint32_t n = cursors[level * kMaxBins + list];
            s_n = n < cap ? n : cap;
            cursors[level * kMaxBins + list] = 0;
        }
        __syncthreads();
        const uint32_t n = s_n;
        if (n != 0 && level_enabled(tab, level)) {
            const uint32_t off0 = (uint32_t)offsets[level];
            const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off0;
            const uint32_t row0 = (list / plan.replicas) * R;
            const uint32_t nrows = hashmap_size - row0 < R ? hashmap_size - row0 : R;
            for (uint32_t i = threadIdx.x * 2; i < nrows * C; i += kTileThreads * 2)
                *reinterpret_cast<double2*>(acc + i) = make_double2(0.0, 0.0);
            __syncthreads();
            const uint32_t* r = recs + ((size_t)level * region + (size_t)list * cap) * (1 + C);
            for (uint32_t i0 = threadIdx.x; i0 < n; i0 += kTileThreads * U) {
                uint32_t loc[U];
                float val[U][C];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint32_t i = i0 + u * kTileThreads;
                    const uint32_t ic = i < n ? i : n - 1;
                    loc[u] = reinterpret_cast<const uint16_t*>(r)[ic];
#pragma unroll
                    for (int c = 0; c < C; c++) val[u][c] = __uint_as_float(r[(size_t)(1 + c) * cap + ic]);
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    if (i0 + u * kTileThreads < n) {
#pragma unroll
                        for (int c = 0; c < C; c++) atomicAdd(acc + loc[u] * C + c, (double)val[u][c]);
                    }
                }
            }
            __syncthreads();
            float* dst = grad_grid + ((size_t)off0 + row0) * C;     // level offsets are multiples of 8 rows: 16-byte aligned
            if (plan.replicas == 1) {
                for (uint32_t i = threadIdx.x * 4; i < nrows * C; i += kTileThreads * 4) {
                    const double2 a0 = *reinterpret_cast<const double2*>(acc + i);
                    const double2 a1 = *reinterpret_cast<const double2*>(acc + i + 2);
                    if (a0.x != 0.0 || a0.y != 0.0 || a1.x != 0.0 || a1.y != 0.0) {
                        float4 o = *reinterpret_cast<float4*>(dst + i);
                        o.x += (float)a0.x; o.y += (float)a0.y; o.z += (float)a1.x; o.w += (float)a1.y;
                        *reinterpret_cast<float4*>(dst + i) = o;
                    }
                }
            } else {
                for (uint32_t i = threadIdx.x; i < nrows * C; i += kTileThreads) {
                    const double a = acc[i];
                    if (a != 0.0) atomicAdd(dst + i, (float)a);
                }
            }
        }
        __syncthreads();
    }
}

// ---- deferred flush: pass B fused with the optimizer --------------------------------------------------------------
// The table's gradient is consumed exactly once, by Adam (main_nerf.py:211), which streams over the whole table anyway
// (p, m, v: 24 B per element).  k_grid_tile_adam therefore takes over from pass B when the caller defers the flush
// (enerf_grid_encode_backward_ex, flag bit 0): every 128-KiB tile of every level is visited once; the tile's record
// lists are summed into the fp64 LDS accumulators as in k_grid_bwd_tile and the Adam update of the tile's rows reads its
// gradient straight from LDS (measured: the streaming part alone runs at the dense Adam kernel's 56 us, the LDS sums add
// ~45 us -- a split of the workgroup into accumulating and streaming waves did not overlap them).  The dense gradient table is never written, read or cleared for the binned levels
// (3 x 52 MB of traffic and pass B's read-modify-write of the table go away), and the LDS atomics run in the shadow
// of the p / m / v stream.  Levels that are not binned (their contributions went into the dense gradient with
// atomics) and runs in which a record list overflowed read -- and clear -- the dense gradient as well.
// Update arithmetic: optim.hip's adam1, term for term.
struct AdamScalars {
    float b1, b2, eps, step_size, inv_bc2_sqrt;
};
// (the next batch's count pass rides in this kernel's launch: common.h MarchCountJob)
#include "march_lattice.h"

// the model's other (small) parameters ride along in the same launch: one workgroup each, after its tiles
constexpr int kMaxSmallAdam = 8;
struct SmallAdam {
    float* p[kMaxSmallAdam];
    const float* g[kMaxSmallAdam];
    float* m[kMaxSmallAdam];
    float* v[kMaxSmallAdam];
    uint32_t n[kMaxSmallAdam];
    float step_size[kMaxSmallAdam], inv_bc2_sqrt[kMaxSmallAdam];
    uint32_t count;
};
__device__ __forceinline__ void tile_adam1(float& p, float g, float& m, float& v, const AdamScalars& a) {
    m = fmaf(g - m, 1.0f - a.b1, m);
    v = fmaf((1.0f - a.b2) * g, g, v * a.b2);
    const float denom = sqrtf(v) * a.inv_bc2_sqrt + a.eps;
    p = p - a.step_size * (m / denom);
}

// AMP (enerf_amp_begin, the fp16 regime's loss scaling): the gradients arrive multiplied by *scale and are divided by it
// here (GradScaler.unscale_: grad * (1 / scale)); a step whose weight gradients were not finite (*found_inf, raised by the
// MLP reduce launch earlier in the stream) leaves p / m / v as they are -- lists and dense gradient are still consumed and
// cleared -- and does not count: the bias corrections use (host step - *skipped), evaluated here.
struct AmpAdam {
    const float* scale;
    const uint32_t* found_inf;
    const uint32_t* skipped;
    float lr;
    uint32_t step;
    float small_lr[kMaxSmallAdam];
    uint32_t small_step[kMaxSmallAdam];
};
__device__ __forceinline__ void amp_scalars(AdamScalars& a, float lr, uint32_t step, uint32_t skipped) {
    const double t = (double)(step > skipped ? step - skipped : 1u);
    a.step_size = (float)((double)lr / (1.0 - pow((double)a.b1, t)));
    a.inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - pow((double)a.b2, t)));
}

// RANGE (enerf_grid_owner_range, the sharded data-parallel tail): Adam only on the elements [own_lo, own_hi) of the flat
// table.  The dense gradient holds what the reduce-scatter (SUM) delivered: the other ranks' contributions, plus this
// rank's own spilled / unbinned ones; a tile wholly inside the range adds the sum of its own record lists -- the one-GPU
// flush for this rank's slice -- and the total is multiplied by rec_scale (1 / ranks: the average over the ranks); a tile
// the range cuts takes its gradient from the dense buffer alone (pass B flushed its lists); everything outside is only
// cleared.
struct OwnerRange {
    size_t lo, hi;
    float rec_scale;
};

// (registers: held to a fifth of a SIMD's file, so that the four workgroups a CU's LDS admits stay resident beside one
//  wavefront per SIMD of the next batch's march, which runs on a second stream under this kernel)
template <int C, bool AMP = false, bool RANGE = false, bool COUNT = false>
__global__ void __launch_bounds__(kTileThreads) ENERF_TA_REGS k_grid_tile_adam(const int32_t* __restrict__ offsets, float* __restrict__ P,
                                                                 float* __restrict__ G, float* __restrict__ M,
                                                                 float* __restrict__ V, uint32_t L, uint32_t min_tiles,
                                                                 const uint32_t* __restrict__ recs,
                                                                 uint32_t* __restrict__ cursors, uint32_t region,
                                                                 uint32_t* __restrict__ overflow,
                                                                 uint32_t* __restrict__ other_overflow, AdamScalars ad,
                                                                 SmallAdam small, AmpAdam amp = AmpAdam{},
                                                                 OwnerRange own = OwnerRange{0, 0, 1.0f},
                                                                 PartialSums ps = PartialSums{nullptr, nullptr, 0, 0, 0},
                                                                 MarchCountJob cj = MarchCountJob{},
                                                                 MarchCountJob cj2 = MarchCountJob{}) {
    // (dynamic: a static 32 KiB array tells the compiler that four workgroups fill the CU, and it then spends the registers
    //  of a fifth wavefront per SIMD on scheduling freedom -- see the note above the kernel)
    extern __shared__ __attribute__((aligned(16))) double acc[];
    __shared__ uint32_t s_n[64];
    // COUNT: the launch's first cj.blocks workgroups are the next batch's march count pass (march_lattice.h; its two
    // lookup tables live in the accumulators' LDS); the rest see themselves as workgroup `bid` of `nb`
    uint32_t bid = blockIdx.x, nb = gridDim.x;
    if constexpr (COUNT) {
        static_assert(kTileThreads == 256 && kTileAccBytes >= (2 * kTabH + 1) * 4, "the count pass runs in 256-thread workgroups");
        const uint32_t ncount = cj.blocks + cj2.blocks;      // (the event step's two renders: two jobs, one after the other)
        if (blockIdx.x < ncount) {
            float* s_face = reinterpret_cast<float*>(acc);
            uint32_t* s_expand = reinterpret_cast<uint32_t*>(s_face + kTabH + 1);
            if (blockIdx.x < cj.blocks) march_count_block(cj, blockIdx.x, cj.blocks, s_face, s_expand);
            else march_count_block(cj2, blockIdx.x - cj.blocks, cj2.blocks, s_face, s_expand);
            return;
        }
        bid -= ncount;
        nb -= ncount;
    }
#ifdef ENERF_TA_TIMING
    if (threadIdx.x == 0 && bid < 2048) g_ta_wg[2 * bid] = (uint32_t)__builtin_amdgcn_s_memrealtime();
#endif
    TA_MARK_IN(4);
    constexpr uint32_t R = kTileElems / C;
    // record pairs per thread and round (the AMP / RANGE forms hold more state: fewer, so that they fit the register cap)
    constexpr int UP0 = (AMP || RANGE) ? ENERF_TA_PAIRS - 2 : ENERF_TA_PAIRS;
    constexpr int UP = C <= 2 ? UP0 : (2 * UP0 / C > 0 ? 2 * UP0 / C : 1);
    float inv_scale = 1.0f;
    bool skip = false;
    uint32_t skipped = 0;
    if (AMP) {
        inv_scale = (float)(1.0 / (double)amp.scale[0]);
        skip = amp.found_inf[0] != 0u;
        skipped = amp.skipped[0];
        amp_scalars(ad, amp.lr, amp.step, skipped);
    }
    const bool have_records = region != 0;
    const bool spilled = have_records && overflow[0] != 0;
    if (other_overflow && bid == 0 && threadIdx.x == 0) other_overflow[0] = 0;     // the next session's counter
    uint32_t total = 0;
    for (uint32_t lv = 0; lv < L; lv++) total += div_up((uint32_t)(offsets[lv + 1] - offsets[lv]), R);
    auto decode = [&](uint32_t item, uint32_t& level, uint32_t& tile) {
        uint32_t rem = item;
        level = 0; tile = 0;
        // coarsest levels first: their few tiles carry the longest lists (rays crowd into the same coarse cells: 41 k
        // records on one tile of level 0 against 4 k on a tile of a hashed level, 33 us against 14) and must not be the
        // kernel's tail; the hashed levels' tiles, all alike, fill the last round evenly
        for (uint32_t lv = 0; lv < L; lv++) {
            const uint32_t t = div_up((uint32_t)(offsets[lv + 1] - offsets[lv]), R);
            if (rem < t) { level = lv; tile = rem; break; }
            rem -= t;
        }
    };
    // the record count of a tile's (first) list is requested one tile ahead: a tile's chain of dependent memory
    // latencies is then records -> p / m / v, not cursor -> records -> p / m / v.  (Only this workgroup reads or
    // resets the cursors of its tiles.)
    auto first_cursor = [&](uint32_t item) -> uint32_t {
        if (!have_records || item >= total) return 0u;
        uint32_t lv, tl;
        decode(item, lv, tl);
        const BinPlan pl = bin_plan(offsets, lv, R, min_tiles);
        return pl.bins != 0 ? cursors[lv * kMaxBins + tl * pl.replicas] : 0u;
    };
    // Tiles on request: a workgroup's first tile is its own number, every further one the next nobody has taken (one
    // atomic per tile, asked for at the tile's start and read at its end).  Dealt out in fixed strides, 107 of the 1024
    // workgroups had a fourth tile of the 3179 and the kernel's last ~15 us ran at a tenth of the chip.
    __shared__ uint32_t s_next;
    uint32_t* next_counter = g_ta_next;
    uint32_t item = bid;
    uint32_t n_ahead = first_cursor(item);
    while (item < total) {
        uint32_t pulled = 0;
        if (threadIdx.x == 0) pulled = nb + atomicAdd(next_counter, 1u);
        uint32_t level, tile;
        decode(item, level, tile);
        uint32_t n_first = n_ahead;
#ifdef ENERF_TA_TIMING
        const uint32_t ta_t0 = (uint32_t)__builtin_amdgcn_s_memrealtime();
        uint32_t ta_n = 0;
#endif
        const uint32_t off0 = (uint32_t)offsets[level];
        const uint32_t rows = (uint32_t)offsets[level + 1] - off0;
        const uint32_t row0 = tile * R;
        const uint32_t nrows = rows - row0 < R ? rows - row0 : R;
        const BinPlan plan = bin_plan(offsets, level, R, min_tiles);
        const size_t base = ((size_t)off0 + row0) * C;          // level offsets are multiples of 8 rows: 16-byte aligned
        uint32_t clip_lo = 0, clip_hi = nrows * C;              // RANGE: the tile's elements this rank updates
        bool whole = true;
        if (RANGE) {
            const size_t t_hi = base + (size_t)nrows * C;
            if (t_hi <= own.lo || base >= own.hi) {             // not mine: the contributions held here are spent
                for (uint32_t i = threadIdx.x * 4; i < nrows * C; i += kTileThreads * 4)
                    *reinterpret_cast<float4*>(G + base + i) = make_float4(0.f, 0.f, 0.f, 0.f);
                if (threadIdx.x == 0) s_next = pulled;
                __syncthreads();
                item = s_next;
                __syncthreads();
                n_ahead = first_cursor(item);
                continue;
            }
            whole = base >= own.lo && t_hi <= own.hi;
            if (!whole) {
                clip_lo = own.lo > base ? (uint32_t)(own.lo - base) : 0u;
                clip_hi = own.hi < t_hi ? (uint32_t)(own.hi - base) : nrows * C;
            }
        }
        const bool binned = have_records && plan.bins != 0 && whole;
        const bool dense = RANGE || !binned || spilled;
        if (binned) {
            const uint32_t cap = bin_cap(region, plan.bins);
            // Every thread holds the first list's count itself (n_first: a wave-uniform load issued one tile ago) instead of
            // waiting for a load -> LDS -> barrier -> LDS round trip; replica lists (levels of few tiles) keep that
            // route for the other counts.  Thread 0 resets the cursors after the barrier.
            const uint32_t li0 = level * kMaxBins + tile * plan.replicas;
            for (uint32_t i = threadIdx.x * 2; i < nrows * C; i += kTileThreads * 2)
                *reinterpret_cast<double2*>(acc + i) = make_double2(0.0, 0.0);
            if (plan.replicas > 1 && threadIdx.x < plan.replicas && threadIdx.x < 64) {
                const uint32_t n = cursors[li0 + threadIdx.x];
                s_n[threadIdx.x] = n < cap ? n : cap;
            }
            n_first = n_first < cap ? n_first : cap;
            asm volatile("" :: "v"(n_first));       // (arrived: a wave's loads return in order, and the previous tile waited on younger ones)
            __syncthreads();
            if (threadIdx.x < plan.replicas && threadIdx.x < 64) cursors[li0 + threadIdx.x] = 0;
            for (uint32_t rep = 0; rep < plan.replicas; rep++) {
                const uint32_t n = plan.replicas == 1 ? n_first : s_n[rep < 64 ? rep : 63];
#ifdef ENERF_TA_TIMING
                ta_n += n;
#endif
                const uint32_t list = tile * plan.replicas + rep;
                const uint32_t* r = recs + ((size_t)level * region + (size_t)list * cap) * (1 + C);
                if (n == 0) continue;
                // A round = UP record pairs per thread, ALL requested before the first is added: the list of a tile of a
                // hashed level (~8.3 k records at the 4096-ray batch) is one round, i.e. one memory latency; the round
                // loop only turns for longer lists.  (Before: 2048 records per round with the next round requested one
                // round ahead -- five dependent latencies for the same list.)
                const uint32_t npairs = (n + 1u) >> 1;
                for (uint32_t p0 = threadIdx.x; p0 < npairs; p0 += kTileThreads * UP) {
                    uint32_t key2[UP];
                    f32x2g val2[UP][C];
#pragma unroll
                    for (int u = 0; u < UP; u++) {
                        const uint32_t pi = p0 + u * kTileThreads;
                        const uint32_t pc = pi < npairs ? pi : npairs - 1;
                        key2[u] = r[pc];                                           // two 16-bit rows
#pragma unroll
                        for (int c = 0; c < C; c++)
                            val2[u][c] = *reinterpret_cast<const f32x2g*>(r + (size_t)(1 + c) * cap + 2 * pc);
                    }
#pragma unroll
                    for (int u = 0; u < UP; u++) {
                        const uint32_t pi = p0 + u * kTileThreads;
                        if (pi < npairs) {
                            const uint32_t la = key2[u] & 0xffffu, lb = key2[u] >> 16;
#pragma unroll
                            for (int c = 0; c < C; c++) atomicAdd(acc + la * C + c, (double)val2[u][c].x);
                            if (2 * pi + 1 < n) {
#pragma unroll
                                for (int c = 0; c < C; c++) atomicAdd(acc + lb * C + c, (double)val2[u][c].y);
                            }
                        }
                    }
                }
            }
        }
        // the tile's p / m / v (and dense gradient) are requested once the tile's LDS adds have been issued: they travel
        // while the adds drain (and while the CU's other workgroup works); requested at the top of the tile they would
        // have to be held in registers beside a whole round of records, which costs the second workgroup per CU
        constexpr int F = kTileElems / (4 * kTileThreads);
        // (no branch per piece: a short last tile re-requests its last piece, and the dense gradient is one
        // workgroup-uniform branch of its own -- conditional loads make the compiler wait between the pieces)
        // the dense gradient is requested beside p / m / v only where every tile reads it (RANGE: the sharded tail); on one
        // GPU it is read by the few tiles of levels too small to bin, piece by piece where it is used -- its 16 registers
        // are what lets four of these workgroups share a CU with a marching wavefront on every SIMD
        constexpr bool DENSE_AHEAD = RANGE;
        float4 p4[F], m4[F], v4[F], d4[DENSE_AHEAD ? F : 1];
        const uint32_t last4 = nrows * C - 4u;                  // nrows is a multiple of 8
#pragma unroll
        for (int f = 0; f < F; f++) {
            const uint32_t i = (threadIdx.x + f * kTileThreads) * 4, ic = i < nrows * C ? i : last4;
            p4[f] = *reinterpret_cast<const float4*>(P + base + ic);      // (p is read again by the next forward: kept in the caches)
            m4[f] = TA_LOAD4(M + base + ic);
            v4[f] = TA_LOAD4(V + base + ic);
        }
        if constexpr (DENSE_AHEAD) {
            if (dense) {
#pragma unroll
                for (int f = 0; f < F; f++) {
                    const uint32_t i = (threadIdx.x + f * kTileThreads) * 4, ic = i < nrows * C ? i : last4;
                    d4[f] = *reinterpret_cast<const float4*>(G + base + ic);
                }
            } else {
#pragma unroll
                for (int f = 0; f < F; f++) d4[f] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        // (the next tile -- the answer to this tile's request -- and its count: requested here, behind p / m / v, so that no load
        //  is pending where the compiler drains the queue -- at the record loop's entry and where registers of conditionally
        //  consumed loads are reused)
        if (threadIdx.x == 0) s_next = pulled;
        __syncthreads();                            // (also: the tile's LDS adds are complete)
        const uint32_t item_next = s_next;
        n_ahead = first_cursor(item_next);
#pragma unroll
        for (int f = 0; f < F; f++) {
            const uint32_t i = (threadIdx.x + f * kTileThreads) * 4;
            if (i < nrows * C) {
                float4 g4;
                if constexpr (DENSE_AHEAD) {
                    g4 = d4[f];
                } else {
                    g4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (dense) g4 = *reinterpret_cast<const float4*>(G + base + i);
                }
                if (binned) {
                    const double2 a0 = *reinterpret_cast<const double2*>(acc + i);
                    const double2 a1 = *reinterpret_cast<const double2*>(acc + i + 2);
                    // (fp64 sum rounded once, then the dense part: what pass B + the dense buffer would have held)
                    g4.x += (float)a0.x; g4.y += (float)a0.y; g4.z += (float)a1.x; g4.w += (float)a1.y;
                }
                if (RANGE) { g4.x *= own.rec_scale; g4.y *= own.rec_scale; g4.z *= own.rec_scale; g4.w *= own.rec_scale; }
                if (AMP) { g4.x *= inv_scale; g4.y *= inv_scale; g4.z *= inv_scale; g4.w *= inv_scale; }
                if ((!AMP || !skip) && (!RANGE || (i >= clip_lo && i < clip_hi))) {
                    tile_adam1(p4[f].x, g4.x, m4[f].x, v4[f].x, ad);
                    tile_adam1(p4[f].y, g4.y, m4[f].y, v4[f].y, ad);
                    tile_adam1(p4[f].z, g4.z, m4[f].z, v4[f].z, ad);
                    tile_adam1(p4[f].w, g4.w, m4[f].w, v4[f].w, ad);
                    *reinterpret_cast<float4*>(P + base + i) = p4[f];
                    TA_STORE4(M + base + i, m4[f]);
                    TA_STORE4(V + base + i, v4[f]);
                }
                if (dense) *reinterpret_cast<float4*>(G + base + i) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        __syncthreads();
#ifdef ENERF_TA_TIMING
        if (threadIdx.x == 0 && item < 4096) {
            g_ta_log[4 * item] = level; g_ta_log[4 * item + 1] = ta_n; g_ta_log[4 * item + 2] = ta_t0;
            g_ta_log[4 * item + 3] = (uint32_t)__builtin_amdgcn_s_memrealtime();
        }
#endif
        item = item_next;
    }
    // small parameters (MLP weights; dense gradient, not cleared): their ~10 k elements are dealt over ALL workgroups, a
    // slice of consecutive elements each, one element per thread -- one memory latency at the end of the kernel.  (One
    // workgroup per tensor, looping, was a tail of sixteen dependent latencies behind that workgroup's tiles.)
    if (!AMP && !RANGE && ps.n != 0) {
        // common.h PartialSums: the small tensors' gradients arrive as per-workgroup partial sums (the fused MLP backward's)
        // and are summed here -- 16 values at a time, 16 threads a value (each 1/16 of the partial sums, four independent
        // chains, fixed order), combined through the tile sums' LDS, which nobody uses any more; the sum is stored as the
        // gradient and goes straight into the element's update
        float* red = reinterpret_cast<float*>(acc);
        const uint32_t chunk = div_up(ps.n, nb);
        const uint32_t j = threadIdx.x & 15u, q = threadIdx.x >> 4;
        for (uint32_t c0 = 0; c0 < chunk; c0 += 16u) {
            const uint32_t i = bid * chunk + c0 + j;
            const bool live = c0 + j < chunk && i < ps.n;
            float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
            if (live) {
                const float* col = ps.partial + i;
                uint32_t b = q;
                for (; b + 48u < ps.parts; b += 64u) {
                    s0 += col[(size_t)b * ps.stride];
                    s1 += col[(size_t)(b + 16u) * ps.stride];
                    s2 += col[(size_t)(b + 32u) * ps.stride];
                    s3 += col[(size_t)(b + 48u) * ps.stride];
                }
                for (; b < ps.parts; b += 16u) s0 += col[(size_t)b * ps.stride];
            }
            red[q * 16u + j] = (s0 + s1) + (s2 + s3);
            __syncthreads();
            if (q == 0 && live) {
                float gsum = 0.0f;
#pragma unroll
                for (uint32_t w = 0; w < kTileThreads / 16u; w++) gsum += red[w * 16u + j];
                const uint32_t code = ps.map[i];
                if (code != 0xffffffffu) {
                    const uint32_t k = code >> 24, e = code & 0xffffffu;
                    AdamScalars a2 = ad;
                    a2.step_size = small.step_size[k];
                    a2.inv_bc2_sqrt = small.inv_bc2_sqrt[k];
                    float pv = small.p[k][e], mv = small.m[k][e], vv = small.v[k][e];
                    tile_adam1(pv, gsum, mv, vv, a2);
                    small.p[k][e] = pv; small.m[k][e] = mv; small.v[k][e] = vv;
                    const_cast<float*>(small.g[k])[e] = gsum;
                }
            }
            __syncthreads();
        }
    } else if (small.count != 0 && (!AMP || !skip)) {
        uint32_t total_small = 0;
        for (uint32_t k = 0; k < small.count; k++) total_small += small.n[k];
        const uint32_t chunk = div_up(total_small, nb);
        for (uint32_t t = threadIdx.x; t < chunk; t += kTileThreads) {
            uint32_t e = bid * chunk + t;
            if (e >= total_small) break;
            uint32_t k = 0;
            while (e >= small.n[k]) { e -= small.n[k]; k++; }
            AdamScalars a2 = ad;
            a2.step_size = small.step_size[k];
            a2.inv_bc2_sqrt = small.inv_bc2_sqrt[k];
            if (AMP) amp_scalars(a2, amp.small_lr[k], amp.small_step[k], skipped);
            float pv = small.p[k][e], mv = small.m[k][e], vv = small.v[k][e];
            tile_adam1(pv, AMP ? small.g[k][e] * inv_scale : small.g[k][e], mv, vv, a2);
            small.p[k][e] = pv; small.m[k][e] = mv; small.v[k][e] = vv;
        }
    }
#ifdef ENERF_TA_TIMING
    __syncthreads();
    if (threadIdx.x == 0 && bid < 2048) g_ta_wg[2 * bid + 1] = (uint32_t)__builtin_amdgcn_s_memrealtime();
#endif
    TA_MARK_OUT(5);
    // (nobody asks for a tile any more once a workgroup is here: its own request came back >= total)
    if (threadIdx.x == 0 && atomicInc(&g_ta_next[1], nb - 1u) == nb - 1u) g_ta_next[0] = 0;
}

#ifdef ENERF_TA_TIMING
}  // namespace
extern "C" int enerf_debug_ta_log(uint32_t* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ta_log), sizeof(uint32_t) * 4 * 4096) == hipSuccess ? 0 : -1;
}
extern "C" int enerf_debug_bin_ph(unsigned long long* out, int reset) {
    if (reset) {
        unsigned long long z[32 * 8] = {0};
        return hipMemcpyToSymbol(HIP_SYMBOL(g_bin_ph), z, sizeof(z)) == hipSuccess ? 0 : -1;
    }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bin_ph), sizeof(unsigned long long) * 32 * 8) == hipSuccess ? 0 : -1;
}
extern "C" int enerf_debug_ta_marks(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ta_marks), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : -1;
}
extern "C" int enerf_debug_ta_wg(uint32_t* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ta_wg), sizeof(uint32_t) * 2 * 2048) == hipSuccess ? 0 : -1;
}
namespace {
#endif
// grad_inputs[b,d] = sum_{l,c} grad[l,b,c] * dy_dx[b,l,d,c]   (gridencoder.cu:314-340)
template <typename T, int D, int C>
__global__ void __launch_bounds__(256) k_grid_input_bwd(const T* __restrict__ grad, const T* __restrict__ dy_dx,
                                                        T* __restrict__ grad_inputs, uint32_t B, uint32_t L,
                                                        int grad_layout) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D;
    const uint32_t d = t - b * D;
    const T* jac = dy_dx + (size_t)b * L * D * C;
    float result = 0;
    for (uint32_t l = 0; l < L; l++) {
        const T* g = grad + (grad_layout == 0 ? ((size_t)l * B + b) * C
                                              : grad_layout == 1 ? ((size_t)b * L + l) * C
                                                                 : ((size_t)l * ((B + 31u) & ~31u) + b) * C);
#pragma unroll
        for (int c = 0; c < C; c++) result = acc_feat(result, to_f(g[c]), jac[((size_t)l * D + d) * C + c]);   // (Half * Half: the product of two halves is exact in float)
    }
    grad_inputs[t] = from_f<T>(result);
}

uint32_t* bin_cursors() {
    if (!g_bin_cursors) {
        const size_t bytes = sizeof(uint32_t) * kMaxLevels * kMaxBins;
        if (hipMalloc((void**)&g_bin_cursors, bytes) != hipSuccess) return nullptr;
        if (hipMemset(g_bin_cursors, 0, bytes) != hipSuccess) return nullptr;
    }
    return g_bin_cursors;
}

int fill_level_tab(LevelTab& tab, uint32_t L, float S, uint32_t H, float in_add, float in_mul) {
    tab.in_add = in_add;
    tab.in_mul = in_mul;
    if (L == 0 || L > kMaxLevels) return -1;
    for (uint32_t l = 0; l < L; l++) {
        // gridencoder.cu:124-126: fp32 exp2f, fp32 multiply/subtract, ceil
        const float scale = exp2f((float)l * S) * (float)H - 1.0f;
        tab.scale[l] = scale;
        tab.resolution[l] = (uint32_t)ceil(scale) + 1;
    }
    tab.level_mask = g_level_mask;
    tab.valid_rows = nullptr;
    tab.valid_base = tab.valid_cap = 0;
    return 0;
}

// enerf::grid_valid_rows (common.h): set by the whole-step entry points around their forward .. backward
static const int32_t* g_grid_valid_rows = nullptr;
static uint32_t g_grid_valid_base = 0, g_grid_valid_cap = 0;
static inline void apply_valid_rows(LevelTab& tab) {
    tab.valid_rows = g_grid_valid_rows;
    tab.valid_base = g_grid_valid_base;
    tab.valid_cap = g_grid_valid_cap;
}

__global__ void k_prof_mark() {}

static SplitJob g_carry;              // enerf::grid_fwd_carry
static bool g_carry_armed = false;
static PartialSums g_partial_sums;     // enerf::grid_adam_partial_sums
static bool g_partial_armed = false;
static MarchCountJob g_count_job[2];    // enerf::tile_adam_carry_count
static uint32_t g_count_armed = 0;      // jobs waiting

template <typename T, int D>
int launch_fwd(const float* inputs, const T* emb, const int32_t* offsets, T* outputs, uint32_t B, uint32_t C, uint32_t L,
               const LevelTab& tab, bool calc, T* dy_dx, uint32_t gridtype, int layout, hipStream_t s,
               hipEvent_t ev_start, hipEvent_t ev_stop, const SweepGen& gen = SweepGen{}) {
    const uint32_t nchunks = div_up(layout == 2 ? ((B + 31u) & ~31u) : B, kPtsPerBlock);
    const uint32_t nblocks = fwd_blocks(nchunks, L);
    SplitJob job{};
    uint32_t carry = 0;
    if (std::is_same<T, float>::value && D == 3 && C == 2 && g_carry_armed) {      // enerf::grid_fwd_carry: a job rides along
        g_carry_armed = false;
        job = g_carry;
        carry = kCarryBlocks;
    }
#define ENERF_GF(CC)                                                                                               \
    do {                                                                                                           \
        if (ev_start) {                                                                                            \
            /* the interval starts at the end of a one-wavefront marker launched right before: two stop events,     \
               which cost the stream nothing, where a start event on the kernel itself costs it ~10 us of idle    \
               queue around the launch (tools/launch_chain.hip; the marker's end is the kernel's dispatch) */     \
            hipExtLaunchKernelGGL(k_prof_mark, dim3(1), dim3(64), 0, s, nullptr, ev_start, 0);                     \
            hipExtLaunchKernelGGL((k_grid_fwd<T, D, CC>), dim3(nblocks + carry), dim3(kPtsPerBlock), 0, s, nullptr, \
                                  ev_stop, 0, inputs, emb, offsets, outputs, B, L, tab, calc, dy_dx, gridtype,     \
                                  layout, nchunks, gen, job, carry);                                               \
        }                                                                                                          \
        else                                                                                                       \
            k_grid_fwd<T, D, CC><<<nblocks + carry, kPtsPerBlock, 0, s>>>(inputs, emb, offsets, outputs, B, L, tab, calc, \
                                                                          dy_dx, gridtype, layout, nchunks, gen, job, carry); \
    } while (0)
    switch (C) {
        case 1: ENERF_GF(1); break;
        case 2: ENERF_GF(2); break;
        case 4: ENERF_GF(4); break;
        case 8: ENERF_GF(8); break;
        default: ENERF_BADARG("GridEncoding: C must be 1, 2, 4, or 8.");
    }
#undef ENERF_GF
    return 0;
}

// A deferred flush in progress: record lists written by one or more backward calls wait for k_grid_tile_adam.  The
// list geometry (`region`) is fixed by the first call of the session.
struct PendingRecords {
    uint32_t region = 0, L = 0, C = 0, D = 0, min_tiles = 0;
};
static PendingRecords g_pending;
// enerf_grid_owner_range: [lo, hi) elements of the flat table this rank owns (hi == lo: none set), records' scale
static size_t g_own_lo = 0, g_own_hi = 0;
static float g_own_scale = 1.0f;
// device counters of records that did not fit their list: two, used by alternate sessions -- the flush of one session
// clears the other's, so opening a session costs no memset launch
static uint32_t* g_overflow = nullptr;
static uint32_t g_session = 0;                  // parity selects the counter of the open / next session

static uint32_t* overflow_counter() {
    if (!g_overflow) {
        if (hipMalloc((void**)&g_overflow, 2 * sizeof(uint32_t)) != hipSuccess) return nullptr;
        if (hipMemset(g_overflow, 0, 2 * sizeof(uint32_t)) != hipSuccess) return nullptr;
    }
    return g_overflow + (g_session & 1u);
}

template <typename T, int D>
int launch_bwd(const T* grad, const float* inputs, const int32_t* offsets, T* grad_emb, uint32_t B, uint32_t C,
               uint32_t L, const LevelTab& tab, bool calc, const T* dy_dx, T* grad_inputs, uint32_t gridtype,
               int layout, hipStream_t s, uint32_t flags = 0, uint32_t reserve_B = 0) {
    const uint32_t nchunks = div_up(B, kPtsPerBlock);
    const uint32_t nblocks = 8u * nchunks * div_up(L, 8u);
    // fp32 tables and enough samples: the binned path (the atomic kernel then only sees levels it cannot bin).  A
    // deferred session (several backward calls, one flush by k_grid_tile_adam) is decided ONCE, from the step's total
    // (reserve_B): a call that joins an open session is binned whatever its own B -- tile_adam reads the dense buffer
    // only for the levels it does not bin, so a call routed through the atomics in mid-session would be lost
    const bool session_open = g_pending.region != 0;
    const uint32_t decide_B = ((flags & 1u) != 0 && reserve_B > B) ? reserve_B : B;
    const bool binned = std::is_same<T, float>::value &&
                        (session_open ? (flags & 1u) != 0 : decide_B >= g_binned_min_batch);
    const bool defer = binned && (flags & 1u) != 0;
    uint32_t* recs = nullptr;
    uint32_t* cursors = nullptr;
    uint32_t* overflow = nullptr;
    // records per level: twice the 2^D * B a level can produce (a deferred session: of all its calls together)
    uint32_t region = 2u * (1u << D) * (defer && reserve_B > B ? reserve_B : B);
    if (g_pending.region != 0) {
        if (!defer || g_pending.L != L || g_pending.C != C || g_pending.D != (uint32_t)D) {
            set_error("grid_encode_backward: a deferred flush is pending (enerf_grid_adam_from_records must run first; "
                      "calls that join it pass the defer flag, the same L / C / D and a binned batch)");
            return ENERF_E_BADARG;
        }
        region = g_pending.region;                    // later calls of the session append to the same lists
    }
    if (binned) {
        if (int eg = single_device_guard("grid_encode_backward")) return eg;
        if (int ew = workspace_family_enter(1, s)) return ew;
        recs = (uint32_t*)workspace(WS_GRIDBWD, sizeof(uint32_t) * (size_t)L * region * (1 + C));
        cursors = bin_cursors();
        overflow = overflow_counter();
        if (!recs || !cursors || !overflow) return ENERF_E_NOMEM;
    }
    const uint32_t min_tiles = binned ? g_binned_min_tiles : 0u;
    if (defer && g_pending.region == 0) {
        g_pending.region = region; g_pending.L = L; g_pending.C = C; g_pending.D = (uint32_t)D;
        g_pending.min_tiles = min_tiles;
    }
    const bool flush_now = g_pending.region == 0;
#define ENERF_GB(CC)                                                                                             \
    do {                                                                                                         \
        if (!binned)                                                                                             \
            k_grid_bwd<T, D, CC><<<nblocks, kPtsPerBlock, 0, s>>>(grad, inputs, offsets, grad_emb, B, L, tab,    \
                                                                  gridtype, layout, nchunks, min_tiles);         \
        if constexpr (std::is_same<T, float>::value) {                                                           \
            if (binned) {                                                                                        \
                constexpr int kBinPts = bin_pts(CC);                                                             \
                const uint32_t bchunks = div_up(B, (uint32_t)kBinPts);                                           \
                k_grid_bwd_bin<D, CC, kBinPts><<<8u * bchunks * div_up(L, 8u), kBinPts, 0, s>>>(                 \
                    grad, inputs, offsets, grad_emb, B, L, tab, gridtype, layout, bchunks, min_tiles, recs, cursors, \
                    region, flush_now ? nullptr : overflow);                                                     \
                if (flush_now)                                                                                   \
                    k_grid_bwd_tile<CC><<<kTilesPerCu * num_cus(), kTileThreads, 0, s>>>(offsets, grad_emb, L, tab, min_tiles, \
                                                                           recs, cursors, region);               \
                else if (g_own_hi > g_own_lo) /* sharded tail: everything outside this rank's slice is flushed now */ \
                    k_grid_bwd_tile<CC><<<kTilesPerCu * num_cus(), kTileThreads, 0, s>>>(offsets, grad_emb, L, tab, min_tiles, \
                                                                           recs, cursors, region, g_own_lo, g_own_hi); \
            }                                                                                                    \
        }                                                                                                        \
        if (calc)                                                                                                \
            k_grid_input_bwd<T, D, CC><<<div_up(B * D, 256), 256, 0, s>>>(grad, dy_dx, grad_inputs, B, L, layout); \
    } while (0)
    switch (C) {
        case 1: ENERF_GB(1); break;
        case 2: ENERF_GB(2); break;
        case 4: ENERF_GB(4); break;
        case 8: ENERF_GB(8); break;
        default: ENERF_BADARG("GridEncoding: C must be 1, 2, 4, or 8.");
    }
#undef ENERF_GB
    return 0;
}

}  // namespace

bool enerf::grid_adam_partial_sums(const PartialSums* job) {
    const bool waiting = g_partial_armed;
    g_partial_armed = job != nullptr && job->n != 0;
    if (g_partial_armed) g_partial_sums = *job;
    return waiting;
}

void enerf::grid_valid_rows(const int32_t* device_count, uint32_t base, uint32_t cap) {
    g_grid_valid_rows = device_count;
    g_grid_valid_base = device_count ? base : 0u;
    g_grid_valid_cap = device_count ? cap : 0u;
}

bool enerf::tile_adam_carry_count(const MarchCountJob* job) {
    const bool waiting = g_count_armed != 0;
    if (job == nullptr || job->blocks == 0 || job->N == 0) {
        g_count_armed = 0;
        return waiting;
    }
    if (g_count_armed >= 2u) g_count_armed = 0;      // (never more than two: a third starts over)
    g_count_job[g_count_armed++] = *job;
    return waiting;
}

bool enerf::grid_fwd_carry(const SplitJob* job) {
    const bool waiting = g_carry_armed;
    g_carry_armed = job != nullptr && job->threads <= kCarryBlocks * kPtsPerBlock;
    if (g_carry_armed) g_carry = *job;
    return waiting;
}

extern "C" {

// profiling aid, not part of the reference surface: restrict both grid kernels to the levels set in `mask`
int enerf_debug_grid_level_mask(uint32_t mask) {
    g_level_mask = mask;
    return 0;
}

// testing / profiling aid: fp32 batches of at least `min_batch` samples send the levels spanning at least `min_tiles`
// 128-KiB tiles through the binned (record list + LDS tile) backward path; everything else takes the global-atomic
// kernel.  Defaults 16384 / 8; min_batch 0xffffffff disables the binned path.
int enerf_debug_grid_bwd_binned(uint32_t min_batch, uint32_t min_tiles) {
    g_binned_min_batch = min_batch;
    g_binned_min_tiles = min_tiles ? min_tiles : 1u;
    return 0;
}

int enerf_grid_encode_forward(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                              uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs,
                              void* dy_dx, uint32_t gridtype, int dtype, int out_layout, float in_add, float in_mul,
                              enerf_stream_t stream) {
    if (B == 0) return 0;
    LevelTab tab;
    if (fill_level_tab(tab, L, S, H, in_add, in_mul)) ENERF_BADARG("GridEncoding: L must be in [1, %d], got %u", kMaxLevels, L);
    if (dtype != ENERF_F32 && dtype != ENERF_F16) ENERF_BADARG("GridEncoding: dtype must be f32 or f16");
    if (out_layout < 0 || out_layout > 2) ENERF_BADARG("GridEncoding: out_layout must be 0, 1 or 2, got %d", out_layout);
    apply_valid_rows(tab);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(ENERF_K_GRID_FWD, s, true);   // timed with the kernel's own begin / end stamps
    prof.units((double)B);
    int rc = 0;
    const bool calc = calc_grad_inputs != 0;
    if (dtype == ENERF_F32) {
        if (D == 3) rc = launch_fwd<float, 3>(inputs, (const float*)embeddings, offsets, (float*)outputs, B, C, L, tab, calc, (float*)dy_dx, gridtype, out_layout, s, prof.start(), prof.stop());
        else if (D == 2) rc = launch_fwd<float, 2>(inputs, (const float*)embeddings, offsets, (float*)outputs, B, C, L, tab, calc, (float*)dy_dx, gridtype, out_layout, s, prof.start(), prof.stop());
        else ENERF_BADARG("GridEncoding: D must be 2 or 3.");
    } else {
        if (D == 3) rc = launch_fwd<__half, 3>(inputs, (const __half*)embeddings, offsets, (__half*)outputs, B, C, L, tab, calc, (__half*)dy_dx, gridtype, out_layout, s, prof.start(), prof.stop());
        else if (D == 2) rc = launch_fwd<__half, 2>(inputs, (const __half*)embeddings, offsets, (__half*)outputs, B, C, L, tab, calc, (__half*)dy_dx, gridtype, out_layout, s, prof.start(), prof.stop());
        else ENERF_BADARG("GridEncoding: D must be 2 or 3.");
    }
    if (rc) return rc;
    ENERF_LAUNCH_CHECK("grid_encode_forward");
    return 0;
}

int enerf_grid_encode_forward_sweep(const void* embeddings, const int32_t* offsets, void* outputs, uint32_t n_cascades,
                                    uint32_t grid_size, float bound, uint64_t seed, uint32_t C, uint32_t L, float S,
                                    uint32_t H, uint32_t gridtype, int out_layout, float in_add, float in_mul,
                                    enerf_stream_t stream) {
    if (n_cascades < 1 || n_cascades > 8 || grid_size < 16 || grid_size > 512 || (grid_size & (grid_size - 1)) != 0)
        ENERF_BADARG("grid_encode_forward_sweep: cascades=%u grid_size=%u", n_cascades, grid_size);
    LevelTab tab;
    if (fill_level_tab(tab, L, S, H, in_add, in_mul)) ENERF_BADARG("GridEncoding: L must be in [1, %d], got %u", kMaxLevels, L);
    if (out_layout < 0 || out_layout > 2) ENERF_BADARG("GridEncoding: out_layout must be 0, 1 or 2, got %d", out_layout);
    SweepGen gen;
    gen.cs = make_cascades(n_cascades, grid_size, bound);
    gen.seed = seed;
    gen.H = grid_size;
    gen.logH = 0;
    while ((1u << gen.logH) < grid_size) gen.logH++;
    gen.enabled = 1;
    const uint32_t B = n_cascades << (3 * gen.logH);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(ENERF_K_GRID_FWD, s, true);
    prof.units((double)B);
    const int rc = launch_fwd<float, 3>(nullptr, (const float*)embeddings, offsets, (float*)outputs, B, C, L, tab, false,
                                        (float*)nullptr, gridtype, out_layout, s, prof.start(), prof.stop(), gen);
    if (rc) return rc;
    ENERF_LAUNCH_CHECK("grid_encode_forward_sweep");
    return 0;
}

int enerf_grid_encode_backward_ex(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                                  void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                  int calc_grad_inputs, const void* dy_dx, void* grad_inputs, uint32_t gridtype, int dtype,
                                  int grad_layout, float in_add, float in_mul, uint32_t flags, uint32_t reserve_B,
                                  enerf_stream_t stream) {
    (void)embeddings;
    if (B == 0) return 0;
    LevelTab tab;
    if (fill_level_tab(tab, L, S, H, in_add, in_mul)) ENERF_BADARG("GridEncoding: L must be in [1, %d], got %u", kMaxLevels, L);
    if (dtype != ENERF_F32 && dtype != ENERF_F16) ENERF_BADARG("GridEncoding: dtype must be f32 or f16");
    if (grad_layout < 0 || grad_layout > 2) ENERF_BADARG("GridEncoding: grad_layout must be 0, 1 or 2, got %d", grad_layout);
    apply_valid_rows(tab);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(ENERF_K_GRID_BWD, s);
    prof.units((double)B);
    int rc = 0;
    const bool calc = calc_grad_inputs != 0;
    if (dtype == ENERF_F32) {
        if (D == 3) rc = launch_bwd<float, 3>((const float*)grad, inputs, offsets, (float*)grad_embeddings, B, C, L, tab, calc, (const float*)dy_dx, (float*)grad_inputs, gridtype, grad_layout, s, flags, reserve_B);
        else if (D == 2) rc = launch_bwd<float, 2>((const float*)grad, inputs, offsets, (float*)grad_embeddings, B, C, L, tab, calc, (const float*)dy_dx, (float*)grad_inputs, gridtype, grad_layout, s, flags, reserve_B);
        else ENERF_BADARG("GridEncoding: D must be 2 or 3.");
    } else {
        if (D == 3) rc = launch_bwd<__half, 3>((const __half*)grad, inputs, offsets, (__half*)grad_embeddings, B, C, L, tab, calc, (const __half*)dy_dx, (__half*)grad_inputs, gridtype, grad_layout, s);
        else if (D == 2) rc = launch_bwd<__half, 2>((const __half*)grad, inputs, offsets, (__half*)grad_embeddings, B, C, L, tab, calc, (const __half*)dy_dx, (__half*)grad_inputs, gridtype, grad_layout, s);
        else ENERF_BADARG("GridEncoding: D must be 2 or 3.");
    }
    if (rc) return rc;
    ENERF_LAUNCH_CHECK("grid_encode_backward");
    return 0;
}

int enerf_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                               void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                               int calc_grad_inputs, const void* dy_dx, void* grad_inputs, uint32_t gridtype, int dtype,
                               int grad_layout, float in_add, float in_mul, enerf_stream_t stream) {
    return enerf_grid_encode_backward_ex(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H,
                                         calc_grad_inputs, dy_dx, grad_inputs, gridtype, dtype, grad_layout, in_add, in_mul,
                                         0u, 0u, stream);
}

// Sharded data-parallel tail: elements [lo, hi) of the flat table (multiples of 4) belong to this rank and `rec_scale`
// (1 / ranks) turns the summed gradient -- reduce-scattered dense part + its own record lists -- into the average.  While set (hi > lo), a deferred
// enerf_grid_encode_backward_ex flushes every list outside the range into the dense gradient right away and
// enerf_grid_adam_from_records(_ex) updates the range only (and clears the dense gradient everywhere).  lo == hi clears.
int enerf_grid_owner_range(uint64_t lo, uint64_t hi, float rec_scale) {
    if (hi < lo) ENERF_BADARG("grid_owner_range: hi < lo");
    g_own_lo = (size_t)lo;
    g_own_hi = (size_t)hi;
    g_own_scale = hi > lo ? rec_scale : 1.0f;
    return 0;
}

int enerf_grid_records_discard(enerf_stream_t stream) {
    // drop a pending deferred flush (the step that opened it failed before its optimizer pass): empty the lists
    if (g_pending.region != 0) {
        uint32_t* cursors = bin_cursors();
        if (cursors) (void)hipMemsetAsync(cursors, 0, sizeof(uint32_t) * kMaxLevels * kMaxBins, (hipStream_t)stream);
        if (g_overflow) (void)hipMemsetAsync(g_overflow + (g_session & 1u), 0, sizeof(uint32_t), (hipStream_t)stream);
        g_pending = PendingRecords();
    }
    return 0;
}

int enerf_grid_adam_from_records(float* p, float* g, float* m, float* v, const int32_t* offsets, uint32_t L, uint32_t C,
                                 float lr, float beta1, float beta2, float eps, uint32_t step, enerf_stream_t stream) {
    return enerf_grid_adam_from_records_ex(p, g, m, v, offsets, L, C, lr, beta1, beta2, eps, step, 0, nullptr, nullptr,
                                           nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}

int enerf_grid_adam_from_records_ex(float* p, float* g, float* m, float* v, const int32_t* offsets, uint32_t L, uint32_t C,
                                    float lr, float beta1, float beta2, float eps, uint32_t step, uint32_t n_small,
                                    float* const* sp, const float* const* sg, float* const* sm, float* const* sv,
                                    const uint32_t* sn, const float* slr, const uint32_t* sstep, enerf_stream_t stream) {
    if (n_small > (uint32_t)kMaxSmallAdam) ENERF_BADARG("grid_adam_from_records: at most %d extra tensors", kMaxSmallAdam);
    SmallAdam small;
    small.count = n_small;
    for (uint32_t k = 0; k < n_small; k++) {
        if (!sp[k] || !sg[k] || !sm[k] || !sv[k] || sstep[k] == 0) ENERF_BADARG("grid_adam_from_records: extra tensor %u", k);
        small.p[k] = sp[k]; small.g[k] = sg[k]; small.m[k] = sm[k]; small.v[k] = sv[k]; small.n[k] = sn[k];
        small.step_size[k] = (float)((double)slr[k] / (1.0 - pow((double)beta1, (double)sstep[k])));
        small.inv_bc2_sqrt[k] = (float)(1.0 / sqrt(1.0 - pow((double)beta2, (double)sstep[k])));
    }
    if (!p || !g || !m || !v || !offsets || L == 0 || L > (uint32_t)kMaxLevels || step == 0)
        ENERF_BADARG("grid_adam_from_records: bad arguments (L=%u step=%u)", L, step);
    if (g_pending.region != 0 && (g_pending.L != L || g_pending.C != C))
        ENERF_BADARG("grid_adam_from_records: the pending records belong to a table with L=%u C=%u", g_pending.L, g_pending.C);
    hipStream_t s = (hipStream_t)stream;
    if (int eg = single_device_guard("grid_adam_from_records")) return eg;
    // torch.optim.Adam's scalars, as enerf_adam_step_multi computes them
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    AdamScalars ad = {beta1, beta2, eps, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2))};
    uint32_t* cursors = bin_cursors();
    uint32_t* overflow = overflow_counter();
    if (!cursors || !overflow) return ENERF_E_NOMEM;
    uint32_t* other = g_overflow + ((g_session + 1u) & 1u);
    const uint32_t region = g_pending.region;
    if (int ew = workspace_family_enter(1, s)) return ew;
    const uint32_t* recs = region ? (const uint32_t*)workspace(WS_GRIDBWD, sizeof(uint32_t) * (size_t)L * region * (1 + C))
                                  : nullptr;
    const uint32_t min_tiles = region ? g_pending.min_tiles : 0u;
    ProfScope prof(ENERF_K_TABLE_ADAM, s);
    if (g_own_hi > g_own_lo) {
        // the sharded data-parallel tail: Adam on this rank's elements only (k_grid_tile_adam RANGE)
        if (C != 2) ENERF_BADARG("grid_adam_from_records: an owner range serves C = 2 tables");
        if (amp_state().scale) ENERF_BADARG("grid_adam_from_records: owner range and loss scaling do not combine");
        if ((g_own_lo | g_own_hi) & 3u) ENERF_BADARG("grid_adam_from_records: owner range must be multiples of 4 elements");
        k_grid_tile_adam<2, false, true><<<kTilesPerCu * num_cus(), kTileThreads, kTileAccBytes, s>>>(
            offsets, p, g, m, v, L, min_tiles, recs, cursors, region, overflow, other, ad, small, AmpAdam{},
            OwnerRange{g_own_lo, g_own_hi, g_own_scale});
        if (g_pending.region != 0) g_session++;
        g_pending = PendingRecords();
        ENERF_LAUNCH_CHECK("grid_adam_from_records(range)");
        return 0;
    }
    const AmpState as = amp_state();
    if (as.scale) {
        // loss scaling armed (enerf_amp_begin): unscale, skip on a non-finite step, bias corrections from the device
        AmpAdam amp;
        amp.scale = as.scale; amp.found_inf = as.found_inf; amp.skipped = as.skipped;
        amp.lr = lr; amp.step = step;
        for (uint32_t k = 0; k < n_small; k++) { amp.small_lr[k] = slr[k]; amp.small_step[k] = sstep[k]; }
        if (C != 2) ENERF_BADARG("grid_adam_from_records: loss scaling serves C = 2 tables");
        k_grid_tile_adam<2, true><<<kTilesPerCu * num_cus(), kTileThreads, kTileAccBytes, s>>>(offsets, p, g, m, v, L, min_tiles, recs,
                                                                                    cursors, region, overflow, other, ad, small, amp);
        if (g_pending.region != 0) g_session++;
        g_pending = PendingRecords();
        ENERF_LAUNCH_CHECK("grid_adam_from_records(amp)");
        return 0;
    }
    PartialSums ps{nullptr, nullptr, 0, 0, 0};
    if (g_partial_armed && C == 2 && n_small == 5) {      // enerf::grid_adam_partial_sums: the small gradients are summed here
        g_partial_armed = false;
        ps = g_partial_sums;
    }
    switch (C) {
        case 1: k_grid_tile_adam<1><<<kTilesPerCu * num_cus(), kTileThreads, kTileAccBytes, s>>>(offsets, p, g, m, v, L, min_tiles, recs, cursors, region, overflow, other, ad, small); break;
        case 2:
            if (g_count_armed) {                     // enerf::tile_adam_carry_count: the next batch's march count pass(es) ride along
                const MarchCountJob j0 = g_count_job[0], j1 = g_count_armed > 1u ? g_count_job[1] : MarchCountJob{};
                g_count_armed = 0;
                k_grid_tile_adam<2, false, false, true><<<kTilesPerCu * num_cus() + j0.blocks + j1.blocks, kTileThreads, kTileAccBytes, s>>>(
                    offsets, p, g, m, v, L, min_tiles, recs, cursors, region, overflow, other, ad, small, AmpAdam{},
                    OwnerRange{0, 0, 1.0f}, ps, j0, j1);
            } else {
                k_grid_tile_adam<2><<<kTilesPerCu * num_cus(), kTileThreads, kTileAccBytes, s>>>(offsets, p, g, m, v, L, min_tiles, recs, cursors, region, overflow, other, ad, small, AmpAdam{}, OwnerRange{0, 0, 1.0f}, ps);
            }
            break;
        case 4: k_grid_tile_adam<4><<<kTilesPerCu * num_cus(), kTileThreads, kTileAccBytes, s>>>(offsets, p, g, m, v, L, min_tiles, recs, cursors, region, overflow, other, ad, small); break;
        case 8: k_grid_tile_adam<8><<<kTilesPerCu * num_cus(), kTileThreads, kTileAccBytes, s>>>(offsets, p, g, m, v, L, min_tiles, recs, cursors, region, overflow, other, ad, small); break;
        default: ENERF_BADARG("grid_adam_from_records: C must be 1, 2, 4, or 8.");
    }
    if (g_pending.region != 0) g_session++;           // the next session counts overflows in the counter just cleared
    g_pending = PendingRecords();
    ENERF_LAUNCH_CHECK("grid_adam_from_records");
    return 0;
}

}  // extern "C"
