// density_update.hip -- the device side of NeRFRenderer.update_extra_state (nerf/renderer.py:472-560 of the reference):
// which cells of the cascaded occupancy grid get re-evaluated, and what happens to their densities afterwards.  The
// density network itself (hash grid + sigma MLP) runs between the two entry points, on the buffers they exchange.
//
//   enerf_density_grid_cells   full sweep : every cell of every cascade, x fastest (renderer.py:484-512)
//                              partial    : per cascade N uniformly drawn cells + N draws (with replacement) from the
//                                           cells with density > 0 (renderer.py:514-527), emitted in Morton order
//                              -> Morton indices + jittered query positions of all cascades in one batch
//   enerf_density_grid_update  tmp_grid[cas, indices] = sigma * scale; EMA max with decay where both are valid;
//                              mean of clamp(grid, 0); packbits against min(mean, density_thresh); step-counter sum
//                              (renderer.py:541-558) -- the mean never visits the host before the bitfield is packed
//
// What the reference does with ~40 torch ops per cascade, three host synchronisations (nonzero, two .item()) and 1 M-point
// scattered gathers is here ~10 launches, one read-back of 16 bytes, and coherent gathers: the sampled cells are sorted
// (23-bit radix sort of (cascade, Morton index) keys) so neighbouring query points share hash-grid rows.  The sampling
// is the same in distribution, not in its random numbers (torch's Philox stream is not reproduced): parity unpinned.

#include "common.h"
#include "sweep_points.h"

using namespace enerf;

namespace {

__device__ __forceinline__ uint32_t morton3(uint32_t x, uint32_t y, uint32_t z) { return sweep_morton3(x, y, z); }
__device__ __forceinline__ uint32_t compact_bits(uint32_t x) {
    x = x & 0x49249249;
    x = (x | (x >> 2)) & 0xc30c30c3;
    x = (x | (x >> 4)) & 0x0f00f00f;
    x = (x | (x >> 8)) & 0xff0000ff;
    x = (x | (x >> 16)) & 0x0000ffff;
    return x;
}

// ---- full sweep: thread per (cascade, cell), x fastest so that consecutive query points are x-neighbours
__global__ void __launch_bounds__(256) k_cells_full(Cascades cs, uint32_t C, uint32_t H, uint32_t logH, uint64_t seed,
                                                    int32_t* __restrict__ indices, float* __restrict__ xyzs) {
    const uint32_t H3 = H * H * H;
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= C * H3) return;
    const uint32_t cas = p / H3, cell = p - cas * H3;
    const uint32_t x = cell & (H - 1), y = (cell >> logH) & (H - 1), z = cell >> (2 * logH);
    indices[p] = (int32_t)morton3(x, y, z);
    cell_position(cs, cas, H, x, y, z, rand4(seed, p), xyzs + (size_t)p * 3);
}

// ---- partial update, step 1: the cells with density > 0, compacted per cascade in index order
constexpr uint32_t kOccBlock = 4096;     // cells per workgroup (256 threads x 16)

__global__ void __launch_bounds__(256) k_occ_count(const float* __restrict__ grid, uint32_t* __restrict__ block_counts) {
    __shared__ uint32_t wsum[4];
    const float* g = grid + (size_t)blockIdx.x * kOccBlock + threadIdx.x * 16;
    uint32_t c = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float4 v = reinterpret_cast<const float4*>(g)[q];
        c += (v.x > 0.0f) + (v.y > 0.0f) + (v.z > 0.0f) + (v.w > 0.0f);
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if (lane_id() == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// one workgroup per cascade: exclusive scan of its block counts (in place), total -> cas_counts[cas]
__global__ void __launch_bounds__(1024) k_occ_scan(uint32_t* __restrict__ block_counts, uint32_t blocks_per_cas,
                                                   uint32_t* __restrict__ cas_counts) {
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t carry_s;
    uint32_t* bc = block_counts + (size_t)blockIdx.x * blocks_per_cas;
    const int lane = lane_id(), wid = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < blocks_per_cas; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < blocks_per_cas ? bc[i] : 0u;
        const uint32_t incl = wave_incl_scan_add_u32(v, lane);
        if (lane == 63) wtot[wid] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wid; w++) woff += wtot[w];
        const uint32_t carry = carry_s;
        if (i < blocks_per_cas) bc[i] = carry + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) cas_counts[blockIdx.x] = carry_s;
}

__global__ void __launch_bounds__(256) k_occ_compact(const float* __restrict__ grid,
                                                     const uint32_t* __restrict__ block_offsets, uint32_t blocks_per_cas,
                                                     uint32_t H3, uint32_t* __restrict__ list) {
    __shared__ uint32_t wsum[4];
    const uint32_t cas = blockIdx.x / blocks_per_cas;
    const uint32_t cell0 = (blockIdx.x - cas * blocks_per_cas) * kOccBlock + threadIdx.x * 16;
    const float* g = grid + (size_t)cas * H3 + cell0;
    uint32_t mask = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float4 v = reinterpret_cast<const float4*>(g)[q];
        mask |= (uint32_t)(v.x > 0.0f) << (4 * q) | (uint32_t)(v.y > 0.0f) << (4 * q + 1) |
                (uint32_t)(v.z > 0.0f) << (4 * q + 2) | (uint32_t)(v.w > 0.0f) << (4 * q + 3);
    }
    const uint32_t c = (uint32_t)__popc(mask);
    const int lane = lane_id(), wid = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan_add_u32(c, lane);
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    uint32_t off = block_offsets[blockIdx.x] + incl - c;
    for (int w = 0; w < wid; w++) off += wsum[w];
    uint32_t* out = list + (size_t)cas * H3 + off;
    while (mask) {
        const int b = __builtin_ctz(mask);
        mask &= mask - 1;
        *out++ = cell0 + (uint32_t)b;
    }
}

// ---- partial update, step 2: (cascade << bits | Morton index) keys of the cells to evaluate
__global__ void __launch_bounds__(256) k_select_keys(uint32_t C, uint32_t N, uint32_t bits, uint32_t H3, uint64_t seed,
                                                     const uint32_t* __restrict__ cas_counts,
                                                     const uint32_t* __restrict__ list, uint32_t* __restrict__ keys,
                                                     uint32_t* __restrict__ sort_counters, uint32_t n_counters) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_counters) sort_counters[i] = 0;                       // (bucket sizes and cursors of the sort that follows)
    if (i >= C * 2 * N) return;
    const uint32_t cas = i / (2 * N), j = i - cas * 2 * N;
    const Rand4 r = rand4(seed ^ 0x5bd1e995u, i);
    uint32_t idx = r.w[0] & (H3 - 1);                               // uniform cell (H^3 is a power of two)
    const uint32_t cnt = cas_counts[cas];
    if (j >= N && cnt) idx = list[(size_t)cas * H3 + (uint32_t)(((uint64_t)r.w[1] * cnt) >> 32)];   // occupied cell
    keys[i] = cas << bits | idx;
}

// ---- partial update, the sort between steps 2 and 3: ascending (cascade << bits | Morton index) keys, so that the encoder's
// gathers of the 1 M scattered evaluations of the reference run in Morton order.  Keys only, values below C << bits:
// a bucket sort by VALUE in four launches --
//   k_sort_count    bucket = key >> low (4096 cells of one cascade in Morton order: a 16^3 block): per-workgroup LDS histogram,
//                   one global atomic per (workgroup, bucket fed);
//   k_sort_scan     exclusive scan of the <= 4096 bucket sizes;
//   k_sort_scatter  the keys into their buckets' ranges, in any order (one reservation per (workgroup, bucket), LDS ranks);
//   k_sort_buckets  one workgroup per bucket: an LDS histogram over the bucket's 2^low possible values, scanned, and every
//                   value written out as often as it occurred -- which IS the sorted order (there is no payload).
// (rounds 1-4: hipcub::DeviceRadixSort, the only library code on the path.)
constexpr uint32_t kSortChunk = 8192;               // keys per workgroup of the count / scatter passes (256 threads x 32)
constexpr uint32_t kSortMaxBuckets = 4096;
constexpr uint32_t kSortMaxLow = 14;                // 2^14 counters of 4 bytes = 64 KiB of LDS in k_sort_buckets

__global__ void __launch_bounds__(256) k_sort_count(const uint32_t* __restrict__ keys, uint32_t P, uint32_t low, uint32_t nb,
                                                    uint32_t* __restrict__ bucket_count) {
    __shared__ uint32_t hist[kSortMaxBuckets];
    for (uint32_t b = threadIdx.x; b < nb; b += 256) hist[b] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kSortChunk;
#pragma unroll 8
    for (uint32_t k = 0; k < kSortChunk / 256; k++) {
        const uint32_t i = base + k * 256 + threadIdx.x;
        if (i < P) atomicAdd(&hist[keys[i] >> low], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += 256)
        if (hist[b]) atomicAdd(&bucket_count[b], hist[b]);
}

// bucket_base = exclusive scan of bucket_count (nb <= 4096: four per thread)
__global__ void __launch_bounds__(1024) k_sort_scan(const uint32_t* __restrict__ bucket_count, uint32_t nb,
                                                    uint32_t* __restrict__ bucket_base) {
    __shared__ uint32_t wtot[16];
    uint32_t v[4], mine = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t b = threadIdx.x * 4 + k;
        v[k] = b < nb ? bucket_count[b] : 0u;
        mine += v[k];
    }
    const int lane = lane_id(), wid = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan_add_u32(mine, lane);
    if (lane == 63) wtot[wid] = incl;
    __syncthreads();
    uint32_t off = incl - mine;
    for (int w = 0; w < wid; w++) off += wtot[w];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t b = threadIdx.x * 4 + k;
        if (b < nb) bucket_base[b] = off;
        off += v[k];
    }
}

__global__ void __launch_bounds__(256) k_sort_scatter(const uint32_t* __restrict__ keys, uint32_t P, uint32_t low, uint32_t nb,
                                                      const uint32_t* __restrict__ bucket_base,
                                                      uint32_t* __restrict__ bucket_cursor, uint32_t* __restrict__ out) {
    __shared__ uint32_t hist[kSortMaxBuckets];       // this chunk's keys per bucket, then the next free slot of its reservation
    for (uint32_t b = threadIdx.x; b < nb; b += 256) hist[b] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kSortChunk;
    uint32_t key[kSortChunk / 256];
#pragma unroll
    for (uint32_t k = 0; k < kSortChunk / 256; k++) {
        const uint32_t i = base + k * 256 + threadIdx.x;
        key[k] = i < P ? keys[i] : 0xffffffffu;
        if (i < P) atomicAdd(&hist[key[k] >> low], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += 256) {
        const uint32_t n = hist[b];
        hist[b] = n ? bucket_base[b] + atomicAdd(&bucket_cursor[b], n) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < kSortChunk / 256; k++)
        if (key[k] != 0xffffffffu) out[atomicAdd(&hist[key[k] >> low], 1u)] = key[k];
}

// one workgroup per bucket (any size): count its keys' low bits, scan, write every value out count[value] times
__global__ void __launch_bounds__(256) k_sort_buckets(const uint32_t* __restrict__ scattered, uint32_t low,
                                                      const uint32_t* __restrict__ bucket_count,
                                                      const uint32_t* __restrict__ bucket_base, uint32_t* __restrict__ out) {
    extern __shared__ uint32_t cnt[];                // 2^low
    __shared__ uint32_t wtot[4];
    const uint32_t b = blockIdx.x, n = bucket_count[b];
    if (n == 0) return;
    const uint32_t base = bucket_base[b], vals = 1u << low, mask = vals - 1u;
    for (uint32_t v = threadIdx.x; v < vals; v += 256) cnt[v] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 256) atomicAdd(&cnt[scattered[base + i] & mask], 1u);
    __syncthreads();
    // thread t owns the values [t * per, (t + 1) * per): their total, an exclusive scan over the threads, then the writes
    const uint32_t per = vals / 256;
    uint32_t mine = 0;
    for (uint32_t k = 0; k < per; k++) mine += cnt[threadIdx.x * per + k];
    const int lane = lane_id(), wid = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan_add_u32(mine, lane);
    if (lane == 63) wtot[wid] = incl;
    __syncthreads();
    uint32_t off = base + incl - mine;
    for (int w = 0; w < wid; w++) off += wtot[w];
    const uint32_t hi = b << low;
    // a value drawn many times (few occupied cells: every draw of the cascade lands on them) is written by the whole
    // workgroup, not by the one thread that owns it
    __shared__ uint32_t heavy_n, heavy[64][3];
    if (threadIdx.x == 0) heavy_n = 0;
    __syncthreads();
    for (uint32_t k = 0; k < per; k++) {
        const uint32_t v = threadIdx.x * per + k, c = cnt[v];
        bool mine_to_write = true;
        if (c > 64u) {
            const uint32_t slot = atomicAdd(&heavy_n, 1u);
            if (slot < 64u) {
                heavy[slot][0] = hi | v; heavy[slot][1] = c; heavy[slot][2] = off;
                mine_to_write = false;
            }
        }
        if (mine_to_write)
            for (uint32_t j = 0; j < c; j++) out[off + j] = hi | v;
        off += c;
    }
    __syncthreads();
    const uint32_t nh = heavy_n < 64u ? heavy_n : 64u;
    for (uint32_t e = 0; e < nh; e++) {
        const uint32_t key = heavy[e][0], c = heavy[e][1], at = heavy[e][2];
        for (uint32_t j = threadIdx.x; j < c; j += 256) out[at + j] = key;
    }
}

// ---- partial update, step 3 (after the sort): keys -> Morton indices + jittered positions
__global__ void __launch_bounds__(256) k_cells_from_keys(Cascades cs, uint32_t P, uint32_t H, uint32_t bits,
                                                         uint64_t seed, const uint32_t* __restrict__ keys,
                                                         int32_t* __restrict__ indices, float* __restrict__ xyzs) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const uint32_t key = keys[p];
    const uint32_t idx = key & ((1u << bits) - 1), cas = key >> bits;
    indices[p] = (int32_t)idx;
    cell_position(cs, cas, H, compact_bits(idx), compact_bits(idx >> 1), compact_bits(idx >> 2), rand4(seed, p),
                  xyzs + (size_t)p * 3);
}

// ---- update: scatter, EMA + mean, packbits
// indices == nullptr: the points are those of the full sweep in its own order (sweep_points.h): cell from p
__global__ void __launch_bounds__(256) k_tmp_scatter(const int32_t* __restrict__ indices, const float* __restrict__ sigmas,
                                                     uint32_t n_per_cas, uint32_t P, uint32_t H3, float scale,
                                                     float* __restrict__ tmp, uint32_t H, uint32_t logH) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const uint32_t cas = p / n_per_cas;
    uint32_t idx;
    if (indices) {
        idx = (uint32_t)indices[p];
    } else {
        const uint32_t cell = p - cas * n_per_cas;
        idx = morton3(cell & (H - 1), (cell >> logH) & (H - 1), cell >> (2 * logH));
    }
    tmp[(size_t)cas * H3 + idx] = sigmas[p] * scale;                        // duplicates: some writer wins, as in torch
}

// partial sums of the grid's mean: kSumSlots doubles, one per 128-byte line
constexpr uint32_t kSumSlots = 16, kSumStride = 16, kSumBytes = kSumSlots * kSumStride * 8;

__global__ void k_zero_sums(double* __restrict__ sum) { sum[threadIdx.x] = 0.0; }

__device__ __forceinline__ float ema1(float g, float t, float decay, float& acc) {
    if (g >= 0.0f && t >= 0.0f) g = fmaxf(g * decay, t);          // `tmp` untouched = NaN bit pattern: never >= 0
    acc += fmaxf(g, 0.0f);
    return g;
}

__global__ void __launch_bounds__(256) k_ema_mean(float* __restrict__ grid, const float* __restrict__ tmp, uint32_t total4,
                                                  float decay, double* __restrict__ sum) {
    __shared__ float wsum[4];
    float acc = 0.0f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += gridDim.x * blockDim.x) {
        float4 g = reinterpret_cast<float4*>(grid)[i];
        const float4 t = reinterpret_cast<const float4*>(tmp)[i];
        g.x = ema1(g.x, t.x, decay, acc); g.y = ema1(g.y, t.y, decay, acc);
        g.z = ema1(g.z, t.z, decay, acc); g.w = ema1(g.w, t.w, decay, acc);
        reinterpret_cast<float4*>(grid)[i] = g;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane_id() == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    // (2048 fp64 atomics on one address queue up for ~12 ns each -- 25 us of a 33 us kernel: sixteen addresses on
    // sixteen lines instead, added up in a fixed order by the reader)
    if (threadIdx.x == 0)
        atomicAdd(sum + (blockIdx.x % kSumSlots) * kSumStride, (double)((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])));
}

__global__ void __launch_bounds__(256) k_packbits_mean(const float* __restrict__ grid, uint32_t nbytes,
                                                       const double* __restrict__ sum, double inv_total,
                                                       float density_thresh, uint8_t* __restrict__ bitfield,
                                                       const int32_t* __restrict__ step_counter, uint32_t total_step,
                                                       double* __restrict__ stats) {
    double total = 0.0;
    for (uint32_t k = 0; k < kSumSlots; k++) total += sum[k * kSumStride];
    const float mean = (float)(total * inv_total);
    const float thresh = fminf(mean, density_thresh);
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n == 0) {
        stats[0] = (double)mean;
        long long c = 0;
        for (uint32_t s = 0; s < total_step; s++) c += step_counter[s * 2];
        stats[1] = (double)c;
    }
    if (n >= nbytes) return;
    const float4 a = reinterpret_cast<const float4*>(grid)[(size_t)n * 2];
    const float4 b = reinterpret_cast<const float4*>(grid)[(size_t)n * 2 + 1];
    uint32_t bits = 0;
    bits |= a.x > thresh ? 1u : 0u;  bits |= a.y > thresh ? 2u : 0u;
    bits |= a.z > thresh ? 4u : 0u;  bits |= a.w > thresh ? 8u : 0u;
    bits |= b.x > thresh ? 16u : 0u; bits |= b.y > thresh ? 32u : 0u;
    bits |= b.z > thresh ? 64u : 0u; bits |= b.w > thresh ? 128u : 0u;
    bitfield[n] = (uint8_t)bits;
}

// ---- mark_untrained_grid (nerf/renderer.py:408-469): a cell no training camera sees gets density -1 for good.
// One thread per (cascade, Morton cell): the cell centre is taken into every camera frame (c2w poses, so
// cam = (p - t) . R, i.e. R^T (p - t)) and tested against the frustum widened by two half-cells; the loop over the
// cameras stops at the first one that sees the cell (the reference counts them all and tests count == 0).
__global__ void __launch_bounds__(256) k_mark_untrained(const float* __restrict__ poses, uint32_t n_poses,
                                                        uint32_t pose_stride, float tan_x, float tan_y, Cascades cs,
                                                        uint32_t C, uint32_t H, float* __restrict__ grid) {
    const uint32_t H3 = H * H * H;
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= C * H3) return;
    const uint32_t cas = p / H3, idx = p - cas * H3;
    const float inv = 1.0f / (float)(H - 1);
    const float span = cs.span[cas], margin = cs.half[cas] * 2.0f;
    const float wx = (2.0f * (float)compact_bits(idx) * inv - 1.0f) * span;
    const float wy = (2.0f * (float)compact_bits(idx >> 1) * inv - 1.0f) * span;
    const float wz = (2.0f * (float)compact_bits(idx >> 2) * inv - 1.0f) * span;
    bool seen = false;
    for (uint32_t b = 0; b < n_poses && !seen; b++) {
        const float* m = poses + (size_t)b * pose_stride;          // row-major 4x4 (or 3x4): m[4 * row + col]
        const float dx = wx - m[3], dy = wy - m[7], dz = wz - m[11];
        const float cx = dx * m[0] + dy * m[4] + dz * m[8];
        const float cy = dx * m[1] + dy * m[5] + dz * m[9];
        const float cz = dx * m[2] + dy * m[6] + dz * m[10];
        seen = cz > 0.0f && fabsf(cx) < tan_x * cz + margin && fabsf(cy) < tan_y * cz + margin;
    }
    if (!seen) grid[p] = -1.0f;
}

bool grid_shape_ok(uint32_t C, uint32_t H) { return C >= 1 && C <= 8 && H >= 16 && H <= 512 && (H & (H - 1)) == 0; }

uint32_t log2u(uint32_t v) {
    uint32_t l = 0;
    while ((1u << l) < v) l++;
    return l;
}

size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

int enerf_density_grid_cells(const float* density_grid, uint32_t C, uint32_t H, float bound, uint32_t n_uniform,
                             uint64_t seed, int32_t* indices, float* xyzs, enerf_stream_t stream) {
    if (!grid_shape_ok(C, H)) ENERF_BADARG("density_grid_cells: C=%u H=%u (need 1..8 cascades, H a power of two in 16..512)", C, H);
    hipStream_t s = (hipStream_t)stream;
    const uint32_t logH = log2u(H), bits = 3 * logH;
    const uint32_t H3 = 1u << bits;
    const Cascades cs = make_cascades(C, H, bound);
    if (!density_grid) {
        const uint32_t P = C * H3;
        k_cells_full<<<div_up(P, 256), 256, 0, s>>>(cs, C, H, logH, seed, indices, xyzs);
        ENERF_LAUNCH_CHECK("density_grid_cells(full)");
        return 0;
    }
    if (n_uniform == 0) return 0;
    if ((uint64_t)C * 2 * n_uniform > 0x7fffffffULL) ENERF_BADARG("density_grid_cells: too many samples");
    const uint32_t P = C * 2 * n_uniform;
    const uint32_t blocks_per_cas = H3 / kOccBlock, nblocks = C * blocks_per_cas;
    // the sort's geometry: buckets of 2^low consecutive key values, at most kSortMaxBuckets of them
    uint32_t key_bits = bits + log2u(C);
    if ((1u << log2u(C)) < C) key_bits++;                                   // (C not a power of two)
    const uint32_t low = key_bits > 24 ? key_bits - 12 : 12;
    if (low > kSortMaxLow)
        ENERF_BADARG("density_grid_cells: the partial update sorts keys of up to 26 bits (C * H^3 <= 2^26 cells), got %u bits", key_bits);
    const uint32_t nb = div_up(C << bits, 1u << low);
    const size_t o_counts = 0, o_cas = o_counts + align256((size_t)nblocks * 4), o_list = o_cas + 256,
                 o_keys = o_list + align256((size_t)C * H3 * 4), o_sorted = o_keys + align256((size_t)P * 4),
                 o_tmp = o_sorted + align256((size_t)P * 4), o_sort = o_tmp + align256((size_t)P * 4),
                 total = o_sort + align256((size_t)3 * kSortMaxBuckets * 4);
    if (int ew = workspace_family_enter(1, s)) return ew;
    char* ws = (char*)workspace(WS_DENSITY, total);
    if (!ws) return ENERF_E_NOMEM;
    uint32_t* block_counts = (uint32_t*)(ws + o_counts);
    uint32_t* cas_counts = (uint32_t*)(ws + o_cas);
    uint32_t* list = (uint32_t*)(ws + o_list);
    uint32_t* keys = (uint32_t*)(ws + o_keys);
    uint32_t* sorted = (uint32_t*)(ws + o_sorted);
    k_occ_count<<<nblocks, 256, 0, s>>>(density_grid, block_counts);
    k_occ_scan<<<C, 1024, 0, s>>>(block_counts, blocks_per_cas, cas_counts);
    k_occ_compact<<<nblocks, 256, 0, s>>>(density_grid, block_counts, blocks_per_cas, H3, list);
    uint32_t* tmp = (uint32_t*)(ws + o_tmp);
    uint32_t* bucket_count = (uint32_t*)(ws + o_sort);             // [nb] sizes | [nb] cursors | [nb] bases
    uint32_t* bucket_cursor = bucket_count + kSortMaxBuckets;
    uint32_t* bucket_base = bucket_cursor + kSortMaxBuckets;
    k_select_keys<<<div_up(P > 2u * kSortMaxBuckets ? P : 2u * kSortMaxBuckets, 256), 256, 0, s>>>(
        C, n_uniform, bits, H3, seed, cas_counts, list, keys, bucket_count, 2u * kSortMaxBuckets);
    k_sort_count<<<div_up(P, kSortChunk), 256, 0, s>>>(keys, P, low, nb, bucket_count);
    k_sort_scan<<<1, 1024, 0, s>>>(bucket_count, nb, bucket_base);
    k_sort_scatter<<<div_up(P, kSortChunk), 256, 0, s>>>(keys, P, low, nb, bucket_base, bucket_cursor, tmp);
    k_sort_buckets<<<nb, 256, sizeof(uint32_t) << low, s>>>(tmp, low, bucket_count, bucket_base, sorted);
    k_cells_from_keys<<<div_up(P, 256), 256, 0, s>>>(cs, P, H, bits, seed, sorted, indices, xyzs);
    ENERF_LAUNCH_CHECK("density_grid_cells(partial)");
    return 0;
}

int enerf_mark_untrained_grid(const float* poses, uint32_t n_poses, uint32_t pose_stride, float fx, float fy, float cx,
                              float cy, uint32_t C, uint32_t H, float bound, float* density_grid,
                              enerf_stream_t stream) {
    if (!grid_shape_ok(C, H)) ENERF_BADARG("mark_untrained_grid: C=%u H=%u", C, H);
    if (pose_stride != 16 && pose_stride != 12) ENERF_BADARG("mark_untrained_grid: poses are [B,4,4] or [B,3,4] fp32");
    if (fx == 0.0f || fy == 0.0f) ENERF_BADARG("mark_untrained_grid: zero focal length");
    const uint32_t cells = C * H * H * H;
    k_mark_untrained<<<div_up(cells, 256), 256, 0, (hipStream_t)stream>>>(poses, n_poses, pose_stride, cx / fx, cy / fy,
                                                                          make_cascades(C, H, bound), C, H,
                                                                          density_grid);
    ENERF_LAUNCH_CHECK("mark_untrained_grid");
    return 0;
}

int enerf_density_grid_update(const int32_t* indices, const float* sigmas, uint32_t n_per_cascade, uint32_t C, uint32_t H,
                              float sigma_scale, float decay, float density_thresh, float* density_grid,
                              uint8_t* bitfield, const int32_t* step_counter, uint32_t total_step, double* stats,
                              enerf_stream_t stream) {
    if (!grid_shape_ok(C, H)) ENERF_BADARG("density_grid_update: C=%u H=%u", C, H);
    if (total_step > 16) ENERF_BADARG("density_grid_update: total_step %u > 16", total_step);
    if (!stats) ENERF_BADARG("density_grid_update: stats is required");
    hipStream_t s = (hipStream_t)stream;
    const uint32_t H3 = H * H * H;
    const size_t cells = (size_t)C * H3;
    if (int ew = workspace_family_enter(1, s)) return ew;
    char* ws = (char*)workspace(WS_DENSITY, kSumBytes + cells * 4);
    if (!ws) return ENERF_E_NOMEM;
    double* sum = (double*)ws;
    float* tmp = (float*)(ws + kSumBytes);
    // (cleared by a launch of our own: a 2 KiB hipMemsetAsync takes a fill kernel of the runtime's that nothing else in
    // the step uses, and its first use in a process cost the first update 15-23 ms on a fresh box)
    k_zero_sums<<<1, kSumSlots * kSumStride, 0, s>>>(sum);
    int e = check_hip(hipMemsetAsync(tmp, 0xFF, cells * 4, s), "density_grid_update: memset");   // NaN: "not evaluated"
    if (e) return e;
    const uint32_t P = n_per_cascade * C;
    if (!indices && n_per_cascade != H3) ENERF_BADARG("density_grid_update: indices may be NULL for a full sweep only");
    if (P) k_tmp_scatter<<<div_up(P, 256), 256, 0, s>>>(indices, sigmas, n_per_cascade, P, H3, sigma_scale, tmp, H, log2u(H));
    k_ema_mean<<<2048, 256, 0, s>>>(density_grid, tmp, (uint32_t)(cells / 4), decay, sum);
    const uint32_t nbytes = (uint32_t)(cells / 8);
    k_packbits_mean<<<div_up(nbytes, 256), 256, 0, s>>>(density_grid, nbytes, sum, 1.0 / (double)cells, density_thresh,
                                                        bitfield, step_counter, total_step, stats);
    ENERF_LAUNCH_CHECK("density_grid_update");
    return 0;
}

}  // extern "C"
