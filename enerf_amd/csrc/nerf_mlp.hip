// nerf_mlp.hip -- the two networks of nerf/network.py:104-132 (sigma net 32-64-16, colour net 31-64-64-out, nn.Linear
// without bias, ReLU) as ONE launch per direction on the split-bf16 matrix pipe (the arithmetic of mlp32s.hip: every fp32
// operand as bf16 hi + lo, three v_mfma_f32_32x32x16_bf16 products per fp32 product, fp32 accumulation).
//
// Why a second pair of kernels next to mlp32s.hip's one-net-per-launch ones: at the 4096-ray training batch (133 k
// samples, 4160 tiles over 1024 SIMDs) half of the five MLP launches' time did not depend on the batch at all
// (profiles/r05_base_step_timeline_rays64.txt: the same five launches on 64 rays take 8.8 + 9.3 + 16.7 + 13.2 + 4.6 us
// of the 17 + 16 + 37 + 26 + 6 us they take on 4096) -- kernel start, staging fp32 weights through LDS, every wave
// building every operand fragment from them (strided LDS gathers + hi / lo splits), the per-workgroup weight-gradient sums
// and their reduce launch.  Here
//   * the operand fragments are built ONCE per optimizer step by k_nerf_frags (44 wavefronts, 88 KB: forward order and
//     transposed order, hi and lo) and a kernel's set-up is a straight 16-byte-per-lane copy of them into LDS;
//   * the forward is one launch: x [16, Bp, 2] -> sigma net -> its 16 outputs stay in the accumulator registers and ARE
//     K-step 0 of the colour net's first layer (the SH basis, evaluated in registers from the directions, is K-step 1)
//     -> colour net -> sigma = exp(h0), rgb = sigmoid(.): the [B, 32] hand-over tensor is never written or read;
//   * the backward is one launch: both forwards recomputed (bit-identical instruction sequence), colour net dgrad +
//     wgrad, its input gradient is the sigma net's output gradient in registers (column 0 replaced by trunc_exp's
//     backward), sigma net dgrad + wgrad, dL/dx level-major: neither the [B, 32] input gradient nor the hand-over
//     tensor exist, one set-up, one set of per-workgroup sums.
// Operand layouts, the flips by the matrix pipe and the accuracy argument are those of mlp32s.hip (mlp32s_ops.h).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "mlp32_common.h"
#include "mlp32s_ops.h"

namespace enerf_mlp32 {

// ---- fragment table (one fragment = 64 lanes x 16 B of hi, then 64 lanes x 16 B of lo: 2 KiB)
// forward order (A operand: lane (j, h) = output row 32 ob + j, its 8 values = the K-step's contraction slots of half h)
constexpr int F_S0 = 0;      // [ob][t]      sigma first layer, level-major input order kmap<1>
constexpr int F_SO = 4;      // [ib][t]      sigma output layer (16 rows)
constexpr int F_C0 = 8;      // [ob][t]      colour first layer: K-step 0 = the sigma net's outputs, K-step 1 = SH
constexpr int F_CH = 12;     // [ob][ib][t]  colour hidden layer
constexpr int F_CO = 20;     // [ib][t]      colour output layer (out_c rows)      (forward only)
constexpr int NF_FWD = 24;
// transposed order (dgrad: lane = input index of the layer, values = output rows)
constexpr int B_COT = 24;    // [ib]         colour output layer, outputs in natural order 8h + e
constexpr int B_CHT = 26;    // [ib][ob][t]
constexpr int B_C0T = 34;    // [ob][t]      rows = the colour net's logical input columns [h0 | geo 15 | SH 16]
constexpr int B_SOT = 38;    // [ib]         sigma output layer, outputs in accumulator order nrow(e, h)
constexpr int B_S0T = 40;    // [ob][t]
constexpr int NF_ALL = 44;
constexpr int NF_BWD = 40;   // what the backward keeps in LDS: forward fragments 0..19 + transposed 24..43
constexpr uint32_t kFragWords = 512;      // dwords per fragment (hi + lo)

// sigma blob 64x32 + 16x64, colour blob 64x32 + 64x64 + out_c x 64 (out_c <= 16): per-workgroup partial sums
constexpr uint32_t NW_S = HID * IN + 16 * HID;
constexpr uint32_t NW_C_MAX = HID * IN + HID * HID + 16 * HID;
constexpr uint32_t P_STRIDE = NW_S + NW_C_MAX;

typedef unsigned u32x4n __attribute__((ext_vector_type(4)));

// colour net, first layer: memory column of logical column c ([SH 16 | geo 15 (| pad)] in memory, nerf/network.py:95)
__device__ __forceinline__ int c0_memcol(int c, uint32_t w0_cols) {
    if (c == 0) return -1;                             // the raw density: no weight
    const int m = c < 16 ? c + 15 : c - 16;
    return (uint32_t)m < w0_cols ? m : -1;
}

#ifndef NERF_MLP_BACKWARD_UNIT
// Where value e of lane (j, h) of fragment f comes from: tensor k = {ws0 [64,32], ws1 [16,64], wc0 [64,w0_cols], wc1 [64,64],
// wc2 [out_c,64]} in the upper half-word, the element's flat index in the lower; kFragZero = a padding zero.
constexpr uint32_t kFragZero = 0xffffffffu;
__device__ __forceinline__ uint32_t frag_source(int f, int j, int h, int e, uint32_t w0_cols, uint32_t out_c) {
    auto at = [](uint32_t k, int idx) { return (k << 16) | (uint32_t)idx; };
    if (f < F_SO) {
        const int ob = (f - F_S0) >> 1, t = f & 1;
        return at(0, (32 * ob + j) * IN + kmap<1>(8 * t + e, h));
    } else if (f < F_C0) {
        const int ib = (f - F_SO) >> 1, t = f & 1;
        return j < 16 ? at(1, j * HID + 32 * ib + nrow(8 * t + e, h)) : kFragZero;
    } else if (f < F_CH) {
        const int ob = (f - F_C0) >> 1, t = f & 1;
        const int m = c0_memcol(16 * t + nrow(e, h), w0_cols);
        return m < 0 ? kFragZero : at(2, (32 * ob + j) * (int)w0_cols + m);
    } else if (f < F_CO) {
        const int q = f - F_CH, ob = q >> 2, ib = (q >> 1) & 1, t = q & 1;
        return at(3, (32 * ob + j) * HID + 32 * ib + nrow(8 * t + e, h));
    } else if (f < B_COT) {
        const int ib = (f - F_CO) >> 1, t = f & 1;
        return (uint32_t)j < out_c ? at(4, j * HID + 32 * ib + nrow(8 * t + e, h)) : kFragZero;
    } else if (f < B_CHT) {
        const int ib = f - B_COT, o = 8 * h + e;
        return (uint32_t)o < out_c ? at(4, o * HID + 32 * ib + j) : kFragZero;
    } else if (f < B_C0T) {
        const int q = f - B_CHT, ib = q >> 2, ob = (q >> 1) & 1, t = q & 1;
        return at(3, (32 * ob + nrow(8 * t + e, h)) * HID + 32 * ib + j);
    } else if (f < B_SOT) {
        const int ob = (f - B_C0T) >> 1, t = f & 1;
        const int m = c0_memcol(j, w0_cols);
        return m < 0 ? kFragZero : at(2, (32 * ob + nrow(8 * t + e, h)) * (int)w0_cols + m);
    } else if (f < B_S0T) {
        const int ib = f - B_SOT;
        return at(1, nrow(e, h) * HID + 32 * ib + j);
    }
    const int ob = (f - B_S0T) >> 1, t = f & 1;
    return at(0, (32 * ob + nrow(8 * t + e, h)) * IN + j);
}

// One wavefront per fragment.
__global__ void __launch_bounds__(64) k_nerf_frags(const float* __restrict__ ws0, const float* __restrict__ ws1,
                                                  const float* __restrict__ wc0, const float* __restrict__ wc1,
                                                  const float* __restrict__ wc2, uint32_t w0_cols, uint32_t out_c,
                                                  uint32_t* __restrict__ frags) {
    const int f = blockIdx.x, lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const uint32_t src = frag_source(f, j, h, e, w0_cols, out_c);
        const uint32_t k = src >> 16, idx = src & 0xffffu;
        const float* w = k == 0 ? ws0 : k == 1 ? ws1 : k == 2 ? wc0 : k == 3 ? wc1 : wc2;
        v[e] = src == kFragZero ? 0.0f : w[idx];
    }
    const FragT<3> w = split8<3>(v);
    u32x4n* dst = reinterpret_cast<u32x4n*>(frags + (size_t)f * kFragWords);
    dst[lane] = __builtin_bit_cast(u32x4n, w.hi);
    dst[64 + lane] = __builtin_bit_cast(u32x4n, w.lo);
}

// The same as a table: map[(f * 64 + lane) * 8 + e] = frag_source(...) -- what a kernel that knows nothing of the layout
// needs to build the fragments (enerf_nerf_mlp_frag_job: the grid forward of the one-call training step carries the build).
__global__ void __launch_bounds__(64) k_nerf_frag_map(uint32_t w0_cols, uint32_t out_c, uint32_t* __restrict__ map) {
    const int f = blockIdx.x, lane = threadIdx.x, j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int e = 0; e < 8; e++) map[((size_t)f * 64 + lane) * 8 + e] = frag_source(f, j, h, e, w0_cols, out_c);
}

#endif
struct NerfRows {                   // enerf_mlp32_valid_rows(_ex): see WSrc
    const int32_t* valid_rows;
    uint32_t valid_base, valid_cap;
};
__device__ __forceinline__ uint32_t nerf_valid_tiles(const NerfRows& r, uint32_t B, uint32_t ntiles) {
    WSrc w;
    w.valid_rows = r.valid_rows;
    w.valid_base = r.valid_base;
    w.valid_cap = r.valid_cap;
    return valid_tiles(w, B, ntiles);
}

// the 16 SH values of a direction, those of the contraction slots of lane half h: slot e <-> component nrow(e, h)
__device__ __forceinline__ void sh_slots(float d0, float d1, float d2, const ShNorm4& nrm, int h, float (&v)[8]) {
    float sh[16];
    sh4(d0, d1, d2, nrm, sh);
    const uint32_t m = 0u - (uint32_t)h;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int i0 = (e & 3) + 8 * (e >> 2);
        v[e] = __uint_as_float((__float_as_uint(sh[i0]) & ~m) | (__float_as_uint(sh[i0 + 4]) & m));
    }
}

__device__ __forceinline__ void relu_tile(f32x16& a) {
#pragma unroll
    for (int q = 0; q < 16; q++) a[q] = __int_as_float(max(__float_as_int(a[q]), 0));
}

// `COUNT` fragments of the global table -> LDS slots; every load of a thread is issued before its first LDS store (the trip
// count is a compile-time constant: as a plain load / store loop the copy cost one memory latency per iteration, twelve of
// them in the forward's set-up)
template <int COUNT>
__device__ __forceinline__ void copy_frags(u32x4n* __restrict__ fr, const uint32_t* __restrict__ frags, int first, int slot0) {
    const u32x4n* src = reinterpret_cast<const u32x4n*>(frags + (size_t)first * kFragWords);
    u32x4n* dst = fr + slot0 * 128;
    constexpr int N = COUNT * 128 / 256;
    u32x4n v[N];
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = src[threadIdx.x + k * 256];
#pragma unroll
    for (int k = 0; k < N; k++) dst[threadIdx.x + k * 256] = v[k];
}

// Residency.  Rounds 4-5 shipped this kernel with ONE workgroup per CU (84 KiB of LDS, the SIMD's whole register file
// claimed): with three workgroups per CU the colour outputs of samples 16..31 of a tile came out wrong in a few per cent
// of the launches (10^-3..10^-2, different rows every launch, sigma never).  The cause is the operand hazard described in
// mlp32s_ops.h (operand_ready): the conversions that make an MFMA's 16-bit operand were not yet complete in lanes 16..31
// / 48..63 when the matrix pipe read it -- only when three wavefronts' conversions share a SIMD.  With the barrier in
// split8 / exact8 the kernel is bit-stable at any residency (tools/nerf_fwd_residency.py: 0 of 400 launches at three
// workgroups per CU, padded and unpadded builds; tests/test_gpu_mlp32.py soaks it), and the forward shares the CU again:
// 48 KiB of LDS, ~136 registers, three workgroups per CU = three wavefronts per SIMD hiding each other's MFMA and LDS
// latencies.  The backward keeps a CU to itself for what it needs (149 KiB of LDS, ~480 registers).
// (-DNERF_FWD_ONE_PER_CU: the forward as shipped in round 5's first half, for A/B runs.)
#ifndef NERF_FWD_ONE_PER_CU
#define NERF_FWD_WHOLE_SIMD() do {} while (0)
constexpr uint32_t kFwdLdsWords = 12 * 1024;              // 48 KiB: the fragments alone, three workgroups per CU
#else
#define NERF_FWD_WHOLE_SIMD() asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a255, 0" ::: "v255", "a255")
constexpr uint32_t kFwdLdsWords = 21 * 1024;              // 84 KiB: two workgroups do not fit a CU's 160 KiB
#endif
// the backward's wave allocation is the SIMD's whole register file (256 + 256): it needs most of it anyway
#define NERF_WHOLE_SIMD() asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a255, 0" ::: "v255", "a255")
#ifndef NERF_MLP_BACKWARD_UNIT
// ---------------------------------------------------------------------------------------------------- forward
__global__ void __launch_bounds__(256) k_nerf_fwd(const float* __restrict__ X, const float* __restrict__ dirs,
                                                   const uint32_t* __restrict__ frags, float* __restrict__ sigma,
                                                   float* __restrict__ rgb, uint32_t B, uint32_t out_c, NerfRows rows,
                                                   ShNorm4 nrm) {
    static_assert(NF_FWD * kFragWords <= kFwdLdsWords, "LDS");
    __shared__ __attribute__((aligned(16))) uint32_t lds[kFwdLdsWords];
    NERF_FWD_WHOLE_SIMD();
    const int lane = lane_id(), j = lane & 31, h = lane >> 5;
    const uint32_t Bp = (B + 31u) & ~31u;
    const uint32_t gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t nw = gridDim.x * 4;
    float x[16];
    if (gw < Bp / 32) load_x<1>(X, gw, j, h, B, Bp, x);           // (travels during the set-up)
    u32x4n* fr = reinterpret_cast<u32x4n*>(lds);
    typedef FragT<3> Frag;
    copy_frags<NF_FWD>(fr, frags, 0, 0);
    __syncthreads();
    auto get = [&](int f) -> Frag {
        Frag w;
        w.hi = __builtin_bit_cast(bf16x8, fr[f * 128 + lane]);
        w.lo = __builtin_bit_cast(bf16x8, fr[f * 128 + 64 + lane]);
        return w;
    };
    const uint32_t ntiles = nerf_valid_tiles(rows, B, Bp / 32);
    for (uint32_t tile = gw; tile < ntiles; tile += nw) {
        const size_t s = (size_t)tile * 32 + j;
        const bool valid = s < B;
        asm volatile("" ::: "memory");                    // the operand reads stay inside the loop
        if (tile != gw) load_x<1>(X, tile, j, h, B, Bp, x);
        float d0 = 0.0f, d1 = 0.0f, d2 = 1.0f;
        if (valid) {
            d0 = dirs[s * 3]; d1 = dirs[s * 3 + 1]; d2 = dirs[s * 3 + 2];
        }
        Frag xf[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = x[8 * t + e];
            xf[t] = split8<3>(v);
        }
        // ---- sigma net
        Frag af[2][2];
#pragma unroll
        for (int ob = 0; ob < 2; ob++) {
            f32x16 a = (f32x16)(0.0f);
#pragma unroll
            for (int t = 0; t < 2; t++) a = mmap(get(F_S0 + 2 * ob + t), xf[t], a);
            relu_tile(a);
            split_tile<3>(a, af[ob]);
        }
        f32x16 os = (f32x16)(0.0f);
#pragma unroll
        for (int ib = 0; ib < 2; ib++)
#pragma unroll
            for (int t = 0; t < 2; t++) os = mmap(get(F_SO + 2 * ib + t), af[ib][t], os);
        if (valid && h == 0) sigma[s] = expf(os[0]);
        // ---- colour net: K-step 0 = the sigma net's outputs (registers 0..7: outputs nrow(e, h)), K-step 1 = SH
        Frag in[2];
        {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = os[e];
            in[0] = split8<3>(v);
            sh_slots(d0, d1, d2, nrm, h, v);
            in[1] = split8<3>(v);
        }
#pragma unroll
        for (int ob = 0; ob < 2; ob++) {
            f32x16 a = (f32x16)(0.0f);
#pragma unroll
            for (int t = 0; t < 2; t++) a = mmap(get(F_C0 + 2 * ob + t), in[t], a);
            relu_tile(a);
            split_tile<3>(a, af[ob]);
        }
        Frag bf[2][2];
#pragma unroll
        for (int ob = 0; ob < 2; ob++) {
            f32x16 n = (f32x16)(0.0f);
#pragma unroll
            for (int ib = 0; ib < 2; ib++)
#pragma unroll
                for (int t = 0; t < 2; t++) n = mmap(get(F_CH + (ob * 2 + ib) * 2 + t), af[ib][t], n);
            relu_tile(n);
            split_tile<3>(n, bf[ob]);
        }
        f32x16 o = (f32x16)(0.0f);
#pragma unroll
        for (int ib = 0; ib < 2; ib++)
#pragma unroll
            for (int t = 0; t < 2; t++) o = mmap(get(F_CO + 2 * ib + t), bf[ib][t], o);
        // outputs r < out_c sit in the accumulator registers of lane half r / 4 mod 2: r = nrow(q, h)
        if (valid) {
            if (h == 0) {
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if ((uint32_t)q < out_c) rgb[s * out_c + q] = out_act_fwd(o[q], 3);
            }
            if (out_c > 4) {
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const uint32_t r = (uint32_t)nrow(q, h);
                    if (r >= 4 && r < out_c) rgb[s * out_c + r] = out_act_fwd(o[q], 3);
                }
            }
        }
    }
}

#endif
#ifdef NERF_MLP_BACKWARD_UNIT
// ---------------------------------------------------------------------------------------------------- backward
struct NerfBwdArgs {
    const float* X;            // [16, Bp, 2]
    const float* dirs;         // [B, 3]
    const float* g_rgb;        // [B, out_c]   dL/d rgb (after the sigmoid)
    const float* rgb;          // [B, out_c]   the forward's rgb
    const float* g_sigma;      // [B]          dL/d sigma (after trunc_exp)
    float sigma_scale;         // multiplies g_sigma (the renderer's density_scale)
    float* dX;                 // [16, Bp, 2]
    float* partial;            // [gridDim.x][P_STRIDE]
    uint32_t B, out_c;
    NerfRows rows;
};

// Register budget: the twelve weight-gradient accumulators (192 registers) live in the AGPR half of the file; the arch
// half holds a tile's working set, which therefore must stay small: the selection matrices of the flips sit in LDS, and the
// two hidden activations that are needed again much later (sigma net hidden layer, colour net first layer) are stashed in
// LDS as the operand halves they were split into (a private 16 KiB per wave: no fence, no barrier) -- the ReLU mask of the
// backward is read off the stashed hi half (an activation is positive iff its bf16 rounding is non-zero).
constexpr uint32_t kSelWords = 5 * 256;                                  // five selection matrices, 64 lanes x 16 B
constexpr uint32_t kStashWords = 8 * 512;                                // per wave: 2 layers x [2 blocks][2 K-steps] fragments
constexpr uint32_t kBwdLdsWords = NF_BWD * kFragWords + kSelWords + 4 * kStashWords;

// g[q] = (element q of the tile whose K-step fragments are f[0], f[1] is non-zero) ? g[q] : 0   (hi halves: bf16 pairs)
__device__ __forceinline__ void mask_by_frag(f32x16& g, const FragT<3> (&f)[2]) {
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const i32x4 w = __builtin_bit_cast(i32x4, f[t].hi);
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const uint32_t u = (uint32_t)w[p];
            g[8 * t + 2 * p] = (u & 0xffffu) != 0u ? g[8 * t + 2 * p] : 0.0f;
            g[8 * t + 2 * p + 1] = u > 0xffffu ? g[8 * t + 2 * p + 1] : 0.0f;
        }
    }
}

__global__ void __launch_bounds__(256) k_nerf_bwd(NerfBwdArgs a, const uint32_t* __restrict__ frags, ShNorm4 nrm) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[kBwdLdsWords];       // operands (then the dW sums), selectors, stashes
    static_assert(NF_BWD * kFragWords >= 2 * P_STRIDE, "the two sum regions reuse the operand area");
    static_assert(kBwdLdsWords * 4 <= 160 * 1024, "LDS");
#if defined(MLP32S_NO_OPERAND_BARRIER)
    // nerf_mlp_bwd.hip compiles this kernel without the operand barrier: sound only while ONE workgroup of it fits a CU,
    // i.e. one wavefront per SIMD (mlp32s_ops.h: operand_ready).  More than half of the CU's 160 KiB of LDS guarantees it.
    static_assert(kBwdLdsWords * 4 > 80 * 1024, "k_nerf_bwd without the operand barrier must stay at one workgroup per CU");
#endif
    NERF_WHOLE_SIMD();
    typedef FragT<3> Frag;
    const int lane = lane_id(), j = lane & 31, h = lane >> 5;
    const int wid = threadIdx.x >> 6;
    const uint32_t B = a.B, out_c = a.out_c;
    const uint32_t Bp = (B + 31u) & ~31u;
    const uint32_t ntiles = Bp / 32;
    const uint32_t nreal = nerf_valid_tiles(a.rows, B, ntiles);
    const uint32_t gw = blockIdx.x * 4 + wid;
    const uint32_t nw = gridDim.x * 4;

    // everything a tile reads from global memory is requested for the wave's NEXT tile where the current tile has used it
    // for the last time, into the same registers
    float x[16], dy_raw[4], ys_raw[4], ds_raw = 0.0f, d0 = 0.0f, d1 = 0.0f, d2 = 1.0f;
    auto request_x = [&](uint32_t t) { load_x<1>(a.X, t, j, h, B, Bp, x); };
    auto request_out = [&](uint32_t t) {
        const size_t sn = (size_t)t * 32 + j;
        const size_t sc = sn < B ? sn : (size_t)B - 1;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t oc = (uint32_t)e < out_c ? (uint32_t)e : out_c - 1;
            dy_raw[e] = a.g_rgb[sc * out_c + oc];
            ys_raw[e] = a.rgb[sc * out_c + oc];
        }
        ds_raw = a.g_sigma[sc];
        d0 = a.dirs[sc * 3]; d1 = a.dirs[sc * 3 + 1]; d2 = a.dirs[sc * 3 + 2];
    };
    {
        const uint32_t t0 = gw < ntiles ? gw : ntiles - 1;
        request_x(t0);
        request_out(t0);
    }
    u32x4n* fr = reinterpret_cast<u32x4n*>(lds);
    u32x4n* sel = fr + NF_BWD * 128;                     // [kind][lane]
    u32x4n* stash = sel + 5 * 64 + wid * (kStashWords / 4);      // this wave's: [layer][block][K-step][hi, lo][lane]
    copy_frags<20>(fr, frags, 0, 0);
    copy_frags<20>(fr, frags, B_COT, 20);
    if (wid == 0) {
        // selection matrices of the flips: natural order (dL/dY), a tile's two register halves, and X^T (K-step t of the
        // input operand holds input columns kmap<1>(8t + e, h))
        sel[0 * 64 + lane] = __builtin_bit_cast(u32x4n, selector(j, h, 0));
        sel[1 * 64 + lane] = __builtin_bit_cast(u32x4n, selector(j, h, 1));
        sel[2 * 64 + lane] = __builtin_bit_cast(u32x4n, selector(j, h, 2));
#pragma unroll
        for (int t = 0; t < 2; t++) {
            bf16x8 sx;
#pragma unroll
            for (int e = 0; e < 8; e++) sx[e] = kmap<1>(8 * t + e, h) == j ? (elem16)1.0f : (elem16)0.0f;
            sel[(3 + t) * 64 + lane] = __builtin_bit_cast(u32x4n, sx);
        }
    }
    __syncthreads();
    auto get = [&](int f) -> Frag {                      // f: index in the global table
        const int slot = f < 20 ? f : f - 4;
        Frag w;
        w.hi = __builtin_bit_cast(bf16x8, fr[slot * 128 + lane]);
        w.lo = __builtin_bit_cast(bf16x8, fr[slot * 128 + 64 + lane]);
        return w;
    };
    auto SEL = [&](int k) -> bf16x8 { return __builtin_bit_cast(bf16x8, sel[k * 64 + lane]); };
    auto put_stash = [&](int layer, int ib, const Frag (&f)[2]) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
            u32x4n* p = stash + (((layer * 2 + ib) * 2 + t) * 2) * 64 + lane;
            p[0] = __builtin_bit_cast(u32x4n, f[t].hi);
            p[64] = __builtin_bit_cast(u32x4n, f[t].lo);
        }
    };
    auto get_stash = [&](int layer, int ib, Frag (&f)[2]) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const u32x4n* p = stash + (((layer * 2 + ib) * 2 + t) * 2) * 64 + lane;
            f[t].hi = __builtin_bit_cast(bf16x8, p[0]);
            f[t].lo = __builtin_bit_cast(bf16x8, p[64]);
        }
    };

    f32x16 aw0S[2], awoS[2], aw0C[2], awhC[2][2], awoC[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        aw0S[k] = (f32x16)(0.0f); awoS[k] = (f32x16)(0.0f); aw0C[k] = (f32x16)(0.0f); awoC[k] = (f32x16)(0.0f);
        awhC[k][0] = (f32x16)(0.0f); awhC[k][1] = (f32x16)(0.0f);
    }

    for (uint32_t tile = gw; tile < ntiles; tile += nw) {
        const uint32_t tnext = tile + nw < nreal ? tile + nw : tile;
        const size_t s = (size_t)tile * 32 + j;
        const bool valid = s < B;
        asm volatile("" ::: "memory");
        if (tile >= nreal) {                             // the sample budget's padding: zero input gradients, nothing else
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                const size_t lv = (size_t)(4 * gq + 2 * h);
                *reinterpret_cast<float2*>(a.dX + (lv * Bp + s) * 2) = make_float2(0.f, 0.f);
                *reinterpret_cast<float2*>(a.dX + ((lv + 1) * Bp + s) * 2) = make_float2(0.f, 0.f);
            }
            continue;
        }
        // ---- both forwards again, instruction for instruction (k_nerf_fwd)
        Frag xop[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = x[8 * t + e];
            xop[t] = split8<3>(v);
        }
        request_x(tnext);
        f32x16 os = (f32x16)(0.0f);
        {
            Frag af[2][2];
#pragma unroll
            for (int ob = 0; ob < 2; ob++) {
                f32x16 t0 = (f32x16)(0.0f);
#pragma unroll
                for (int t = 0; t < 2; t++) t0 = mmap(get(F_S0 + 2 * ob + t), xop[t], t0);
                relu_tile(t0);
                split_tile<3>(t0, af[ob]);
                put_stash(0, ob, af[ob]);
            }
#pragma unroll
            for (int ib = 0; ib < 2; ib++)
#pragma unroll
                for (int t = 0; t < 2; t++) os = mmap(get(F_SO + 2 * ib + t), af[ib][t], os);
        }
        const float h0 = os[0];                          // (lanes of half 0: the raw density)
        Frag inop[2];
        {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = os[e];
            inop[0] = split8<3>(v);
            sh_slots(valid ? d0 : 0.0f, valid ? d1 : 0.0f, valid ? d2 : 1.0f, nrm, h, v);
            inop[1] = split8<3>(v);
        }
        Frag c1f[2][2];                                  // the colour net's second hidden layer, as operands
        {
            Frag af[2][2];
#pragma unroll
            for (int ob = 0; ob < 2; ob++) {
                f32x16 t0 = (f32x16)(0.0f);
#pragma unroll
                for (int t = 0; t < 2; t++) t0 = mmap(get(F_C0 + 2 * ob + t), inop[t], t0);
                relu_tile(t0);
                split_tile<3>(t0, af[ob]);
                put_stash(1, ob, af[ob]);
            }
#pragma unroll
            for (int ob = 0; ob < 2; ob++) {
                f32x16 n = (f32x16)(0.0f);
#pragma unroll
                for (int ib = 0; ib < 2; ib++)
#pragma unroll
                    for (int t = 0; t < 2; t++) n = mmap(get(F_CH + (ob * 2 + ib) * 2 + t), af[ib][t], n);
                relu_tile(n);
                split_tile<3>(n, c1f[ob]);
            }
        }
        // ---- colour net, output layer: dL/dY = (dY (1 - y)) y  (torch's sigmoid_backward), outputs in natural order
        Frag dyf;
        {
            float dy[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                float gq = 0.0f;
                if (e < 4) gq = (dy_raw[e] * (1.0f - ys_raw[e])) * ys_raw[e];
                dy[e] = (valid && h == 0 && (uint32_t)e < out_c) ? gq : 0.0f;
            }
            if (out_c > 4) {                             // (wider colour outputs: not prefetched, read here)
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const uint32_t o = (uint32_t)(8 * h + e);
                    if (o >= 4 && o < out_c && valid) {
                        const float y = a.rgb[s * out_c + o];
                        dy[e] = (a.g_rgb[s * out_c + o] * (1.0f - y)) * y;
                    }
                }
            }
            dyf = split8<3>(dy);
        }
        const float dsig = valid ? (ds_raw * a.sigma_scale) * expf(fminf(fmaxf(h0, -15.0f), 15.0f)) : 0.0f;
        request_out(tnext);
        Frag gf[2][2], gT[2][2];                         // a layer's output gradient: as operands, and flipped
        {
            Frag dyT[2];
            flip_natural<3>(dyf, SEL(0), dyT);
#pragma unroll
            for (int ib = 0; ib < 2; ib++) {
                f32x16 g = mmap(get(B_COT + ib), dyf, (f32x16)(0.0f));
                mask_by_frag(g, c1f[ib]);
                Frag ft[2];
                flip_tile<3>(c1f[ib], SEL(1), SEL(2), ft);
#pragma unroll
                for (int t = 0; t < 2; t++) awoC[ib] = mmap(dyT[t], ft[t], awoC[ib]);
                split_tile<3>(g, gf[ib]);
            }
#pragma unroll
            for (int ib = 0; ib < 2; ib++) flip_tile<3>(gf[ib], SEL(1), SEL(2), gT[ib]);
        }
        // ---- colour net, hidden layer
        {
            Frag g0f[2][2];
#pragma unroll
            for (int ib = 0; ib < 2; ib++) {
                f32x16 n = (f32x16)(0.0f);
#pragma unroll
                for (int ob = 0; ob < 2; ob++)
#pragma unroll
                    for (int t = 0; t < 2; t++) n = mmap(get(B_CHT + (ib * 2 + ob) * 2 + t), gf[ob][t], n);
                Frag c0f[2], ft[2];
                get_stash(1, ib, c0f);
                mask_by_frag(n, c0f);
                flip_tile<3>(c0f, SEL(1), SEL(2), ft);
#pragma unroll
                for (int ob = 0; ob < 2; ob++)
#pragma unroll
                    for (int t = 0; t < 2; t++) awhC[ob][ib] = mmap(gT[ob][t], ft[t], awhC[ob][ib]);
                split_tile<3>(n, g0f[ib]);
            }
#pragma unroll
            for (int ib = 0; ib < 2; ib++) {
                gf[ib][0] = g0f[ib][0];
                gf[ib][1] = g0f[ib][1];
                flip_tile<3>(gf[ib], SEL(1), SEL(2), gT[ib]);
            }
        }
        // ---- colour net, input layer: its input's transpose is a flip like any other (K-step 0: columns nrow(e, h),
        // K-step 1: 16 + nrow(e, h) -- the selection matrices of a tile's two register halves)
        Frag dysf;
        {
            Frag inT[2];
            flip_tile<3>(inop, SEL(1), SEL(2), inT);
#pragma unroll
            for (int ob = 0; ob < 2; ob++)
#pragma unroll
                for (int t = 0; t < 2; t++) aw0C[ob] = mmap(gT[ob][t], inT[t], aw0C[ob]);
            f32x16 d = (f32x16)(0.0f);
#pragma unroll
            for (int ob = 0; ob < 2; ob++)
#pragma unroll
                for (int t = 0; t < 2; t++) d = mmap(get(B_C0T + 2 * ob + t), gf[ob][t], d);
            // rows 0..15 of d (registers 0..7) = dL/d(sigma net outputs nrow(e, h)); output 0 went through trunc_exp
            float dys[8];
#pragma unroll
            for (int e = 0; e < 8; e++) dys[e] = d[e];
            if (h == 0) dys[0] = dsig;
            dysf = split8<3>(dys);
        }
        // ---- sigma net, output layer
        {
            Frag dyT[2];
            flip_natural<3>(dysf, SEL(1), dyT);          // (slots in accumulator order: the first half's selection matrix)
#pragma unroll
            for (int ib = 0; ib < 2; ib++) {
                f32x16 g = mmap(get(B_SOT + ib), dysf, (f32x16)(0.0f));
                Frag asf[2], ft[2];
                get_stash(0, ib, asf);
                mask_by_frag(g, asf);
                flip_tile<3>(asf, SEL(1), SEL(2), ft);
#pragma unroll
                for (int t = 0; t < 2; t++) awoS[ib] = mmap(dyT[t], ft[t], awoS[ib]);
                split_tile<3>(g, gf[ib]);
            }
#pragma unroll
            for (int ib = 0; ib < 2; ib++) flip_tile<3>(gf[ib], SEL(1), SEL(2), gT[ib]);
        }
        // ---- sigma net, input layer
        {
            Frag xT[2];
            flip_tile<3>(xop, SEL(3), SEL(4), xT);
#pragma unroll
            for (int ob = 0; ob < 2; ob++)
#pragma unroll
                for (int t = 0; t < 2; t++) aw0S[ob] = mmap(gT[ob][t], xT[t], aw0S[ob]);
            f32x16 d = (f32x16)(0.0f);
#pragma unroll
            for (int ob = 0; ob < 2; ob++)
#pragma unroll
                for (int t = 0; t < 2; t++) d = mmap(get(B_S0T + 2 * ob + t), gf[ob][t], d);
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                const size_t lv = (size_t)(4 * gq + 2 * h);
                *reinterpret_cast<float2*>(a.dX + (lv * Bp + s) * 2) = make_float2(d[4 * gq], d[4 * gq + 1]);
                *reinterpret_cast<float2*>(a.dX + ((lv + 1) * Bp + s) * 2) = make_float2(d[4 * gq + 2], d[4 * gq + 3]);
            }
        }
    }

    // per-workgroup sums in a fixed order (deterministic): waves 0 / 1 write their accumulators into a region each, waves
    // 2 / 3 add into the same regions, then (w0 + w2) + (w1 + w3) leaves as this workgroup's partial sum
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds) + (size_t)(wid & 1) * P_STRIDE;
    const uint32_t NW_C = HID * IN + HID * HID + out_c * HID;
    auto flush = [&](const f32x16& acc, uint32_t base, int ld, int ob, int nb, uint32_t nrows, bool add) {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const uint32_t o = (uint32_t)(32 * ob + nrow(q, h));
            if (o < nrows) {
                float* p = red + base + o * ld + 32 * nb + j;
                *p = add ? *p + acc[q] : acc[q];
            }
        }
    };
    auto flush_all = [&](bool add) {
#pragma unroll
        for (int ob = 0; ob < 2; ob++) flush(aw0S[ob], 0, IN, ob, 0, HID, add);
#pragma unroll
        for (int nb = 0; nb < 2; nb++) flush(awoS[nb], HID * IN, HID, 0, nb, 16, add);
#pragma unroll
        for (int ob = 0; ob < 2; ob++) flush(aw0C[ob], NW_S, IN, ob, 0, HID, add);
#pragma unroll
        for (int ob = 0; ob < 2; ob++)
#pragma unroll
            for (int nb = 0; nb < 2; nb++) flush(awhC[ob][nb], NW_S + HID * IN, HID, ob, nb, HID, add);
#pragma unroll
        for (int nb = 0; nb < 2; nb++) flush(awoC[nb], NW_S + HID * IN + HID * HID, HID, 0, nb, out_c, add);
    };
    if (wid < 2) flush_all(false);
    __syncthreads();
    if (wid >= 2) flush_all(true);
    __syncthreads();
    float* dst = a.partial + (size_t)blockIdx.x * P_STRIDE;
    const float* r0 = reinterpret_cast<const float*>(lds);
    for (uint32_t i = threadIdx.x; i < NW_S + NW_C; i += blockDim.x) dst[i] = r0[i] + r0[P_STRIDE + i];
}

#endif
// ---------------------------------------------------------------------------------------------------- launchers
#ifdef NERF_MLP_BACKWARD_UNIT
#define k_nerf_mark k_nerf_mark_bwd_unit          // (one marker kernel per translation unit)
#endif
__global__ void k_nerf_mark() {}

#ifndef NERF_MLP_BACKWARD_UNIT
void nerf_launch_frags(const float* ws0, const float* ws1, const float* wc0, const float* wc1, const float* wc2,
                       uint32_t w0_cols, uint32_t out_c, uint32_t* frags, hipStream_t s) {
    hipLaunchKernelGGL(k_nerf_frags, dim3(NF_ALL), dim3(64), 0, s, ws0, ws1, wc0, wc1, wc2, w0_cols, out_c, frags);
}

void nerf_launch_frag_map(uint32_t w0_cols, uint32_t out_c, uint32_t* map, hipStream_t s) {
    hipLaunchKernelGGL(k_nerf_frag_map, dim3(NF_ALL), dim3(64), 0, s, w0_cols, out_c, map);
}

void nerf_launch_fwd(const float* X, const float* dirs, const uint32_t* frags, float* sigma, float* rgb, uint32_t B,
                     uint32_t out_c, const int32_t* valid_rows, uint32_t valid_base, uint32_t valid_cap, uint32_t grid,
                     hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (ev_start) hipExtLaunchKernelGGL(k_nerf_mark, dim3(1), dim3(64), 0, s, nullptr, ev_start, 0);
    hipExtLaunchKernelGGL(k_nerf_fwd, dim3(grid), dim3(256), 0, s, nullptr, ev_stop, 0, X, dirs, frags, sigma, rgb, B, out_c,
                          NerfRows{valid_rows, valid_base, valid_cap}, make_sh_norm4());
}

#else
void nerf_launch_bwd(const float* X, const float* dirs, const float* g_rgb, const float* rgb, const float* g_sigma,
                     float sigma_scale, const uint32_t* frags, float* dX, float* partial, uint32_t B, uint32_t out_c,
                     const int32_t* valid_rows, uint32_t valid_base, uint32_t valid_cap, uint32_t grid, hipStream_t s,
                     hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (ev_start) hipExtLaunchKernelGGL(k_nerf_mark, dim3(1), dim3(64), 0, s, nullptr, ev_start, 0);
    NerfBwdArgs a;
    a.X = X; a.dirs = dirs; a.g_rgb = g_rgb; a.rgb = rgb; a.g_sigma = g_sigma; a.sigma_scale = sigma_scale;
    a.dX = dX; a.partial = partial; a.B = B; a.out_c = out_c;
    a.rows = NerfRows{valid_rows, valid_base, valid_cap};
    hipExtLaunchKernelGGL(k_nerf_bwd, dim3(grid), dim3(256), 0, s, nullptr, ev_stop, 0, a, frags, make_sh_norm4());
}
#endif

}  // namespace enerf_mlp32
