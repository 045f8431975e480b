// ffmlp.hip -- fully fused 64-wide MLP on the CDNA4 matrix cores (bf16 or fp16 storage, fp32 accumulation).
//
// Replaces ffmlp/src/ffmlp.cu (+ utils.h, cutlass_matmul.h and the CUTLASS split-K GEMMs) of the reference:
// ffmlp_forward / ffmlp_inference / ffmlp_backward / allocate_splitk / free_splitk.
//
// MI355X mapping (v_mfma_f32_32x32x16_{bf16,f16}, wave64)
//  * Everything is computed "transposed": D[neuron i][sample j] = W[i][k] * X^T[k][j].  The MFMA B operand then wants,
//    per lane, 8 consecutive k of ONE sample -- exactly a 16-byte load from the row-major [B, in] input -- and the
//    D tile (lane = sample, registers = 16 neurons) of one layer IS the B operand of the next layer once the
//    contraction index is permuted consistently on the weight side.  No shuffles, no LDS traffic between layers.
//  * A wavefront owns 32 samples at a time and keeps every layer's weight fragments in registers (64..96 VGPRs for
//    the sigma / colour nets); the blob is staged once per workgroup through LDS.  Workgroups are persistent and
//    grid-stride over 32-sample tiles.
//  * Backward = dgrad chain in the same orientation (writes backward_buffer / grad_inputs), then a weight-gradient
//    kernel that re-reads the [B,64] activation / gradient buffers, turns each 32x32 tile into the "lane = neuron,
//    registers = samples" orientation with an MFMA against the identity (exact), and accumulates dW tiles in registers
//    over the whole batch slab; per-workgroup partial sums go to a workspace and are reduced by a final pass
//    (deterministic; replaces the reference's per-layer split-K CUTLASS GEMMs on side streams).
//  * fp32 accumulation everywhere (the reference accumulates in fp16).
#include <hip/hip_runtime.h>

#include "common.h"

using namespace enerf;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int HID = 64;
constexpr int OUT = 16;
constexpr int MAX_NL = 4;


template <typename E> struct V;
template <> struct V<__bf16> { using x8 = bf16x8; using x4 = bf16x4; };
template <> struct V<_Float16> { using x8 = f16x8; using x4 = f16x4; };

__device__ __forceinline__ f32x16 mma(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

#define K_ACT 10.0f
__device__ __forceinline__ float act_fwd(float x, uint32_t a) {   // ffmlp/src/utils.h:424-470
    switch (a) {
        case 0: return x > 0 ? x : 0.0f;
        case 1: return __expf(x);
        case 2: return __sinf(x);
        case 3: return 1.0f / (1.0f + __expf(-x));
        case 4: { const float v = x * K_ACT; return 0.5f * (v + sqrtf(v * v + 4)) / K_ACT; }
        case 5: return __logf(__expf(x * K_ACT) + 1.0f) / K_ACT;
        default: return x;
    }
}
__device__ __forceinline__ float act_bwd(float g, float fwd, uint32_t a) {   // utils.h:534-582 (post-activation input)
    switch (a) {
        case 0: return fwd > 0 ? g : 0.0f;
        case 1: return g * fwd;
        case 3: return g * (fwd * (1.0f - fwd));
        case 4: { const float y = fwd * K_ACT; return g * (y * y / (y * y + 1)); }
        case 5: return g * (1.0f - __expf(-fwd * K_ACT));
        default: return g;
    }
}

// Whole-tile activation with the switch OUTSIDE the element loop: the relu / none paths used by the NeRF nets are a
// handful of instructions; the transcendental variants live in their own (cold) blocks instead of being expanded and
// branched over per element.
__device__ __forceinline__ void apply_act(f32x16& t, uint32_t a) {
    if (a == 0) {
#pragma unroll
        for (int q = 0; q < 16; q++) t[q] = fmaxf(t[q], 0.0f);
    } else if (a != 6) {
#define ENERF_ACT_CASE(ID)                                    \
    case ID:                                                  \
        _Pragma("unroll") for (int q = 0; q < 16; q++) t[q] = act_fwd(t[q], ID); \
        break;
        switch (a) {
            ENERF_ACT_CASE(1) ENERF_ACT_CASE(2) ENERF_ACT_CASE(3) ENERF_ACT_CASE(4) ENERF_ACT_CASE(5)
            default: break;
        }
#undef ENERF_ACT_CASE
    }
}
__device__ __forceinline__ void apply_act_bwd(f32x16& g, const float (&fw)[16], uint32_t a) {
    if (a == 0) {
#pragma unroll
        for (int q = 0; q < 16; q++) g[q] = fw[q] > 0.0f ? g[q] : 0.0f;
    } else if (a != 6) {
#define ENERF_ACTB_CASE(ID)                                   \
    case ID:                                                  \
        _Pragma("unroll") for (int q = 0; q < 16; q++) g[q] = act_bwd(g[q], fw[q], ID); \
        break;
        switch (a) {
            ENERF_ACTB_CASE(1) ENERF_ACTB_CASE(3) ENERF_ACTB_CASE(4) ENERF_ACTB_CASE(5)
            default: break;
        }
#undef ENERF_ACTB_CASE
    }
}

// ---- weight-fragment builders (from the LDS copy of the blob) -------------------------------------------------
// natural K order: element e of K-block kb, lane half h  <->  column 16*kb + 8*h + e          (operand fed from memory)
// permuted K order: K-block (ib,kbb), element e           <->  column 32*ib + 16*kbb + 4*h + (e&3) + 8*(e>>2)
//                                                              (operand fed from the previous layer's D tile)
template <typename E>
__device__ __forceinline__ typename V<E>::x8 frag_row_nat(const E* m, int ld, int row, int kb, int h) {
    typename V<E>::x8 f;
#pragma unroll
    for (int e = 0; e < 8; e++) f[e] = m[row * ld + 16 * kb + 8 * h + e];
    return f;
}
template <typename E>
__device__ __forceinline__ typename V<E>::x8 frag_row_perm(const E* m, int ld, int row, int blk, int h) {
    typename V<E>::x8 f;   // blk = 2*ib + kbb
#pragma unroll
    for (int e = 0; e < 8; e++) f[e] = m[row * ld + 16 * blk + 4 * h + (e & 3) + 8 * (e >> 2)];
    return f;
}
// transposed: A[i = column `col` of m][k = row index]; natural / permuted order over the ROW index
template <typename E>
__device__ __forceinline__ typename V<E>::x8 frag_col_nat(const E* m, int ld, int col, int kb, int h, int nrows) {
    typename V<E>::x8 f;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int r = 16 * kb + 8 * h + e;
        f[e] = r < nrows ? m[r * ld + col] : (E)0.0f;
    }
    return f;
}
template <typename E>
__device__ __forceinline__ typename V<E>::x8 frag_col_perm(const E* m, int ld, int col, int blk, int h) {
    typename V<E>::x8 f;
#pragma unroll
    for (int e = 0; e < 8; e++) f[e] = m[(16 * blk + 4 * h + (e & 3) + 8 * (e >> 2)) * ld + col];
    return f;
}

// D tile (fp32, lane = sample) -> two permuted-order K-blocks of 16-bit operands, with an elementwise map
template <typename E>
__device__ __forceinline__ void tile_to_frags(const f32x16& acc, typename V<E>::x8 (&out)[2]) {
#pragma unroll
    for (int kbb = 0; kbb < 2; kbb++)
#pragma unroll
        for (int e = 0; e < 8; e++) out[kbb][e] = (E)acc[8 * kbb + e];
}

// store / load one [32 samples][32 neurons] D-tile-shaped block of a row-major [B,64] 16-bit buffer:
// lane (j, h) owns neurons 32*ib + 8*g + 4*h + r  (g = 0..3, r = 0..3)  <->  4 consecutive elements per g
template <typename E>
__device__ __forceinline__ void store_tile(E* rowptr /* row of sample j */, int ib, int h,
                                           const typename V<E>::x8 (&fr)[2]) {
#pragma unroll
    for (int g = 0; g < 4; g++) {
        typename V<E>::x4 q;
#pragma unroll
        for (int r = 0; r < 4; r++) q[r] = fr[g >> 1][(g & 1) * 4 + r];
        *reinterpret_cast<typename V<E>::x4*>(rowptr + 32 * ib + 8 * g + 4 * h) = q;
    }
}
template <typename E>
__device__ __forceinline__ void load_tile_f32(const E* rowptr, int ib, int h, float (&v)[16]) {
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const typename V<E>::x4 q = *reinterpret_cast<const typename V<E>::x4*>(rowptr + 32 * ib + 8 * g + 4 * h);
#pragma unroll
        for (int r = 0; r < 4; r++) v[4 * g + r] = (float)q[r];
    }
}

template <typename E>
__device__ __forceinline__ void stage_weights(E* wl, const E* __restrict__ w, uint32_t n) {
    // n is a multiple of 8 elements (hidden = 64, input_dim % 16 == 0)
    const uint4* src = reinterpret_cast<const uint4*>(w);
    uint4* dst = reinterpret_cast<uint4*>(wl);
    for (uint32_t i = threadIdx.x; i < n / 8; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}

// ================================================================== forward / inference
template <typename E, int IN_KB, int NL, bool TRAIN>
__global__ void __launch_bounds__(256) k_ffmlp_fwd(const E* __restrict__ X, const E* __restrict__ W, E* __restrict__ fb,
                                                   E* __restrict__ Y, uint32_t B, uint32_t act, uint32_t out_act) {
    using x8 = typename V<E>::x8;
    using x4 = typename V<E>::x4;
    constexpr int IN = 16 * IN_KB;
    constexpr uint32_t NW = HID * (IN + HID * (NL - 1) + OUT);
    __shared__ __attribute__((aligned(16))) E wl[NW];
    stage_weights(wl, W, NW);

    const int lane = lane_id(), j = lane & 31, h = lane >> 5;
    // weight fragments (A operands): lane = output neuron
    x8 w0[2][IN_KB], wh[NL - 1][2][4], wo[4];
#pragma unroll
    for (int ob = 0; ob < 2; ob++)
#pragma unroll
        for (int kb = 0; kb < IN_KB; kb++) w0[ob][kb] = frag_row_nat<E>(wl, IN, 32 * ob + j, kb, h);
#pragma unroll
    for (int l = 0; l < NL - 1; l++)
#pragma unroll
        for (int ob = 0; ob < 2; ob++)
#pragma unroll
            for (int blk = 0; blk < 4; blk++)
                wh[l][ob][blk] = frag_row_perm<E>(wl + HID * IN + l * HID * HID, HID, 32 * ob + j, blk, h);
    {
        const E* wout = wl + HID * IN + (NL - 1) * HID * HID;   // [16][64]; rows 16..31 of the MFMA tile are zero
#pragma unroll
        for (int blk = 0; blk < 4; blk++) {
            if (j < OUT) wo[blk] = frag_row_perm<E>(wout, HID, j, blk, h);
            else
#pragma unroll
                for (int e = 0; e < 8; e++) wo[blk][e] = (E)0.0f;
        }
    }

    const uint32_t ntiles = B / 32;
    const uint32_t gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t nw = gridDim.x * (blockDim.x >> 6);
    for (uint32_t tile = gw; tile < ntiles; tile += nw) {
        const size_t s = (size_t)tile * 32 + j;
        x8 xb[IN_KB];
#pragma unroll
        for (int kb = 0; kb < IN_KB; kb++) xb[kb] = *reinterpret_cast<const x8*>(X + s * IN + 16 * kb + 8 * h);

        f32x16 acc[2];
        x8 hb[2][2];
#pragma unroll
        for (int ob = 0; ob < 2; ob++) {
            acc[ob] = (f32x16)(0.0f);
#pragma unroll
            for (int kb = 0; kb < IN_KB; kb++) acc[ob] = mma(w0[ob][kb], xb[kb], acc[ob]);
        }
#pragma unroll
        for (int ob = 0; ob < 2; ob++) {
            apply_act(acc[ob], act);
            tile_to_frags<E>(acc[ob], hb[ob]);
            if (TRAIN) store_tile<E>(fb + s * HID, ob, h, hb[ob]);
        }
#pragma unroll
        for (int l = 1; l < NL; l++) {
#pragma unroll
            for (int ob = 0; ob < 2; ob++) {
                acc[ob] = (f32x16)(0.0f);
#pragma unroll
                for (int blk = 0; blk < 4; blk++) acc[ob] = mma(wh[l - 1][ob][blk], hb[blk >> 1][blk & 1], acc[ob]);
            }
            x8 nb[2][2];
#pragma unroll
            for (int ob = 0; ob < 2; ob++) {
                apply_act(acc[ob], act);
                tile_to_frags<E>(acc[ob], nb[ob]);
                if (TRAIN) store_tile<E>(fb + ((size_t)l * B + s) * HID, ob, h, nb[ob]);
            }
#pragma unroll
            for (int ob = 0; ob < 2; ob++)
#pragma unroll
                for (int kbb = 0; kbb < 2; kbb++) hb[ob][kbb] = nb[ob][kbb];
        }
        f32x16 ao = (f32x16)(0.0f);
#pragma unroll
        for (int blk = 0; blk < 4; blk++) ao = mma(wo[blk], hb[blk >> 1][blk & 1], ao);
        apply_act(ao, out_act);
        // rows 0..15 of the tile: registers 0..7 (g = 0, 1): outputs 8*g + 4*h + r
#pragma unroll
        for (int g = 0; g < 2; g++) {
            x4 q;
#pragma unroll
            for (int r = 0; r < 4; r++) q[r] = (E)ao[4 * g + r];
            *reinterpret_cast<x4*>(Y + s * OUT + 8 * g + 4 * h) = q;
        }
    }
}

// ================================================================== backward: activation gradients (dgrad chain)
template <typename E, int IN_KB, int NL>
__global__ void __launch_bounds__(256) k_ffmlp_bwd_act(const E* __restrict__ dY, const E* __restrict__ W,
                                                       const E* __restrict__ fb, E* __restrict__ bb,
                                                       E* __restrict__ dX, uint32_t B, uint32_t act) {
    using x8 = typename V<E>::x8;
    using x4 = typename V<E>::x4;
    constexpr int IN = 16 * IN_KB;
    constexpr int IN_MB = (IN + 31) / 32;
    constexpr uint32_t NW = HID * (IN + HID * (NL - 1) + OUT);
    __shared__ __attribute__((aligned(16))) E wl[NW];
    stage_weights(wl, W, NW);

    const int lane = lane_id(), j = lane & 31, h = lane >> 5;
    // transposed weight fragments: lane = INPUT neuron of the layer, K runs over its output neurons
    x8 woT[2], whT[NL - 1][2][4], wiT[IN_MB][4];
    {
        const E* wout = wl + HID * IN + (NL - 1) * HID * HID;
#pragma unroll
        for (int ib = 0; ib < 2; ib++) woT[ib] = frag_col_nat<E>(wout, HID, 32 * ib + j, 0, h, OUT);
    }
#pragma unroll
    for (int l = 0; l < NL - 1; l++)
#pragma unroll
        for (int ib = 0; ib < 2; ib++)
#pragma unroll
            for (int blk = 0; blk < 4; blk++)
                whT[l][ib][blk] = frag_col_perm<E>(wl + HID * IN + l * HID * HID, HID, 32 * ib + j, blk, h);
    if (dX) {
#pragma unroll
        for (int mb = 0; mb < IN_MB; mb++)
#pragma unroll
            for (int blk = 0; blk < 4; blk++) {
                if (32 * mb + j < IN) wiT[mb][blk] = frag_col_perm<E>(wl, IN, 32 * mb + j, blk, h);
                else
#pragma unroll
                    for (int e = 0; e < 8; e++) wiT[mb][blk][e] = (E)0.0f;
            }
    }

    const uint32_t ntiles = B / 32;
    const uint32_t gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t nw = gridDim.x * (blockDim.x >> 6);
    for (uint32_t tile = gw; tile < ntiles; tile += nw) {
        const size_t s = (size_t)tile * 32 + j;
        const x8 dyb = *reinterpret_cast<const x8*>(dY + s * OUT + 8 * h);   // one K-block: the 16 outputs
        x8 gb[2][2];
#pragma unroll
        for (int ib = 0; ib < 2; ib++) {
            f32x16 a = mma(woT[ib], dyb, (f32x16)(0.0f));
            float fw[16];
            load_tile_f32<E>(fb + ((size_t)(NL - 1) * B + s) * HID, ib, h, fw);
            apply_act_bwd(a, fw, act);
            tile_to_frags<E>(a, gb[ib]);
            store_tile<E>(bb + s * HID, ib, h, gb[ib]);
        }
#pragma unroll
        for (int jj = 1; jj < NL; jj++) {
            const int l = NL - jj;              // weights W_h[l-1]; produces dL/d(pre-activation of matmul l-1)
            x8 ng[2][2];
#pragma unroll
            for (int ib = 0; ib < 2; ib++) {
                f32x16 a = (f32x16)(0.0f);
#pragma unroll
                for (int blk = 0; blk < 4; blk++) a = mma(whT[l - 1][ib][blk], gb[blk >> 1][blk & 1], a);
                float fw[16];
                load_tile_f32<E>(fb + ((size_t)(l - 1) * B + s) * HID, ib, h, fw);
                apply_act_bwd(a, fw, act);
                tile_to_frags<E>(a, ng[ib]);
                store_tile<E>(bb + ((size_t)jj * B + s) * HID, ib, h, ng[ib]);
            }
#pragma unroll
            for (int ib = 0; ib < 2; ib++)
#pragma unroll
                for (int kbb = 0; kbb < 2; kbb++) gb[ib][kbb] = ng[ib][kbb];
        }
        if (dX) {
#pragma unroll
            for (int mb = 0; mb < IN_MB; mb++) {
                f32x16 a = (f32x16)(0.0f);
#pragma unroll
                for (int blk = 0; blk < 4; blk++) a = mma(wiT[mb][blk], gb[blk >> 1][blk & 1], a);
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int col = 32 * mb + 8 * g + 4 * h;
                    if (col < IN) {
                        x4 q;
#pragma unroll
                        for (int r = 0; r < 4; r++) q[r] = (E)a[4 * g + r];
                        *reinterpret_cast<x4*>(dX + s * IN + col) = q;
                    }
                }
            }
        }
    }
}

// ================================================================== backward: weight gradients
// dW_l[o][i] = sum_s dOut_l[s][o] * In_l[s][i].   Both operands live in memory as row-major [B, F] 16-bit buffers.
// A [32 samples][32 features] tile is loaded "lane = sample" (16 B per lane), flipped to "lane = feature, registers =
// samples" by D = tile x I on the matrix core (exact), and fed to the MFMA whose contraction index is the sample.
template <typename E>
__device__ __forceinline__ void flip_tile(const E* base /* row of sample j, feature 32*nb */, int h, int nfeat,
                                          const typename V<E>::x8 (&ident)[2], typename V<E>::x8 (&q)[2]) {
    using x8 = typename V<E>::x8;
    f32x16 t = (f32x16)(0.0f);
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
        if (16 * kb < nfeat) {
            const x8 a = *reinterpret_cast<const x8*>(base + 16 * kb + 8 * h);
            t = mma(a, ident[kb], t);
        }
    }
#pragma unroll
    for (int kbs = 0; kbs < 2; kbs++)
#pragma unroll
        for (int e = 0; e < 8; e++) q[kbs][e] = (E)t[8 * kbs + e];
}

template <typename E, int IN_KB, int NL>
__global__ void __launch_bounds__(256) k_ffmlp_bwd_w(const E* __restrict__ dY, const E* __restrict__ X,
                                                     const E* __restrict__ fb, const E* __restrict__ bb,
                                                     float* __restrict__ partial, uint32_t B) {
    using x8 = typename V<E>::x8;
    constexpr int IN = 16 * IN_KB;
    constexpr int IN_NB = (IN + 31) / 32;
    constexpr uint32_t NW = HID * (IN + HID * (NL - 1) + OUT);
    __shared__ float red[NW];
    for (uint32_t i = threadIdx.x; i < NW; i += blockDim.x) red[i] = 0.0f;
    __syncthreads();

    const int lane = lane_id(), j = lane & 31, h = lane >> 5;
    x8 ident[2];
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int e = 0; e < 8; e++) ident[kb][e] = (16 * kb + 8 * h + e == j) ? (E)1.0f : (E)0.0f;

    // accumulators: input layer [2 x IN_NB], hidden layers [NL-1][2 x 2], output layer [1 x 2]
    f32x16 aw0[2][IN_NB], awh[NL - 1][2][2], awo[2];
#pragma unroll
    for (int a = 0; a < 2; a++) {
#pragma unroll
        for (int b = 0; b < IN_NB; b++) aw0[a][b] = (f32x16)(0.0f);
#pragma unroll
        for (int l = 0; l < NL - 1; l++)
#pragma unroll
            for (int b = 0; b < 2; b++) awh[l][a][b] = (f32x16)(0.0f);
        awo[a] = (f32x16)(0.0f);
    }

    const uint32_t ntiles = B / 32;
    const uint32_t gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t nw = gridDim.x * (blockDim.x >> 6);
    for (uint32_t tile = gw; tile < ntiles; tile += nw) {
        const size_t s = (size_t)tile * 32 + j;
        // activation gradients of every matmul, flipped: gq[m][ob] = dL/d(pre-act of matmul m), m = 0..NL-1,
        // stored in backward_buffer[NL-1-m]
        x8 prev[2][2];    // flipped INPUT of the current matmul (features in 2 blocks of 32)
        // ---- input layer: In = X
        {
            x8 xq[IN_NB][2];
#pragma unroll
            for (int nb = 0; nb < IN_NB; nb++) flip_tile<E>(X + s * IN + 32 * nb, h, IN - 32 * nb, ident, xq[nb]);
#pragma unroll
            for (int ob = 0; ob < 2; ob++) {
                x8 gq[2];
                flip_tile<E>(bb + ((size_t)(NL - 1) * B + s) * HID + 32 * ob, h, 32, ident, gq);
#pragma unroll
                for (int nb = 0; nb < IN_NB; nb++)
#pragma unroll
                    for (int kbs = 0; kbs < 2; kbs++) aw0[ob][nb] = mma(gq[kbs], xq[nb][kbs], aw0[ob][nb]);
            }
        }
        // ---- hidden layers: matmul m (1..NL-1), In = forward_buffer[m-1], dOut = backward_buffer[NL-1-m]
#pragma unroll
        for (int m = 1; m < NL; m++) {
#pragma unroll
            for (int nb = 0; nb < 2; nb++)
                flip_tile<E>(fb + ((size_t)(m - 1) * B + s) * HID + 32 * nb, h, 32, ident, prev[nb]);
#pragma unroll
            for (int ob = 0; ob < 2; ob++) {
                x8 gq[2];
                flip_tile<E>(bb + ((size_t)(NL - 1 - m) * B + s) * HID + 32 * ob, h, 32, ident, gq);
#pragma unroll
                for (int nb = 0; nb < 2; nb++)
#pragma unroll
                    for (int kbs = 0; kbs < 2; kbs++)
                        awh[m - 1][ob][nb] = mma(gq[kbs], prev[nb][kbs], awh[m - 1][ob][nb]);
            }
        }
        // ---- output layer: In = forward_buffer[NL-1], dOut = dY (16 features)
        {
#pragma unroll
            for (int nb = 0; nb < 2; nb++)
                flip_tile<E>(fb + ((size_t)(NL - 1) * B + s) * HID + 32 * nb, h, 32, ident, prev[nb]);
            x8 gq[2];
            flip_tile<E>(dY + s * OUT, h, OUT, ident, gq);
#pragma unroll
            for (int nb = 0; nb < 2; nb++)
#pragma unroll
                for (int kbs = 0; kbs < 2; kbs++) awo[nb] = mma(gq[kbs], prev[nb][kbs], awo[nb]);
        }
    }

    // D tile: lane (col j = input neuron within block nb, half h), register q -> output neuron 32*ob + (q&3) + 8*(q>>2) + 4*h.
    // The four waves add their tiles into the LDS copy one after the other (fixed order => run-to-run deterministic).
    auto flush = [&](const f32x16& a, uint32_t base, int ld, int ob, int nb, int nrows, int ncols) {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int o = 32 * ob + (q & 3) + 8 * (q >> 2) + 4 * h;
            const int i = 32 * nb + j;
            if (o < nrows && i < ncols) red[base + o * ld + i] += a[q];
        }
    };
    const int wid = threadIdx.x >> 6;
    for (int turn = 0; turn < 4; turn++) {
        if (wid == turn) {
#pragma unroll
            for (int ob = 0; ob < 2; ob++)
#pragma unroll
                for (int nb = 0; nb < IN_NB; nb++) flush(aw0[ob][nb], 0, IN, ob, nb, HID, IN);
#pragma unroll
            for (int l = 0; l < NL - 1; l++)
#pragma unroll
                for (int ob = 0; ob < 2; ob++)
#pragma unroll
                    for (int nb = 0; nb < 2; nb++)
                        flush(awh[l][ob][nb], HID * IN + l * HID * HID, HID, ob, nb, HID, HID);
#pragma unroll
            for (int nb = 0; nb < 2; nb++) flush(awo[nb], HID * IN + (NL - 1) * HID * HID, HID, 0, nb, OUT, HID);
        }
        __syncthreads();
    }
    float* dst = partial + (size_t)blockIdx.x * NW;
    for (uint32_t i = threadIdx.x; i < NW; i += blockDim.x) dst[i] = red[i];
}

template <typename E>
__global__ void __launch_bounds__(256) k_ffmlp_reduce_w(const float* __restrict__ partial, uint32_t nblocks, uint32_t NW,
                                                        E* __restrict__ gw) {
    __shared__ float acc[4][64];
    const uint32_t i = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t part = threadIdx.x >> 6;
    float s = 0.0f;
    if (i < NW)
        for (uint32_t b = part; b < nblocks; b += 4) s += partial[(size_t)b * NW + i];
    acc[part][threadIdx.x & 63] = s;
    __syncthreads();
    // grad_weights arrives zero-filled; accumulate like the reference's beta = 0/1 GEMMs
    if (part == 0 && i < NW)
        gw[i] = (E)((float)gw[i] + ((acc[0][threadIdx.x] + acc[1][threadIdx.x]) + (acc[2][threadIdx.x] + acc[3][threadIdx.x])));
}

// ------------------------------------------------------------------ host dispatch
uint32_t persistent_grid(uint32_t B) {
    const uint32_t tiles = B / 32;
    const uint32_t blocks = div_up(tiles, 4);
    return blocks < 512u ? blocks : 512u;
}

int check_shape(uint32_t B, uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers,
                int dtype) {
    if (dtype != ENERF_F16 && dtype != ENERF_BF16) ENERF_BADARG("ffmlp: dtype must be f16 or bf16");
    if (hidden_dim != HID) ENERF_BADARG("ffmlp: this build supports hidden_dim == 64 (got %u)", hidden_dim);
    if (output_dim != OUT) ENERF_BADARG("ffmlp: output_dim must be 16 (padded), got %u", output_dim);
    if (input_dim != 16 && input_dim != 32 && input_dim != 64)
        ENERF_BADARG("ffmlp: input_dim must be 16, 32 or 64 (got %u)", input_dim);
    if (num_layers < 2 || num_layers > MAX_NL) ENERF_BADARG("ffmlp: num_layers must be in [2, %d] (got %u)", MAX_NL, num_layers);
    if (B % 128 != 0) ENERF_BADARG("ffmlp: batch size must be a multiple of 128 (got %u)", B);
    return 0;
}

#define FFMLP_DISPATCH(E, CALL)                                                     \
    switch (input_dim * 10 + num_layers) {                                          \
        case 162: { constexpr int KB = 1, NL = 2; CALL; } break;                    \
        case 163: { constexpr int KB = 1, NL = 3; CALL; } break;                    \
        case 164: { constexpr int KB = 1, NL = 4; CALL; } break;                    \
        case 322: { constexpr int KB = 2, NL = 2; CALL; } break;                    \
        case 323: { constexpr int KB = 2, NL = 3; CALL; } break;                    \
        case 324: { constexpr int KB = 2, NL = 4; CALL; } break;                    \
        case 642: { constexpr int KB = 4, NL = 2; CALL; } break;                    \
        case 643: { constexpr int KB = 4, NL = 3; CALL; } break;                    \
        case 644: { constexpr int KB = 4, NL = 4; CALL; } break;                    \
        default: ENERF_BADARG("ffmlp: unsupported input_dim/num_layers");           \
    }

template <typename E, bool TRAIN>
int run_fwd(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim, uint32_t num_layers, uint32_t act,
            uint32_t out_act, void* buffer, void* outputs, hipStream_t s) {
    const uint32_t grid = persistent_grid(B);
    FFMLP_DISPATCH(E, (k_ffmlp_fwd<E, KB, NL, TRAIN><<<grid, 256, 0, s>>>((const E*)inputs, (const E*)weights, (E*)buffer,
                                                                            (E*)outputs, B, act, out_act)));
    return 0;
}

template <typename E>
int run_bwd(const void* grad, const void* inputs, const void* weights, const void* fb, uint32_t B, uint32_t input_dim,
            uint32_t num_layers, uint32_t act, bool calc_grad_inputs, void* bb, void* grad_inputs, void* grad_weights,
            hipStream_t s) {
    const uint32_t grid = persistent_grid(B);
    const uint32_t NWn = HID * (input_dim + HID * (num_layers - 1) + OUT);
    const uint32_t wgrid = grid < 256u ? grid : 256u;
    float* partial = (float*)workspace(WS_FFMLP, sizeof(float) * (size_t)wgrid * NWn);
    if (!partial) return ENERF_E_NOMEM;
    FFMLP_DISPATCH(E, (k_ffmlp_bwd_act<E, KB, NL><<<grid, 256, 0, s>>>((const E*)grad, (const E*)weights, (const E*)fb,
                                                                        (E*)bb, calc_grad_inputs ? (E*)grad_inputs : nullptr,
                                                                        B, act)));
    FFMLP_DISPATCH(E, (k_ffmlp_bwd_w<E, KB, NL><<<wgrid, 256, 0, s>>>((const E*)grad, (const E*)inputs, (const E*)fb,
                                                                       (const E*)bb, partial, B)));
    k_ffmlp_reduce_w<E><<<div_up(NWn, 64), 256, 0, s>>>(partial, wgrid, NWn, (E*)grad_weights);
    return 0;
}

}  // namespace

extern "C" {

int enerf_ffmlp_forward(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                        uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                        void* forward_buffer, void* outputs, int dtype, enerf_stream_t stream) {
    if (B == 0) return 0;
    int rc = check_shape(B, input_dim, output_dim, hidden_dim, num_layers, dtype);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(ENERF_K_FFMLP_FWD, s);
    rc = dtype == ENERF_BF16
             ? run_fwd<__bf16, true>(inputs, weights, B, input_dim, num_layers, activation, output_activation, forward_buffer, outputs, s)
             : run_fwd<_Float16, true>(inputs, weights, B, input_dim, num_layers, activation, output_activation, forward_buffer, outputs, s);
    if (rc) return rc;
    ENERF_LAUNCH_CHECK("ffmlp_forward");
    return 0;
}

int enerf_ffmlp_inference(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                          uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                          void* inference_buffer, void* outputs, int dtype, enerf_stream_t stream) {
    (void)inference_buffer;   // activations never leave the registers in inference
    if (B == 0) return 0;
    int rc = check_shape(B, input_dim, output_dim, hidden_dim, num_layers, dtype);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(ENERF_K_FFMLP_FWD, s);
    rc = dtype == ENERF_BF16
             ? run_fwd<__bf16, false>(inputs, weights, B, input_dim, num_layers, activation, output_activation, nullptr, outputs, s)
             : run_fwd<_Float16, false>(inputs, weights, B, input_dim, num_layers, activation, output_activation, nullptr, outputs, s);
    if (rc) return rc;
    ENERF_LAUNCH_CHECK("ffmlp_inference");
    return 0;
}

int enerf_ffmlp_backward(const void* grad, const void* inputs, const void* weights, const void* forward_buffer,
                         uint32_t B, uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers,
                         uint32_t activation, uint32_t output_activation, int calc_grad_inputs, void* backward_buffer,
                         void* grad_inputs, void* grad_weights, int dtype, enerf_stream_t stream) {
    (void)output_activation;   // ignored in backward, as in the reference (ffmlp.cu:781)
    if (B == 0) return 0;
    int rc = check_shape(B, input_dim, output_dim, hidden_dim, num_layers, dtype);
    if (rc) return rc;
    if (activation == 2) ENERF_BADARG("ffmlp: sine activation has no backward (as in the reference)");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(ENERF_K_FFMLP_BWD, s);
    rc = dtype == ENERF_BF16
             ? run_bwd<__bf16>(grad, inputs, weights, forward_buffer, B, input_dim, num_layers, activation, calc_grad_inputs != 0, backward_buffer, grad_inputs, grad_weights, s)
             : run_bwd<_Float16>(grad, inputs, weights, forward_buffer, B, input_dim, num_layers, activation, calc_grad_inputs != 0, backward_buffer, grad_inputs, grad_weights, s);
    if (rc) return rc;
    ENERF_LAUNCH_CHECK("ffmlp_backward");
    return 0;
}

int enerf_allocate_splitk(size_t size) {
    (void)size;   // weight gradients are reduced inside the fused backward; nothing to pre-create
    return 0;
}
int enerf_free_splitk(void) { return 0; }

}  // extern "C"
