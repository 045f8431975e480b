// mlp32_common.h -- types, operand layouts, weight staging and tile I/O shared by mlp32.hip (fp32 MFMA kernels, host
// entry points) and mlp32s.hip (the split-bf16 kernels, compiled with MFMA results in arch VGPRs).  Design notes are at
// the top of mlp32.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"
#include "sh_basis.h"

namespace enerf_mlp32 {
using namespace enerf;

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int HID = 64;
constexpr int IN = 32;

__device__ __forceinline__ f32x16 mma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// neuron (within a 32-block) held in accumulator register q by lane half h
__device__ __forceinline__ int nrow(int q, int h) { return (q & 3) + 8 * (q >> 2) + 4 * h; }

__device__ __forceinline__ float act_fwd(float x, uint32_t a) { return a == 0 ? (x > 0 ? x : 0.0f) : x; }
// output activation: relu (0), sigmoid (3, torch's 1 / (1 + exp(-x))), none (6)
__device__ __forceinline__ float out_act_fwd(float x, uint32_t a) {
    return a == 0 ? (x > 0 ? x : 0.0f) : (a == 3 ? 1.0f / (1.0f + expf(-x)) : x);
}

// Where the backward kernels take dL/dY from.  Plain: dY[s * stride + o].  Optional fusions of the caller's epilogue:
//   y_sig != NULL : the forward applied a sigmoid; dY is the gradient of the sigmoid's output and y_sig its value:
//                   dL/dY = (dY * (1 - y)) * y                                     (torch sigmoid_backward)
//   dsigma != NULL: output column 0 went through trunc_exp (activation.py:5-17); its gradient is
//                   dsigma[s] * exp(clamp(h0[s * h0_stride], -15, 15)) instead of dY[.., 0]
struct DySource {
    const float* dY;
    uint32_t stride;
    const float* y_sig;
    uint32_t y_sig_stride;
    const float* dsigma;
    const float* h0;
    uint32_t h0_stride;
};
__device__ __forceinline__ float load_dy(const DySource& d, size_t s, uint32_t o) {
    if (d.dsigma && o == 0) return d.dsigma[s] * expf(fminf(fmaxf(d.h0[s * d.h0_stride], -15.0f), 15.0f));
    float g = d.dY[s * d.stride + o];
    if (d.y_sig) {
        const float y = d.y_sig[s * d.y_sig_stride + o];
        g = (g * (1.0f - y)) * y;
    }
    return g;
}
__device__ __forceinline__ float act_bwd(float g, float fwd, uint32_t a) { return a == 0 ? (fwd > 0 ? g : 0.0f) : g; }

// blob: [W0 64 x 32 | Wh (NH-1) x 64 x 64 | Wout out_dim x 64], row-major W[out][in]
__device__ __forceinline__ uint32_t blob_size(int NH, uint32_t out_dim) {
    return HID * IN + (NH - 1) * HID * HID + out_dim * HID;
}

// Where the matrices of the logical blob live.  A contiguous blob is the special case seg[k] = blob + offset; the
// fused NeRF network points straight at its nn.Linear weights instead of packing them every step: w0 rows may then be
// 31 floats long and ordered [SH 16 | geo_feat 15] in memory (nerf/network.py:95) while the kernels' input rows are
// [raw density | geo_feat 15 | SH 16] -- the permutation (and the zero column) is applied while staging.
struct WSrc {
    const float* seg[4];     // first layer, hidden 0, hidden 1, output layer
    uint32_t w0_cols;        // floats per first-layer row in memory: 32, or 31 with nerf_perm
    uint32_t nerf_perm;
    // optional (enerf_mlp32_valid_rows): device int32, rows >= min(*valid_rows, B) are padding the caller never reads --
    // a training batch is a budget of M rows of which the marcher filled counter[0].  The forward skips their tiles,
    // the fused backward writes zero input gradients for them and skips the rest.  A wave's tiles are strided, so what
    // is skipped is its LAST round: at 4163 tiles over 2048 (1024) resident waves, a batch of <= 4096 real tiles takes
    // two (four) rounds instead of three (five).
    const int32_t* valid_rows;
    // enerf_mlp32_valid_rows_ex: the real rows are valid_base + min(*valid_rows, valid_cap) (two renders' samples in one
    // batch: the first render's M rows -- its padding included -- then the second's counter, capped at its own M)
    uint32_t valid_base, valid_cap;
};
__device__ __forceinline__ uint32_t valid_tiles(const WSrc& W, uint32_t B, uint32_t ntiles) {
    if (!W.valid_rows) return ntiles;
    int32_t v = W.valid_rows[0];
    if (W.valid_cap) v = (int32_t)W.valid_base + (v <= 0 ? 0 : (v < (int32_t)W.valid_cap ? v : (int32_t)W.valid_cap));
    const uint32_t rows = v <= 0 ? 0u : ((uint32_t)v < B ? (uint32_t)v : B);
    const uint32_t t = (rows + 31u) / 32u;
    return t < ntiles ? t : ntiles;
}
struct WDst {                // the same for the weight gradients the reduce pass writes
    float* seg[4];
    uint32_t w0_cols, nerf_perm, overwrite;      // overwrite: dW = sum (no zero-filled accumulator needed), else +=
};
// memory column of kernel column c of the first layer (-1: the kernel column has no weight: zero)
__device__ __forceinline__ int w0_col(uint32_t c, uint32_t nerf_perm) {
    if (!nerf_perm) return (int)c;
    return c == 0 ? -1 : (c < 16 ? (int)c + 15 : (int)c - 16);
}
__device__ __forceinline__ float wsrc_at(const WSrc& w, uint32_t i) {      // element i of the logical blob
    if (i < HID * IN) {
        const int c = w0_col(i % IN, w.nerf_perm);
        return c < 0 ? 0.0f : w.seg[0][(i / IN) * w.w0_cols + c];
    }
    i -= HID * IN;
    // hidden matrices and the output layer follow one another; which segment is decided by the caller's NH through
    // the pointers: unused hidden slots are null and skipped
    if (w.seg[1]) {
        if (i < HID * HID) return w.seg[1][i];
        i -= HID * HID;
    }
    if (w.seg[2]) {
        if (i < HID * HID) return w.seg[2][i];
        i -= HID * HID;
    }
    return w.seg[3][i];
}
__device__ __forceinline__ float* wdst_at(const WDst& w, uint32_t i) {     // nullptr: nowhere (the zero column)
    if (i < HID * IN) {
        const int c = w0_col(i % IN, w.nerf_perm);
        return c < 0 ? nullptr : w.seg[0] + (i / IN) * w.w0_cols + c;
    }
    i -= HID * IN;
    if (w.seg[1]) {
        if (i < HID * HID) return w.seg[1] + i;
        i -= HID * HID;
    }
    if (w.seg[2]) {
        if (i < HID * HID) return w.seg[2] + i;
        i -= HID * HID;
    }
    return w.seg[3] + i;
}

// Copy the logical blob into LDS, matrix by matrix (a thread keeps its column, so the first layer's permutation is
// resolved once per thread and nothing is divided per element).  ROT: rows rotated by their index (stage_rot below).
// Every global load of the thread is issued before the first LDS store: the trip counts are compile-time (256
// threads, at most 32 output rows), so the copy costs one memory latency -- as a plain load / store loop it cost one
// per iteration, 36 of them for the colour net (measured: 15 us of fixed cost per forward launch, tools/mlp32_fit.sh).
template <bool ROT>
__device__ __forceinline__ void stage_segments(float* wl, const WSrc& w, uint32_t n) {
    constexpr uint32_t R0 = 256 / IN, N0 = HID / R0;           // first layer: rows per pass, passes
    constexpr uint32_t RH = 256 / HID, NHID = HID / RH;        // 64-wide matrices
    const uint32_t c0 = threadIdx.x & (IN - 1), r0 = threadIdx.x / IN;
    const uint32_t c = threadIdx.x & (HID - 1), rh = threadIdx.x / HID;
    const int sc = w0_col(c0, w.nerf_perm);
    float v0[N0], v[3][NHID];
#pragma unroll
    for (uint32_t k = 0; k < N0; k++) v0[k] = w.seg[0][(r0 + k * R0) * w.w0_cols + (sc < 0 ? 0 : sc)];
    uint32_t cnt[3];
    {
        uint32_t base = HID * IN;
#pragma unroll
        for (int m = 1; m < 4; m++) {
            cnt[m - 1] = !w.seg[m] ? 0u : (m < 3 ? HID * HID : n - base);    // the output layer takes what is left
            base += cnt[m - 1];
#pragma unroll
            for (uint32_t k = 0; k < NHID; k++) {
                // rows past the end of the output layer re-read its last row: branch-free, discarded below
                const uint32_t r = rh + k * RH, rc = r * HID < cnt[m - 1] ? r : cnt[m - 1] / HID - 1;
                v[m - 1][k] = cnt[m - 1] ? w.seg[m][rc * HID + c] : 0.0f;
            }
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < N0; k++) {
        const uint32_t r = r0 + k * R0;
        wl[ROT ? r * IN + ((c0 + r) & (IN - 1)) : r * IN + c0] = sc < 0 ? 0.0f : v0[k];
    }
    uint32_t row0 = 0;                            // rows of the 64-wide matrices, numbered through
#pragma unroll
    for (int m = 1; m < 4; m++) {
#pragma unroll
        for (uint32_t k = 0; k < NHID; k++) {
            const uint32_t r = rh + k * RH, rr = row0 + r;
            if (r * HID < cnt[m - 1])
                wl[HID * IN + (ROT ? rr * HID + ((c + rr) & (HID - 1)) : rr * HID + c)] = v[m - 1][k];
        }
        row0 += cnt[m - 1] / HID;
    }
    __syncthreads();
}
__device__ __forceinline__ void stage(float* wl, const WSrc& w, uint32_t n) { stage_segments<false>(wl, w, n); }

// Forward kernel staging: each matrix row is rotated by its row index (element (r, c) of a K-wide matrix sits at
// r * K + (c + r) % K).  The forward's register set-up reads one column per instruction, rows across lanes -- a
// 32-way bank conflict in the plain layout, conflict-free in the rotated one.
__device__ __forceinline__ uint32_t rot(uint32_t r, uint32_t c, uint32_t K) { return r * K + ((c + r) & (K - 1)); }
__device__ __forceinline__ void stage_rot(float* wl, const WSrc& w, int NH, uint32_t out_dim) {
    stage_segments<true>(wl, w, blob_size(NH, out_dim));
}

// store / load a D-tile-shaped [32 samples][32 neurons] block of a row-major [B,64] fp32 buffer (16 B per g)
__device__ __forceinline__ void store_tile(float* rowptr, int ib, int h, const f32x16& v) {
#pragma unroll
    for (int g = 0; g < 4; g++)
        *reinterpret_cast<float4*>(rowptr + 32 * ib + 8 * g + 4 * h) =
            make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
}
__device__ __forceinline__ void load_tile(const float* rowptr, int ib, int h, f32x16& v) {
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const float4 t = *reinterpret_cast<const float4*>(rowptr + 32 * ib + 8 * g + 4 * h);
        v[4 * g] = t.x; v[4 * g + 1] = t.y; v[4 * g + 2] = t.z; v[4 * g + 3] = t.w;
    }
}

// The forward buffer `fb` is private to this file (written by the training forward, read by the backward), so it is
// kept TILE-NATIVE instead of row-major: within the 2048 floats of a (layer, 32-sample tile) block, element (sample j,
// neuron 32*ib + 8*g + 4*h + r) sits at ib*1024 + g*256 + (j + 32*h)*4 + r -- every store / load instruction of a wave
// then covers 1 KB of contiguous memory (8 full lines) instead of 64 scattered 16-byte pieces of 32 different rows.
// `rowptr` = what the row-major address of sample j's row would be (fb + (l*Bp + s)*64); the tile base follows from it.
__device__ __forceinline__ void store_tile_fb(float* rowptr, int ib, int h, const f32x16& v) {
    const int j = lane_id() & 31;
    float* tb = rowptr - j * HID + ib * 1024 + (j + 32 * h) * 4;
#pragma unroll
    for (int g = 0; g < 4; g++)
        *reinterpret_cast<float4*>(tb + g * 256) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
}
__device__ __forceinline__ void load_tile_fb(const float* rowptr, int ib, int h, f32x16& v) {
    const int j = lane_id() & 31;
    const float* tb = rowptr - j * HID + ib * 1024 + (j + 32 * h) * 4;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const float4 t = *reinterpret_cast<const float4*>(tb + g * 256);
        v[4 * g] = t.x; v[4 * g + 1] = t.y; v[4 * g + 2] = t.z; v[4 * g + 3] = t.w;
    }
}
// single element (sample row `srow` of the tile starting at `tilebase`, neuron n) of a tile-native fb block
__device__ __forceinline__ float fb_at(const float* tilebase, int srow, int n) {
    return tilebase[(n >> 5) * 1024 + ((n >> 3) & 3) * 256 + (srow + 32 * ((n >> 2) & 1)) * 4 + (n & 3)];
}

// layer-0 contraction index handled by MFMA p of lane half h
//   XL 0: pairs (p, 16 + p): lane half h reads the contiguous input columns 16h .. 16h+15
//   XL 1: lane half h reads the float2 of level 2q + h (q = 0..7): MFMA 2q + c contracts columns 4q + c and 4q + 2 + c
template <int XL>
__device__ __forceinline__ int kmap(int p, int h) {
    return XL == 0 ? 16 * h + p : 4 * (p >> 1) + 2 * h + (p & 1);
}

// the 16 inputs lane (j, h) feeds to the first layer for sample 32 * tile + j (zeros past the end of the batch)
template <int XL>
__device__ __forceinline__ void load_x(const float* __restrict__ X, uint32_t tile, int j, int h, uint32_t B, uint32_t Bp,
                                       float (&x)[16]) {
    const size_t s = (size_t)tile * 32 + j;
    const bool valid = s < B;
    if (XL == 0) {
        const size_t sc = valid ? s : (size_t)B - 1;
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const float4 t = *reinterpret_cast<const float4*>(X + sc * IN + 16 * h + 4 * v);
            x[4 * v] = t.x; x[4 * v + 1] = t.y; x[4 * v + 2] = t.z; x[4 * v + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const float2 t = *reinterpret_cast<const float2*>(X + ((size_t)(2 * q + h) * Bp + s) * 2);
            x[2 * q] = t.x; x[2 * q + 1] = t.y;
        }
    }
    if (!valid) {
#pragma unroll
        for (int p = 0; p < 16; p++) x[p] = 0.0f;
    }
}

// split-bf16 kernels (mlp32s.hip); prec = 3 (hi + lo operands: fp32 accuracy) or 1 (bf16 operands: the FFMLP nets);
// the launchers choose the template instance, the caller has validated the arguments
void mlp32s_launch_fwd(int prec, uint32_t num_hidden, bool train, uint32_t x_layout, bool sigma_only, const float* X, const WSrc& W,
                       float* fb, float* Y, uint32_t B, uint32_t out_dim, uint32_t act, uint32_t out_act,
                       uint32_t y_stride, float* y0_exp, const float* sh_dirs, uint32_t grid, size_t lds, hipStream_t s,
                       hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, bool io16 = false);
void mlp32s_launch_bwd(int prec, uint32_t num_hidden, uint32_t x_layout, const DySource& dys, const float* X, const WSrc& W,
                       const float* fb, float* dX, float* partial, uint32_t B, uint32_t out_dim, uint32_t act,
                       uint32_t grid, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr,
                       bool recompute = false, bool io16 = false);

// the same kernels with IEEE half operands (mlp32s_f16.hip; prec is 1 there: one product per operand pair)
void mlp32s_f16_launch_fwd(int prec, uint32_t num_hidden, bool train, uint32_t x_layout, bool sigma_only, const float* X,
                           const WSrc& W, float* fb, float* Y, uint32_t B, uint32_t out_dim, uint32_t act, uint32_t out_act,
                           uint32_t y_stride, float* y0_exp, const float* sh_dirs, uint32_t grid, size_t lds, hipStream_t s,
                           hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, bool io16 = false);
void mlp32s_f16_launch_bwd(int prec, uint32_t num_hidden, uint32_t x_layout, const DySource& dys, const float* X,
                           const WSrc& W, const float* fb, float* dX, float* partial, uint32_t B, uint32_t out_dim,
                           uint32_t act, uint32_t grid, hipStream_t s, hipEvent_t ev_start = nullptr,
                           hipEvent_t ev_stop = nullptr, bool recompute = false, bool io16 = false);

// nerf_mlp.hip: sigma + colour net of nerf/network.py as one launch per direction (split-bf16).  `frags`: 44 operand
// fragments of 2 KiB built by nerf_launch_frags from the five weight matrices; `partial`: grid x kNerfPartialStride floats
// of per-workgroup weight-gradient sums, [sigma blob 3072 | colour blob 6144 + 64 out_c] each.
constexpr uint32_t kNerfFragBytes = 44 * 2048;
constexpr uint32_t kNerfMapBytes = 44 * 512 * 4;      // nerf_launch_frag_map's table, kept behind the fragments
constexpr uint32_t kNerfPartMapBytes = ((HID * IN + 16 * HID) + (HID * IN + HID * HID + 16 * HID)) * 4;   // PartialSums.map
constexpr uint32_t kNerfWsBytes = kNerfFragBytes + kNerfMapBytes + kNerfPartMapBytes;
constexpr uint32_t kNerfPartialStride = (HID * IN + 16 * HID) + (HID * IN + HID * HID + 16 * HID);
constexpr uint32_t kNerfSigmaWords = HID * IN + 16 * HID;
// workgroups of k_nerf_fwd per CU (its 48 KiB of LDS and ~136 registers admit three; -DNERF_FWD_ONE_PER_CU: one)
#ifndef NERF_FWD_ONE_PER_CU
constexpr uint32_t kNerfFwdPerCu = 3;
#else
constexpr uint32_t kNerfFwdPerCu = 1;
#endif
void nerf_launch_frags(const float* ws0, const float* ws1, const float* wc0, const float* wc1, const float* wc2,
                       uint32_t w0_cols, uint32_t out_c, uint32_t* frags, hipStream_t s);
void nerf_launch_frag_map(uint32_t w0_cols, uint32_t out_c, uint32_t* map, hipStream_t s);
void nerf_launch_fwd(const float* X, const float* dirs, const uint32_t* frags, float* sigma, float* rgb, uint32_t B,
                     uint32_t out_c, const int32_t* valid_rows, uint32_t valid_base, uint32_t valid_cap, uint32_t grid,
                     hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
void nerf_launch_bwd(const float* X, const float* dirs, const float* g_rgb, const float* rgb, const float* g_sigma,
                     float sigma_scale, const uint32_t* frags, float* dX, float* partial, uint32_t B, uint32_t out_c,
                     const int32_t* valid_rows, uint32_t valid_base, uint32_t valid_cap, uint32_t grid, hipStream_t s,
                     hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);

// The reference's FFMLP entry points on the data flow above (mlp32.hip; called by ffmlp.hip): 16-bit row-major X / Y /
// dY / dX (dtype ENERF_BF16 or ENERF_F16), weights and weight gradients as fp32 blobs [W0 64x32 | Wh | Wout 16x64]; the
// hidden activations are recomputed in the backward, nothing is stored between the two calls.  num_hidden = 2 or 3.
int ffmlp16_forward(int dtype, const void* X, const float* W32, uint32_t B, uint32_t num_hidden, uint32_t activation,
                    void* Y, hipStream_t s);
int ffmlp16_backward(int dtype, const void* dY, const void* X, const float* W32, uint32_t B, uint32_t num_hidden,
                     uint32_t activation, void* dX, float* dW32, hipStream_t s);

}  // namespace enerf_mlp32
