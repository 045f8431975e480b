// mlp32.hip -- fused fp32 MLP (64-wide hidden layers, no bias) on the fp32 matrix-core path of gfx950.
//
// The reference's default networks (nerf/network.py: sigma_net 32->64->16, color_net 31->64->64->3) are nn.Linear
// stacks; on MI355X the skinny fp32 GEMMs they turn into dominate a 4096-ray training step.  This file evaluates such
// a stack -- forward, activation gradients, weight gradients -- in fused kernels built on
// v_mfma_f32_32x32x2_f32, whose result is bit-for-bit an fp32 fmaf chain (MI355X_MICROARCH.md), so sigma / rgb stay
// within fp32 round-off of the nn.Linear statement (parity target 1e-4 rel).  It is not part of the reference's
// native surface; enerf_amd/network.py routes its MLPs here (enerf_amd/fused_mlp.py), state_dict unchanged.
//
// Orientation as in ffmlp.hip: D[neuron][sample].  With one fp32 per lane per operand the D tile of one layer is the
// B operand of the next layer *as is* (MFMA q of input block ib consumes accumulator register q).  The backward of a
// net with up to two hidden layers is ONE kernel (k_mlp32_bwd_fused): the dgrad chain passes each activation /
// gradient tile through LDS once and consumes it as a weight-gradient operand on the spot, so no backward buffer is
// written.  Deeper nets use the separate dgrad (k_mlp32_bwd_act) and weight-gradient (k_mlp32_bwd_w) kernels; the
// latter reads its operands straight from the row-major [B,64] buffers (two 128-byte row segments per MFMA).
//
// Batches are ragged: B is the number of valid samples, every scratch buffer (fb, bb) and the level-major input has
// Bp = B rounded up to 32 rows, and samples >= B carry zeros through every kernel, so callers never pad or copy.
// Input layouts (XL): 0 = row-major [B,32]; 1 = level-major [16,Bp,2], exactly what the grid encoder writes with
// out_layout 2 (gridencoder.hip), so the encoding never has to be transposed into rows: a wavefront reads / writes
// 256 contiguous bytes per level.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <hip/hip_ext.h>

#include "common.h"
#include "mlp32_common.h"
#include "sh_basis.h"

using namespace enerf;
using namespace enerf_mlp32;

namespace {

// (types, operand layouts, weight staging and tile I/O: mlp32_common.h)

// ================================================================== forward
// SIG: only exp(output 0) is wanted (Y == NULL, y0_exp set: the density-grid update) -- the output layer is then one
// 64-term dot product per sample on the VALU (each half-wave holds 32 of the 64 hidden activations of its sample)
// instead of a 32-row MFMA tile of which 31 rows would be thrown away.
// SH: the kernel also writes the degree-4 SH encoding of sh_dirs[s] into columns 16..31 of row s of Y (the colour
// net's direction inputs, nerf/network.py:95): the separate encoder launch and its 10 us disappear into the MFMA shadow.
#ifdef ENERF_MLP_TIMING
// development aid: shader-clock cycles per phase of the fused backward, summed over waves (tools/bench_mlp32.py)
__device__ unsigned long long g_mlp_phase[16];
#define MLP_PH(k)                                                  \
    do {                                                           \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        ph[k] += t_ - t_last;                                      \
        t_last = t_;                                               \
    } while (0)
#else
#define MLP_PH(k)
#endif
template <int NH, bool TRAIN, int XL, bool SIG = false, bool SH = false>
__global__ void __launch_bounds__(256) k_mlp32_fwd(const float* __restrict__ X, WSrc W,
                                                   float* __restrict__ fb, float* __restrict__ Y, uint32_t B,
                                                   uint32_t out_dim, uint32_t act, uint32_t out_act, uint32_t y_stride,
                                                   float* __restrict__ y0_exp, const float* __restrict__ sh_dirs = nullptr,
                                                   ShNorm4 nrm = ShNorm4{}) {
    extern __shared__ __attribute__((aligned(16))) float wl[];
    const int lane = lane_id(), j = lane & 31, h = lane >> 5;
    const uint32_t Bp = (B + 31u) & ~31u;
    // the first tile's inputs are requested before the weights are staged: one memory latency hidden behind the set-up
#ifdef ENERF_MLP_TIMING
    unsigned long long ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_last = __builtin_amdgcn_s_memtime();
#endif
    float x[16];
    {
        const uint32_t tile0 = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (tile0 < Bp / 32) load_x<XL>(X, tile0, j, h, B, Bp, x);
    }
    stage_rot(wl, W, NH, out_dim);
    MLP_PH(0);          // weights -> LDS

    float w0[2][16], wh[NH > 1 ? NH - 1 : 1][2][2][16], wo[2][16];
#pragma unroll
    for (int ob = 0; ob < 2; ob++)
#pragma unroll
        for (int p = 0; p < 16; p++) w0[ob][p] = wl[rot(32 * ob + j, kmap<XL>(p, h), IN)];
#pragma unroll
    for (int l = 0; l < NH - 1; l++)
#pragma unroll
        for (int ob = 0; ob < 2; ob++)
#pragma unroll
            for (int ib = 0; ib < 2; ib++)
#pragma unroll
                for (int q = 0; q < 16; q++)
                    wh[l][ob][ib][q] = wl[HID * IN + rot(l * HID + 32 * ob + j, 32 * ib + nrow(q, h), HID)];
    {
        const float* w64 = wl + HID * IN;                 // the 64-wide matrices; Wout's rows follow the hidden ones
        const uint32_t r0 = (NH - 1) * HID;
#pragma unroll
        for (int ib = 0; ib < 2; ib++)
#pragma unroll
            for (int q = 0; q < 16; q++)
                wo[ib][q] = SIG ? w64[rot(r0, 32 * ib + nrow(q, h), HID)]
                                : ((uint32_t)j < out_dim ? w64[rot(r0 + j, 32 * ib + nrow(q, h), HID)] : 0.0f);
    }

    MLP_PH(1);          // fragments -> registers
    const uint32_t ntiles = valid_tiles(W, B, Bp / 32);
    const uint32_t gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t nw = gridDim.x * (blockDim.x >> 6);
    for (uint32_t tile = gw; tile < ntiles; tile += nw) {
        const size_t s = (size_t)tile * 32 + j;
        const bool valid = s < B;
        if (!SIG && tile != gw) load_x<XL>(X, tile, j, h, B, Bp, x);
        float dir0 = 0.0f, dir1 = 0.0f, dir2 = 0.0f;            // requested now, used after the MFMAs
        if (SH && valid) {
            dir0 = sh_dirs[s * 3]; dir1 = sh_dirs[s * 3 + 1]; dir2 = sh_dirs[s * 3 + 2];
        }
        f32x16 a[2];
#pragma unroll
        for (int ob = 0; ob < 2; ob++) {
            a[ob] = (f32x16)(0.0f);
#pragma unroll
            for (int p = 0; p < 16; p++) a[ob] = mma(w0[ob][p], x[p], a[ob]);
        }
        // density-only sweeps (SIG: ~100 tiles per wavefront): the first layer has taken x, the same registers receive
        // the wave's next tile now, and the rest of this tile hides the latency
        if (SIG && tile + nw < ntiles) load_x<XL>(X, tile + nw, j, h, B, Bp, x);
#pragma unroll
        for (int ob = 0; ob < 2; ob++) {
#pragma unroll
            for (int q = 0; q < 16; q++) a[ob][q] = act_fwd(a[ob][q], act);
            if (TRAIN) store_tile_fb(fb + s * HID, ob, h, a[ob]);
        }
        MLP_PH(2);      // inputs + first layer
#pragma unroll
        for (int l = 1; l < NH; l++) {
            f32x16 n[2];
#pragma unroll
            for (int ob = 0; ob < 2; ob++) {
                n[ob] = (f32x16)(0.0f);
#pragma unroll
                for (int ib = 0; ib < 2; ib++)
#pragma unroll
                    for (int q = 0; q < 16; q++) n[ob] = mma(wh[l - 1][ob][ib][q], a[ib][q], n[ob]);
#pragma unroll
                for (int q = 0; q < 16; q++) n[ob][q] = act_fwd(n[ob][q], act);
                if (TRAIN) store_tile_fb(fb + ((size_t)l * Bp + s) * HID, ob, h, n[ob]);
            }
            a[0] = n[0];
            a[1] = n[1];
        }
        MLP_PH(3);      // hidden layers
        if (SIG) {
            float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
            for (int q = 0; q < 16; q++) {
                p0 = fmaf(wo[0][q], a[0][q], p0);
                p1 = fmaf(wo[1][q], a[1][q], p1);
            }
            float p = p0 + p1;
            p += __shfl_xor(p, 32, 64);
            if (valid && h == 0) y0_exp[s] = expf(p);
            continue;
        }
        f32x16 o = (f32x16)(0.0f);
#pragma unroll
        for (int ib = 0; ib < 2; ib++)
#pragma unroll
            for (int q = 0; q < 16; q++) o = mma(wo[ib][q], a[ib][q], o);
        if (valid) {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const uint32_t r = (uint32_t)nrow(q, h);
                if (Y && r < out_dim) Y[s * y_stride + r] = out_act_fwd(o[q], out_act);
                if (r == 0 && y0_exp) y0_exp[s] = expf(o[q]);      // trunc_exp forward of output column 0
            }
            if (SH) {
                float sh[16];
                sh4(dir0, dir1, dir2, nrm, sh);
                // lane half h stores components 8h .. 8h+7 (bit select: `h ? :` would index the array through scratch)
                const uint32_t m = 0u - (uint32_t)h;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; e++)
                    v[e] = __uint_as_float((__float_as_uint(sh[e]) & ~m) | (__float_as_uint(sh[8 + e]) & m));
                float* dst = Y + s * y_stride + 16 + 8 * h;
                *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
        }
        MLP_PH(4);      // output layer + stores
    }
#ifdef ENERF_MLP_TIMING
    MLP_PH(5);
    if (lane == 0)
        for (int k = 0; k < 6; k++) atomicAdd(&g_mlp_phase[k], ph[k]);
    if (threadIdx.x == 0) atomicAdd(&g_mlp_phase[15], 1ull);
#endif
}

template <int NH, int KPO, int XL>
__global__ void __launch_bounds__(256) k_mlp32_bwd_act(DySource dys, WSrc W,
                                                       const float* __restrict__ fb, float* __restrict__ bb,
                                                       float* __restrict__ dX, uint32_t B, uint32_t out_dim,
                                                       uint32_t act, float* __restrict__ dy_eff) {
    extern __shared__ __attribute__((aligned(16))) float wl[];
    stage(wl, W, blob_size(NH, out_dim));
    const int lane = lane_id(), j = lane & 31, h = lane >> 5;
    const float* wout = wl + HID * IN + (NH - 1) * HID * HID;
    const uint32_t Bp = (B + 31u) & ~31u;

    float woT[2][KPO], whT[NH > 1 ? NH - 1 : 1][2][2][16], wiT[2][16];
#pragma unroll
    for (int ib = 0; ib < 2; ib++)
#pragma unroll
        for (int p = 0; p < KPO; p++) {
            const uint32_t o = (uint32_t)(p + KPO * h);
            woT[ib][p] = o < out_dim ? wout[o * HID + 32 * ib + j] : 0.0f;
        }
#pragma unroll
    for (int l = 0; l < NH - 1; l++)
#pragma unroll
        for (int ib = 0; ib < 2; ib++)          // block of INPUT neurons of the layer (rows of the transposed operand)
#pragma unroll
            for (int ob = 0; ob < 2; ob++)
#pragma unroll
                for (int q = 0; q < 16; q++)
                    whT[l][ib][ob][q] = wl[HID * IN + l * HID * HID + (32 * ob + nrow(q, h)) * HID + 32 * ib + j];
#pragma unroll
    for (int ob = 0; ob < 2; ob++)
#pragma unroll
        for (int q = 0; q < 16; q++) wiT[ob][q] = wl[(32 * ob + nrow(q, h)) * IN + j];

    const uint32_t ntiles = Bp / 32;
    const uint32_t gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t nw = gridDim.x * (blockDim.x >> 6);
    for (uint32_t tile = gw; tile < ntiles; tile += nw) {
        const size_t s = (size_t)tile * 32 + j;
        const bool valid = s < B;
        float dy[KPO];
#pragma unroll
        for (int p = 0; p < KPO; p++) {
            const uint32_t o = (uint32_t)(p + KPO * h);
            dy[p] = (valid && o < out_dim) ? load_dy(dys, s, o) : 0.0f;
            // with a fused epilogue gradient, leave the effective dL/dY behind for the weight-gradient kernel
            if (dy_eff && valid && o < out_dim) dy_eff[s * out_dim + o] = dy[p];
        }
        f32x16 g[2];
#pragma unroll
        for (int ib = 0; ib < 2; ib++) {
            g[ib] = (f32x16)(0.0f);
#pragma unroll
            for (int p = 0; p < KPO; p++) g[ib] = mma(woT[ib][p], dy[p], g[ib]);
            f32x16 fw;
            load_tile_fb(fb + ((size_t)(NH - 1) * Bp + s) * HID, ib, h, fw);
#pragma unroll
            for (int q = 0; q < 16; q++) g[ib][q] = act_bwd(g[ib][q], fw[q], act);
            store_tile(bb + s * HID, ib, h, g[ib]);
        }
#pragma unroll
        for (int jj = 1; jj < NH; jj++) {
            const int l = NH - jj;
            f32x16 n[2];
#pragma unroll
            for (int ib = 0; ib < 2; ib++) {
                n[ib] = (f32x16)(0.0f);
#pragma unroll
                for (int ob = 0; ob < 2; ob++)
#pragma unroll
                    for (int q = 0; q < 16; q++) n[ib] = mma(whT[l - 1][ib][ob][q], g[ob][q], n[ib]);
                f32x16 fw;
                load_tile_fb(fb + ((size_t)(l - 1) * Bp + s) * HID, ib, h, fw);
#pragma unroll
                for (int q = 0; q < 16; q++) n[ib][q] = act_bwd(n[ib][q], fw[q], act);
                store_tile(bb + ((size_t)jj * Bp + s) * HID, ib, h, n[ib]);
            }
            g[0] = n[0];
            g[1] = n[1];
        }
        if (dX) {
            f32x16 d = (f32x16)(0.0f);
#pragma unroll
            for (int ob = 0; ob < 2; ob++)
#pragma unroll
                for (int q = 0; q < 16; q++) d = mma(wiT[ob][q], g[ob][q], d);
            if (XL == 0) {
                if (valid) {
#pragma unroll
                    for (int gq = 0; gq < 4; gq++)
                        *reinterpret_cast<float4*>(dX + s * IN + 8 * gq + 4 * h) =
                            make_float4(d[4 * gq], d[4 * gq + 1], d[4 * gq + 2], d[4 * gq + 3]);
                }
            } else {
                // registers 4gq .. 4gq+3 hold input columns 8gq + 4h + {0,1,2,3} = levels 4gq + 2h and 4gq + 2h + 1
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const size_t lv = (size_t)(4 * gq + 2 * h);
                    *reinterpret_cast<float2*>(dX + (lv * Bp + s) * 2) = make_float2(d[4 * gq], d[4 * gq + 1]);
                    *reinterpret_cast<float2*>(dX + ((lv + 1) * Bp + s) * 2) = make_float2(d[4 * gq + 2], d[4 * gq + 3]);
                }
            }
        }
    }
}

// ================================================================== backward: weight gradients
// dW[o][i] = sum_s dOut[s][o] * In[s][i]; MFMA p of a 32-sample tile contracts samples (2p, 2p+1): lane half h of the
// A operand reads row 2p+h of dOut (32 consecutive floats per half-wave), likewise the B operand of In.
// With a level-major X (XL 1) each wave first copies its tile's 16 x 256-byte level segments into LDS (row stride 66
// floats: the 32 lanes of a half-wave then read 32 distinct banks) and takes the B operand from there.
constexpr int XT_LD = 66;
template <int NH, int XL>
__global__ void __launch_bounds__(256, 2) k_mlp32_bwd_w(DySource dys, const float* __restrict__ X,
                                                     const float* __restrict__ fb, const float* __restrict__ bb,
                                                     float* __restrict__ partial, uint32_t B, uint32_t out_dim) {
    extern __shared__ __attribute__((aligned(16))) float red[];
    const uint32_t NW = blob_size(NH, out_dim);
    for (uint32_t i = threadIdx.x; i < NW; i += blockDim.x) red[i] = 0.0f;
    __syncthreads();
    const int lane = lane_id(), j = lane & 31, h = lane >> 5;
    const uint32_t Bp = (B + 31u) & ~31u;
    float* xt = red + ((NW + 3u) & ~3u) + (threadIdx.x >> 6) * (16 * XT_LD);

    f32x16 aw0[2], awh[NH > 1 ? NH - 1 : 1][2][2], awo[2];
#pragma unroll
    for (int a = 0; a < 2; a++) {
        aw0[a] = (f32x16)(0.0f);
        awo[a] = (f32x16)(0.0f);
#pragma unroll
        for (int l = 0; l < NH - 1; l++)
#pragma unroll
            for (int b = 0; b < 2; b++) awh[l][a][b] = (f32x16)(0.0f);
    }
    const uint32_t ntiles = Bp / 32;
    const uint32_t gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t nw = gridDim.x * (blockDim.x >> 6);
    for (uint32_t tile = gw; tile < ntiles; tile += nw) {
        const size_t s0 = (size_t)tile * 32;
        if (XL == 1) {
            // tile-local flat index f = level * 64 + 2 * sample + c; a lane moves four 16-byte pieces
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int f = t * 256 + lane * 4, lv = f >> 6, off = f & 63;
                const float4 v = *reinterpret_cast<const float4*>(X + ((size_t)lv * Bp + s0) * 2 + off);
                *reinterpret_cast<float2*>(xt + lv * XT_LD + off) = make_float2(v.x, v.y);
                *reinterpret_cast<float2*>(xt + lv * XT_LD + off + 2) = make_float2(v.z, v.w);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
#pragma unroll 4
        for (int p = 0; p < 16; p++) {
            const size_t s = s0 + 2 * p + h;      // < Bp; rows >= B of fb / bb / level-major X hold zeros
            // input layer
            float xin;
            if (XL == 0) xin = s < B ? X[s * IN + j] : 0.0f;
            else xin = xt[(j >> 1) * XT_LD + 2 * (2 * p + h) + (j & 1)];
            const float* g0 = bb + ((size_t)(NH - 1) * Bp + s) * HID;
#pragma unroll
            for (int ob = 0; ob < 2; ob++) aw0[ob] = mma(g0[32 * ob + j], xin, aw0[ob]);
            // hidden layers
#pragma unroll
            for (int m = 1; m < NH; m++) {
                const float* in = fb + ((size_t)(m - 1) * Bp + s0) * HID;          // tile-native block
                const float* go = bb + ((size_t)(NH - 1 - m) * Bp + s) * HID;
                const float i0 = fb_at(in, 2 * p + h, j), i1 = fb_at(in, 2 * p + h, 32 + j), o0 = go[j], o1 = go[32 + j];
                awh[m - 1][0][0] = mma(o0, i0, awh[m - 1][0][0]);
                awh[m - 1][0][1] = mma(o0, i1, awh[m - 1][0][1]);
                awh[m - 1][1][0] = mma(o1, i0, awh[m - 1][1][0]);
                awh[m - 1][1][1] = mma(o1, i1, awh[m - 1][1][1]);
            }
            // output layer
            const float* in = fb + ((size_t)(NH - 1) * Bp + s0) * HID;             // tile-native block
            const float dyv = ((uint32_t)j < out_dim && s < B) ? load_dy(dys, s, (uint32_t)j) : 0.0f;
            awo[0] = mma(dyv, fb_at(in, 2 * p + h, j), awo[0]);
            awo[1] = mma(dyv, fb_at(in, 2 * p + h, 32 + j), awo[1]);
        }
    }
    auto flush = [&](const f32x16& a, uint32_t base, int ld, int ob, int nb, uint32_t nrows) {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const uint32_t o = (uint32_t)(32 * ob + nrow(q, h));
            if (o < nrows) red[base + o * ld + 32 * nb + j] += a[q];
        }
    };
    const int wid = threadIdx.x >> 6;
    for (int turn = 0; turn < 4; turn++) {      // fixed order: deterministic sums
        if (wid == turn) {
#pragma unroll
            for (int ob = 0; ob < 2; ob++) flush(aw0[ob], 0, IN, ob, 0, HID);
#pragma unroll
            for (int l = 0; l < NH - 1; l++)
#pragma unroll
                for (int ob = 0; ob < 2; ob++)
#pragma unroll
                    for (int nb = 0; nb < 2; nb++) flush(awh[l][ob][nb], HID * IN + l * HID * HID, HID, ob, nb, HID);
#pragma unroll
            for (int nb = 0; nb < 2; nb++) flush(awo[nb], HID * IN + (NH - 1) * HID * HID, HID, 0, nb, out_dim);
        }
        __syncthreads();
    }
    float* dst = partial + (size_t)blockIdx.x * NW;
    for (uint32_t i = threadIdx.x; i < NW; i += blockDim.x) dst[i] = red[i];
}

// ================================================================== backward: dgrad + wgrad in one kernel
// The dgrad chain holds every operand of the weight gradients in registers already -- the activation gradients it
// computes and the forward activations it loads -- only in the "lane = sample" orientation, while
// dW[o][i] = sum_s G[o][s] In[i][s] contracts over samples.  Each 32x32 tile is therefore passed through LDS once
// (written from the accumulator layout as T[row][sample], row stride 33: both the writes and the operand reads
// T[lane][2p + h] are bank-conflict free) and consumed as A / B operands right there: the backward_buffer never
// exists in HBM (140 MB of traffic per 138 K samples for the colour net) and the activations are read once.
// One workgroup per CU (the tiles of 4 waves + the weights take ~110 KiB of LDS); per-workgroup partial sums are
// reduced by k_mlp32_reduce_w as before.  NH <= 2.
constexpr int T_LD = 33;
constexpr int T_SZ = 32 * T_LD;
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
__device__ __forceinline__ void tile_to_lds(float* t, int j, int h, const f32x16& v) {
#pragma unroll
    for (int q = 0; q < 16; q++) t[nrow(q, h) * T_LD + j] = v[q];
}

template <int NH, int KPO, int XL>
__global__ void __launch_bounds__(256) k_mlp32_bwd_fused(DySource dys, const float* __restrict__ X,
                                                         WSrc W, const float* __restrict__ fb,
                                                         float* __restrict__ dX, float* __restrict__ partial, uint32_t B,
                                                         uint32_t out_dim, uint32_t act) {
    constexpr uint32_t NW_MAX = HID * IN + (NH - 1) * HID * HID + 32 * HID;
    constexpr uint32_t PER_WAVE = 5 * T_SZ + (XL == 1 ? 16 * XT_LD : 32 * IN);
    __shared__ __attribute__((aligned(16))) float lds[NW_MAX + 4 * PER_WAVE];
    float* wl = lds;                                   // weights during set-up, block-level dW sums at the end
#ifdef ENERF_MLP_TIMING
    unsigned long long ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_last = __builtin_amdgcn_s_memtime();
#endif
    const uint32_t NW = blob_size(NH, out_dim);
    stage(wl, W, NW);
    const int lane = lane_id(), j = lane & 31, h = lane >> 5;
    const int wid = threadIdx.x >> 6;
    const float* wout = wl + HID * IN + (NH - 1) * HID * HID;
    const uint32_t Bp = (B + 31u) & ~31u;
    float* wv = lds + NW_MAX + wid * PER_WAVE;
    float* ga[2] = {wv, wv + T_SZ};                    // activation gradients of the current layer (A operands)
    float* ft[2] = {wv + 2 * T_SZ, wv + 3 * T_SZ};     // forward activations feeding it (B operands)
    float* dyt = wv + 4 * T_SZ;                        // dL/dY tile
    float* xt = wv + 5 * T_SZ;                         // X tile: level-major rows (XL == 1) or [32 samples][32] (XL == 0)

    float woT[2][KPO], whT[NH > 1 ? NH - 1 : 1][2][2][16], wiT[2][16];
#pragma unroll
    for (int ib = 0; ib < 2; ib++)
#pragma unroll
        for (int p = 0; p < KPO; p++) {
            const uint32_t o = (uint32_t)(p + KPO * h);
            woT[ib][p] = o < out_dim ? wout[o * HID + 32 * ib + j] : 0.0f;
        }
#pragma unroll
    for (int l = 0; l < NH - 1; l++)
#pragma unroll
        for (int ib = 0; ib < 2; ib++)
#pragma unroll
            for (int ob = 0; ob < 2; ob++)
#pragma unroll
                for (int q = 0; q < 16; q++)
                    whT[l][ib][ob][q] = wl[HID * IN + l * HID * HID + (32 * ob + nrow(q, h)) * HID + 32 * ib + j];
#pragma unroll
    for (int ob = 0; ob < 2; ob++)
#pragma unroll
        for (int q = 0; q < 16; q++) wiT[ob][q] = wl[(32 * ob + nrow(q, h)) * IN + j];

    f32x16 aw0[2], awh[NH > 1 ? NH - 1 : 1][2][2], awo[2];
#pragma unroll
    for (int a = 0; a < 2; a++) {
        aw0[a] = (f32x16)(0.0f);
        awo[a] = (f32x16)(0.0f);
#pragma unroll
        for (int l = 0; l < NH - 1; l++)
#pragma unroll
            for (int b = 0; b < 2; b++) awh[l][a][b] = (f32x16)(0.0f);
    }

    const uint32_t ntiles = Bp / 32;
    const uint32_t nreal = valid_tiles(W, B, ntiles);
    const uint32_t gw = blockIdx.x * (blockDim.x >> 6) + wid;
    const uint32_t nw = gridDim.x * (blockDim.x >> 6);
    // Everything a tile reads from global memory lives in the same registers from tile to tile: each group is
    // requested for the wave's NEXT tile at the point where the current tile has used it for the last time, so a whole
    // tile of work hides its latency and no extra registers are held.  The loads are plain (the compiler counts them:
    // no s_waitcnt vmcnt(0) anywhere in the loop) and branch-free (clamped addresses, the selection happens at use).
    float dy_raw[KPO], ys_raw[KPO], ds_raw = 0.0f, h0_raw = 0.0f;
    f32x16 fwl[NH][2];
    float4 xr0, xr1, xr2, xr3;                         // (named: as an array the compiler parks them in LDS)
    auto request_out = [&](uint32_t t) {               // dL/dY and the last hidden layer's activations of tile t
        const size_t sn = (size_t)t * 32 + j;
        const size_t sc = sn < B ? sn : (size_t)B - 1;
#pragma unroll
        for (int p = 0; p < KPO; p++) {
            const uint32_t o = (uint32_t)(p + KPO * h), oc = o < out_dim ? o : out_dim - 1;
#ifdef ENERF_MLP_NOLOAD
            dy_raw[p] = (float)(t & 3);
            ys_raw[p] = 0.5f;
#else
            dy_raw[p] = dys.dY[sc * dys.stride + oc];
            ys_raw[p] = dys.y_sig ? dys.y_sig[sc * dys.y_sig_stride + oc] : 0.0f;
#endif
        }
        if (dys.dsigma) {
            ds_raw = dys.dsigma[sc];
            h0_raw = dys.h0[sc * dys.h0_stride];
        }
#pragma unroll
        for (int ib = 0; ib < 2; ib++) {
#ifdef ENERF_MLP_NOLOAD
            fwl[NH - 1][ib] = (f32x16)((float)(t & 7) - 3.0f);
#else
            load_tile_fb(fb + ((size_t)(NH - 1) * Bp + sn) * HID, ib, h, fwl[NH - 1][ib]);
#endif
        }
    };
    auto request_hidden = [&](uint32_t t, int l) {     // activations of hidden layer l of tile t
        const size_t sn = (size_t)t * 32 + j;
#pragma unroll
        for (int ib = 0; ib < 2; ib++) {
#ifdef ENERF_MLP_NOLOAD
            fwl[l][ib] = (f32x16)((float)(t & 7) - 3.0f);
#else
            load_tile_fb(fb + ((size_t)l * Bp + sn) * HID, ib, h, fwl[l][ib]);
#endif
        }
    };
    auto x_addr = [&](uint32_t t, int q) -> const float4* {
        const size_t t0 = (size_t)t * 32;
        if (XL == 0) {
            // rows past the end of the batch: any finite row does (their activation gradients are zero)
            const size_t row = t0 + 8 * q + (lane >> 3), rc = row < B ? row : (size_t)B - 1;
            return reinterpret_cast<const float4*>(X + rc * IN + 4 * (lane & 7));
        }
        const int f = q * 256 + lane * 4, lv = f >> 6, off = f & 63;
        return reinterpret_cast<const float4*>(X + ((size_t)lv * Bp + t0) * 2 + off);
    };
    auto request_x = [&](uint32_t t) {                 // the tile's inputs, 1 KB per instruction
        xr0 = *x_addr(t, 0);
        xr1 = *x_addr(t, 1);
        xr2 = *x_addr(t, 2);
        xr3 = *x_addr(t, 3);
    };
    auto x_piece_to_lds = [&](int q, const float4& v) {
        if (XL == 0) {
            *reinterpret_cast<float4*>(xt + 256 * q + 4 * lane) = v;
        } else {
            const int f = q * 256 + lane * 4, lv = f >> 6, off = f & 63;
            *reinterpret_cast<float2*>(xt + lv * XT_LD + off) = make_float2(v.x, v.y);
            *reinterpret_cast<float2*>(xt + lv * XT_LD + off + 2) = make_float2(v.z, v.w);
        }
    };
    auto x_to_lds = [&]() {
        x_piece_to_lds(0, xr0);
        x_piece_to_lds(1, xr1);
        x_piece_to_lds(2, xr2);
        x_piece_to_lds(3, xr3);
    };
    {
        // (requests are never conditional: a branch around them would make the compiler copy the loaded registers, and
        // so wait for them, on the spot -- a wave without a further tile re-requests a tile it is allowed to read)
        const uint32_t t0 = gw < ntiles ? gw : ntiles - 1;
        request_out(t0);
#pragma unroll
        for (int l = NH - 2; l >= 0; l--) request_hidden(t0, l);
        request_x(t0);
    }
    for (uint32_t tile = gw; tile < ntiles; tile += nw) {
        const uint32_t tnext = tile + nw < nreal ? tile + nw : tile;      // the wave's next real tile (or this one again)
        const size_t s0 = (size_t)tile * 32;
        const size_t s = s0 + j;
        const bool valid = s < B;
        MLP_PH(0);      // set-up (first tile) / loop overhead
        if (tile >= nreal) {
            // padding rows: their upstream gradient is zero and nobody reads their activations -- the input gradient
            // is zero, the weight gradients get nothing
            if (dX) {
                if (XL == 0) {
                    if (valid) {
#pragma unroll
                        for (int gq = 0; gq < 4; gq++)
                            *reinterpret_cast<float4*>(dX + s * IN + 8 * gq + 4 * h) = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                } else {
#pragma unroll
                    for (int gq = 0; gq < 4; gq++) {
                        const size_t lv = (size_t)(4 * gq + 2 * h);
                        *reinterpret_cast<float2*>(dX + (lv * Bp + s) * 2) = make_float2(0.f, 0.f);
                        *reinterpret_cast<float2*>(dX + ((lv + 1) * Bp + s) * 2) = make_float2(0.f, 0.f);
                    }
                }
            }
            continue;
        }
        // ---- output layer
        float dy[KPO];
#pragma unroll
        for (int p = 0; p < KPO; p++) {
            const uint32_t o = (uint32_t)(p + KPO * h);
            float gq = dy_raw[p];                      // load_dy's arithmetic on the values requested a tile ago
            if (dys.y_sig) gq = (gq * (1.0f - ys_raw[p])) * ys_raw[p];
            if (dys.dsigma && o == 0) gq = ds_raw * expf(fminf(fmaxf(h0_raw, -15.0f), 15.0f));
            dy[p] = (valid && o < out_dim) ? gq : 0.0f;
        }
        f32x16 g[2];
        f32x16(&fw)[2] = fwl[NH - 1];
        wave_lds_fence();
#pragma unroll
        for (int p = 0; p < KPO; p++) dyt[(p + KPO * h) * T_LD + j] = dy[p];
#pragma unroll
        for (int ib = 0; ib < 2; ib++) {
            g[ib] = (f32x16)(0.0f);
#pragma unroll
            for (int p = 0; p < KPO; p++) g[ib] = mma(woT[ib][p], dy[p], g[ib]);
            tile_to_lds(ft[ib], j, h, fw[ib]);
#pragma unroll
            for (int q = 0; q < 16; q++) g[ib][q] = act_bwd(g[ib][q], fw[ib][q], act);
        }
        request_out(tnext);
        wave_lds_fence();
        MLP_PH(2);      // output layer dgrad (waits for dY and the last layer's activations)
        // dWout[o][i] += dY[o][s] * fb_last[i][s]
        // (operands of four contraction steps are read from LDS before their MFMAs are issued: left to itself the
        // compiler alternates one read, one wait, one MFMA, and the matrix pipe idles for an LDS latency each time)
#pragma unroll 1
        for (int p4 = 0; p4 < 16; p4 += 4) {
            float a[4], b0[4], b1[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int c = j * T_LD + 2 * (p4 + u) + h;
                a[u] = (uint32_t)j < 2u * KPO ? dyt[c] : 0.0f;
                b0[u] = ft[0][c];
                b1[u] = ft[1][c];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                awo[0] = mma(a[u], b0[u], awo[0]);
                awo[1] = mma(a[u], b1[u], awo[1]);
            }
        }
        wave_lds_fence();
        tile_to_lds(ga[0], j, h, g[0]);
        tile_to_lds(ga[1], j, h, g[1]);
        MLP_PH(3);      // dWout
        // ---- hidden layers
#pragma unroll
        for (int jj = 1; jj < NH; jj++) {
            const int l = NH - jj;
            f32x16 n[2];
            f32x16(&fw)[2] = fwl[l - 1];
#pragma unroll
            for (int ib = 0; ib < 2; ib++) {
                n[ib] = (f32x16)(0.0f);
#pragma unroll
                for (int ob = 0; ob < 2; ob++)
#pragma unroll
                    for (int q = 0; q < 16; q++) n[ib] = mma(whT[l - 1][ib][ob][q], g[ob][q], n[ib]);
            }
            MLP_PH(4);  // hidden dgrad
            wave_lds_fence();
            tile_to_lds(ft[0], j, h, fw[0]);
            tile_to_lds(ft[1], j, h, fw[1]);
            wave_lds_fence();
            // dWh[l-1][o][i] += G_l[o][s] * fb[l-1][i][s]
#pragma unroll 1
            for (int p4 = 0; p4 < 16; p4 += 4) {
                float a0[4], a1[4], b0[4], b1[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int c = j * T_LD + 2 * (p4 + u) + h;
                    a0[u] = ga[0][c]; a1[u] = ga[1][c];
                    b0[u] = ft[0][c]; b1[u] = ft[1][c];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    awh[l - 1][0][0] = mma(a0[u], b0[u], awh[l - 1][0][0]);
                    awh[l - 1][0][1] = mma(a0[u], b1[u], awh[l - 1][0][1]);
                    awh[l - 1][1][0] = mma(a1[u], b0[u], awh[l - 1][1][0]);
                    awh[l - 1][1][1] = mma(a1[u], b1[u], awh[l - 1][1][1]);
                }
            }
#pragma unroll
            for (int ib = 0; ib < 2; ib++)
#pragma unroll
                for (int q = 0; q < 16; q++) g[ib][q] = act_bwd(n[ib][q], fw[ib][q], act);
            request_hidden(tnext, l - 1);
            wave_lds_fence();
            tile_to_lds(ga[0], j, h, g[0]);
            tile_to_lds(ga[1], j, h, g[1]);
        }
        MLP_PH(5);      // hidden wgrad + activation mask
        // ---- input layer
        MLP_PH(6);      // X tile (level-major)
        wave_lds_fence();
        x_to_lds();
        request_x(tnext);
        wave_lds_fence();
        // dW0[o][i] += G_0[o][s] * X[s][i]
#pragma unroll 1
        for (int p4 = 0; p4 < 16; p4 += 4) {
            float xin[4], a0[4], a1[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int p = p4 + u;
                if (XL == 0) xin[u] = xt[(2 * p + h) * IN + j];
                else xin[u] = xt[(j >> 1) * XT_LD + 2 * (2 * p + h) + (j & 1)];
                a0[u] = ga[0][j * T_LD + 2 * p + h];
                a1[u] = ga[1][j * T_LD + 2 * p + h];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                aw0[0] = mma(a0[u], xin[u], aw0[0]);
                aw0[1] = mma(a1[u], xin[u], aw0[1]);
            }
        }
        MLP_PH(7);      // dW0
        if (dX) {
            f32x16 d = (f32x16)(0.0f);
#pragma unroll
            for (int ob = 0; ob < 2; ob++)
#pragma unroll
                for (int q = 0; q < 16; q++) d = mma(wiT[ob][q], g[ob][q], d);
            if (XL == 0) {
                if (valid) {
#pragma unroll
                    for (int gq = 0; gq < 4; gq++)
                        *reinterpret_cast<float4*>(dX + s * IN + 8 * gq + 4 * h) =
                            make_float4(d[4 * gq], d[4 * gq + 1], d[4 * gq + 2], d[4 * gq + 3]);
                }
            } else {
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const size_t lv = (size_t)(4 * gq + 2 * h);
                    *reinterpret_cast<float2*>(dX + (lv * Bp + s) * 2) = make_float2(d[4 * gq], d[4 * gq + 1]);
                    *reinterpret_cast<float2*>(dX + ((lv + 1) * Bp + s) * 2) = make_float2(d[4 * gq + 2], d[4 * gq + 3]);
                }
            }
        }
        MLP_PH(8);      // input dgrad + dX stores
    }

    // block-level sums, then one partial per workgroup: every wave spreads its accumulators over a region of its own
    // (the tile buffers are free by now: 4 * NW floats fit in the kernel's LDS for the NH <= 2 this kernel serves), all
    // four at once, and the sums are taken in wave order (deterministic) on the way out -- taking turns at one
    // shared region cost four serialised passes of 128 read-modify-writes per lane
    static_assert(4 * NW_MAX <= NW_MAX + 4 * PER_WAVE, "the four per-wave regions must fit");
    __syncthreads();                                   // nobody reads a tile buffer or the staged weights any more
    float* red = lds + (size_t)wid * NW;
    auto flush = [&](const f32x16& a, uint32_t base, int ld, int ob, int nb, uint32_t nrows) {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const uint32_t o = (uint32_t)(32 * ob + nrow(q, h));
            if (o < nrows) red[base + o * ld + 32 * nb + j] = a[q];
        }
    };
#pragma unroll
    for (int ob = 0; ob < 2; ob++) flush(aw0[ob], 0, IN, ob, 0, HID);
#pragma unroll
    for (int l = 0; l < NH - 1; l++)
#pragma unroll
        for (int ob = 0; ob < 2; ob++)
#pragma unroll
            for (int nb = 0; nb < 2; nb++) flush(awh[l][ob][nb], HID * IN + l * HID * HID, HID, ob, nb, HID);
#pragma unroll
    for (int nb = 0; nb < 2; nb++) flush(awo[nb], HID * IN + (NH - 1) * HID * HID, HID, 0, nb, out_dim);
    __syncthreads();
    float* dst = partial + (size_t)blockIdx.x * NW;
    for (uint32_t i = threadIdx.x; i < NW; i += blockDim.x)
        dst[i] = ((lds[i] + lds[NW + i]) + lds[2 * NW + i]) + lds[3 * NW + i];
#ifdef ENERF_MLP_TIMING
    MLP_PH(10);
    if (lane == 0)
        for (int k = 0; k < 11; k++) atomicAdd(&g_mlp_phase[k], ph[k]);
    if (threadIdx.x == 0) atomicAdd(&g_mlp_phase[15], 1ull);
#endif
}

__global__ void k_signal_mark() {}

// common.h PartialSums.map for the two gradient blobs of csrc/nerf_mlp.hip's backward: whose element the reduce pass would
// write sum i to
struct SmallDst {
    const float* g[5];
    uint32_t n[5];
};
__global__ void __launch_bounds__(256) k_nerf_partial_map(WDst ds, WDst dc, uint32_t nw_s, uint32_t nw_c, SmallDst sm,
                                                          uint32_t* __restrict__ map) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nw_s + nw_c) return;
    const float* dst = i < nw_s ? wdst_at(ds, i) : wdst_at(dc, i - nw_s);
    uint32_t code = 0xffffffffu;
    if (dst) {
#pragma unroll
        for (uint32_t k = 0; k < 5; k++)
            if (dst >= sm.g[k] && dst < sm.g[k] + sm.n[k]) code = (k << 24) | (uint32_t)(dst - sm.g[k]);
    }
    map[i] = code;
}

struct ReduceJob {
    const float* partial;
    uint32_t nblocks, NW;
    WDst dst;
    uint32_t stride;          // floats between two workgroups' partial sums (0: NW)
};

__device__ __forceinline__ void reduce_w_body(const float* __restrict__ partial, uint32_t nblocks, uint32_t NW,
                                              const WDst& gw, uint32_t blk, uint32_t* found_inf, uint32_t stride = 0) {
    const size_t ST = stride ? stride : NW;
    // 64 weights per workgroup; each of the 16 waves sums every 16th partial block with four independent chains
    // (fixed order), then the waves' sums are combined in a fixed order: deterministic.
    __shared__ float acc[16][64];
    const uint32_t i = blk * 64 + (threadIdx.x & 63);
    const uint32_t part = threadIdx.x >> 6;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (i < NW) {
        uint32_t b = part;
        for (; b + 48 < nblocks; b += 64) {
            s0 += partial[(size_t)b * ST + i];
            s1 += partial[(size_t)(b + 16) * ST + i];
            s2 += partial[(size_t)(b + 32) * ST + i];
            s3 += partial[(size_t)(b + 48) * ST + i];
        }
        for (; b < nblocks; b += 16) s0 += partial[(size_t)b * ST + i];
    }
    acc[part][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (part == 0 && i < NW) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < 16; w++) t += acc[w][threadIdx.x];
        float* dst = wdst_at(gw, i);
        if (dst) *dst = gw.overwrite ? t : *dst + t;
        // loss scaling (enerf_amp_begin): an activation gradient that left fp16's range shows up here as inf / NaN
        if (found_inf && !(fabsf(t) <= 3.402823466e38f)) atomicOr(found_inf, 1u);
    }
}

__global__ void __launch_bounds__(1024) k_mlp32_reduce_w(const float* __restrict__ partial, uint32_t nblocks, uint32_t NW,
                                                         WDst gw, uint32_t* found_inf) {
    reduce_w_body(partial, nblocks, NW, gw, blockIdx.x, found_inf);
}

// two networks' weight gradients in one launch (the first job was left pending by enerf_mlp32_defer_reduce)
__global__ void __launch_bounds__(1024) k_mlp32_reduce_w2(ReduceJob a, ReduceJob b, uint32_t* found_inf) {
    const uint32_t na = (a.NW + 63u) / 64u;
    if (blockIdx.x < na) reduce_w_body(a.partial, a.nblocks, a.NW, a.dst, blockIdx.x, found_inf, a.stride);
    else reduce_w_body(b.partial, b.nblocks, b.NW, b.dst, blockIdx.x - na, found_inf, b.stride);
}

static const int32_t* g_valid_rows = nullptr;      // enerf_mlp32_valid_rows
static uint32_t g_valid_base = 0, g_valid_cap = 0;  // enerf_mlp32_valid_rows_ex
static bool g_signal_armed = false;      // enerf_mlp32_signal_next_reduce
static bool g_signal_recorded = false;
static hipEvent_t g_signal_event = nullptr;
// (device-scope release: what waits for it is another stream of this device, never the host)
#ifndef ENERF_SIGNAL_SYSTEM_RELEASE
constexpr unsigned kSignalEventFlags = hipEventDisableTiming | hipEventReleaseToDevice;
#else
constexpr unsigned kSignalEventFlags = hipEventDisableTiming;
#endif
static bool g_defer_next = false;        // one-shot: set by enerf_mlp32_defer_reduce
static bool g_have_pending = false;
static ReduceJob g_pending;

bool g_fused_bwd = true;            // dgrad + wgrad in one kernel (num_hidden <= 2)
int g_precision = 1;                // enerf_mlp32_precision: 0 = fp32 MFMA (bit-exact fmaf chains), 1 = split-bf16 (x3),
                                    // 2 = bf16 operands (the FFMLP nets' arithmetic: one product, 16-bit roundings),
                                    // 3 = fp16 operands (the same kernels on IEEE half: the reference's fp16 regime)
inline bool ops16() { return g_precision == 2 || g_precision == 3; }
bool g_io16 = false;                // ffmlp16_forward / _backward: X, Y, dY, dX are 16-bit row-major tensors
bool g_recompute = true;            // enerf_mlp32_recompute: the split backward recomputes the hidden activations
// three hidden layers (the FFMLP colour net on 16-bit operands) recompute whatever the switch says: mlp32s.hip has no
// activation-loading instance of that shape
inline bool recompute_for(uint32_t num_hidden) { return g_recompute || num_hidden == 3; }
uint32_t g_wgrad_blocks = 0;        // 0: 768 workgroups for one hidden layer, 512 otherwise (measured optimum)

uint32_t g_fwd_blocks = 0;          // 0: default cap of the forward grid
uint32_t g_bwd_blocks = 0;          // 0: default cap of the fused backward grid

// shapes the split / bf16 backward kernel serves: one or two hidden layers in either mode, three (the FFMLP colour net,
// row-major input) with bf16 operands
bool split_bwd_shape(uint32_t num_hidden, uint32_t out_dim, uint32_t x_layout) {
    return g_fused_bwd && g_precision != 0 && out_dim <= 16 &&
           (num_hidden <= 2 || (ops16() && num_hidden == 3 && x_layout == 0));
}

uint32_t pgrid(uint32_t B, uint32_t cap) {
    const uint32_t blocks = div_up(div_up(B, 32), 4);
    return blocks < cap ? blocks : cap;
}

}  // namespace

extern "C" {

// Applies to the mlp32 forward / backward calls that follow, until set again (NULL: every row is real).
int enerf_mlp32_valid_rows(const int32_t* device_count) {
    g_valid_rows = device_count;
    g_valid_base = g_valid_cap = 0;
    return 0;
}
// real rows = base + min(*device_count, cap)  (cap == 0: as enerf_mlp32_valid_rows)
int enerf_mlp32_valid_rows_ex(const int32_t* device_count, uint32_t base, uint32_t cap) {
    g_valid_rows = device_count;
    g_valid_base = device_count ? base : 0;
    g_valid_cap = device_count ? cap : 0;
    return 0;
}

// Arithmetic of the mlp32 kernels: 0 = v_mfma_f32_32x32x2_f32 (every dot product an fp32 fmaf chain, bit-comparable
// with an fp32 GEMM), 1 (default) = split-bf16: operands carried as bf16 hi + lo, three bf16 MFMA products per fp32
// product, fp32 accumulation (~2^-16 relative per product), 2 = bf16 operands: inputs, weights and every layer's
// activations / activation gradients rounded to bf16, one product, fp32 accumulation, outputs rounded to bf16 -- the
// arithmetic of the FFMLP nets (ffmlp.hip) on fp32 buffers, for the fused training step of nerf/network_ff.py.
// Returns the previous mode; mode < 0 only queries.
int enerf_mlp32_precision(int mode) {
    const int prev = g_precision;
    if (mode >= 0 && mode <= 3) g_precision = mode;
    return prev;
}

// 1 (default): in the split / bf16 modes the backward kernel recomputes the hidden activations from X (bit-identical to
// the forward's) and the training forward does not write them: `fb` is then neither written nor read (it may be NULL
// in the backward; the forward still takes "fb != NULL" as "training").  0: the forward stores them, the backward loads
// them.  Forward and backward of a batch must run under the same setting.  Returns the previous setting; on < 0 queries.
int enerf_mlp32_recompute(int on) {
    const int prev = g_recompute ? 1 : 0;
    if (on >= 0) g_recompute = on != 0;
    return prev;
}

// testing aid: 1 (default) = fused dgrad + wgrad kernel for num_hidden <= 2, 0 = separate dgrad / wgrad kernels
int enerf_debug_mlp32_fused_backward(int on) {
    g_fused_bwd = on != 0;
    return 0;
}

// tuning aids: workgroup caps of the forward / fused backward grids (0 = defaults)
int enerf_debug_mlp32_grid_caps(uint32_t fwd_blocks, uint32_t bwd_blocks) {
    g_fwd_blocks = fwd_blocks;
    g_bwd_blocks = bwd_blocks;
    return 0;
}

// tuning aid: number of workgroups (= partial sums) of the weight-gradient kernel
int enerf_debug_mlp32_wgrad_blocks(uint32_t blocks) {
    g_wgrad_blocks = blocks;
    return 0;
}

// Fused fp32 MLP: X -> (64 x num_hidden, ReLU/none) -> Y [B,out_dim], out_dim <= 32, no bias.  B is ragged; with
// Bp = B rounded up to 32: X is [B,32] row-major (x_layout 0) or [16,Bp,2] level-major (x_layout 1);
// weights: [W0 64x32 | Wh (num_hidden-1) x 64x64 | Wout out_dim x 64]; fb: num_hidden*Bp*64 floats of forward
// activations in this file's own tile order (opaque to the caller; only enerf_mlp32_backward reads it) or NULL (inference).

static WSrc blob_src(const float* W, uint32_t num_hidden) {
    WSrc w;
    w.seg[0] = W;
    w.seg[1] = num_hidden > 1 ? W + HID * IN : nullptr;
    w.seg[2] = num_hidden > 2 ? W + HID * IN + HID * HID : nullptr;
    w.seg[3] = W + HID * IN + (num_hidden - 1) * HID * HID;
    w.w0_cols = IN;
    w.nerf_perm = 0;
    w.valid_rows = g_valid_rows;
    w.valid_base = g_valid_base;
    w.valid_cap = g_valid_cap;
    return w;
}
static int segs_ok(const void* const* seg, uint32_t num_hidden, uint32_t w0_cols, uint32_t nerf_perm, const char* what) {
    if (!seg || !seg[0] || !seg[3] || (num_hidden > 1 && !seg[1]) || (num_hidden > 2 && !seg[2])) {
        set_error("%s: a weight segment pointer is missing", what);
        return ENERF_E_BADARG;
    }
    if ((nerf_perm && w0_cols != 31 && w0_cols != IN) || (!nerf_perm && w0_cols != IN)) {
        set_error("%s: first-layer rows are 32 floats, or 31 / 32 with the NeRF colour permutation", what);
        return ENERF_E_BADARG;
    }
    return 0;
}

static int mlp32_forward_impl(const float* X, WSrc W, uint32_t B, uint32_t in_dim, uint32_t out_dim,
                              uint32_t num_hidden, uint32_t activation, uint32_t output_activation, float* fb, float* Y,
                              uint32_t x_layout, uint32_t y_stride, float* y0_exp, const float* sh_dirs,
                              enerf_stream_t stream) {
    if (B == 0) return 0;
    if (sh_dirs && !((num_hidden == 1 || (num_hidden == 2 && g_precision != 0)) && x_layout == 1 && Y && out_dim <= 16 &&
                     (y_stride == 0 ? out_dim : y_stride) >= 32))
        ENERF_BADARG("mlp32_forward_sh: needs one hidden layer (two in the bf16 modes), level-major input, out_dim <= 16 "
                     "and rows of >= 32 floats");
    if (in_dim != IN) ENERF_BADARG("mlp32: in_dim must be 32 (pad the input), got %u", in_dim);
    if (out_dim == 0 || out_dim > 32) ENERF_BADARG("mlp32: out_dim must be in [1, 32], got %u", out_dim);
    if (num_hidden < 1 || num_hidden > 3) ENERF_BADARG("mlp32: num_hidden must be 1..3, got %u", num_hidden);
    if (activation != 0 && activation != 6) ENERF_BADARG("mlp32: activation must be relu (0) or none (6)");
    if (x_layout > 1) ENERF_BADARG("mlp32: x_layout must be 0 (row-major) or 1 (level-major), got %u", x_layout);
    if (output_activation != 0 && output_activation != 3 && output_activation != 6)
        ENERF_BADARG("mlp32: output activation must be relu (0), sigmoid (3) or none (6)");
    if (y_stride == 0) y_stride = out_dim;
    if (y_stride < out_dim) ENERF_BADARG("mlp32: y_stride %u < out_dim %u", y_stride, out_dim);
    if (int eg = single_device_guard("mlp32_forward")) return eg;
    hipStream_t s = (hipStream_t)stream;
    // (the split kernels are timed by their own begin / end stamps; the fp32 MFMA kernels between two event packets)
    ProfScope prof(ENERF_K_FFMLP_FWD, s, g_precision != 0);
    prof.units((double)B);
    // two workgroups per CU are resident (the weights sit in ~130-210 registers): one round of them, each wave
    // setting up once, beats four short-lived ones per CU (measured at the 138 k-sample training batch)
    // (split operands: the weights sit in LDS as operands -- 2 KiB per fragment, hi + lo -- over the staged fp32 copy,
    // and three workgroups per CU are resident)
    const bool sigma_only_shape = num_hidden == 1 && !fb && !Y && y0_exp && x_layout == 1;
    const bool lds_operands = g_precision == 1 && !sigma_only_shape;
    const uint32_t grid = pgrid(B, g_fwd_blocks ? g_fwd_blocks : (lds_operands ? 768 : 512));
    size_t lds = sizeof(float) * (HID * IN + (num_hidden - 1) * HID * HID + out_dim * HID);
    if (lds_operands && lds < (size_t)(8 + 8 * (num_hidden - 1)) * 2048) lds = (size_t)(8 + 8 * (num_hidden - 1)) * 2048;
#define MLP32_FWD2(NHV, TR, XLV) \
    k_mlp32_fwd<NHV, TR, XLV><<<grid, 256, lds, s>>>(X, W, fb, Y, B, out_dim, activation, output_activation, y_stride, y0_exp)
#define MLP32_FWD(NHV)                                        \
    do {                                                      \
        if (fb) {                                             \
            if (x_layout == 0) MLP32_FWD2(NHV, true, 0);      \
            else MLP32_FWD2(NHV, true, 1);                    \
        } else {                                              \
            if (x_layout == 0) MLP32_FWD2(NHV, false, 0);     \
            else MLP32_FWD2(NHV, false, 1);                   \
        }                                                     \
    } while (0)
    const bool sigma_only = num_hidden == 1 && !fb && !Y && y0_exp && x_layout == 1;
    if (g_precision != 0) {
        // (a training forward whose backward recomputes the activations is the inference kernel)
        const bool store_fb = fb != nullptr && !(recompute_for(num_hidden) && split_bwd_shape(num_hidden, out_dim, x_layout));
        (g_precision == 3 ? mlp32s_f16_launch_fwd : mlp32s_launch_fwd)(
            g_precision == 1 ? 3 : 1, num_hidden, store_fb, x_layout, sigma_only, X, W, fb, Y, B, out_dim, activation,
            output_activation, y_stride, y0_exp, sh_dirs, grid, lds, s, prof.start(), prof.stop(), g_io16);
    } else if (sh_dirs) {
        const ShNorm4 nrm = make_sh_norm4();
        if (fb)
            k_mlp32_fwd<1, true, 1, false, true><<<grid, 256, lds, s>>>(X, W, fb, Y, B, out_dim, activation,
                                                                        output_activation, y_stride, y0_exp, sh_dirs, nrm);
        else
            k_mlp32_fwd<1, false, 1, false, true><<<grid, 256, lds, s>>>(X, W, fb, Y, B, out_dim, activation,
                                                                         output_activation, y_stride, y0_exp, sh_dirs, nrm);
    } else if (sigma_only)
        k_mlp32_fwd<1, false, 1, true><<<grid, 256, lds, s>>>(X, W, fb, Y, B, out_dim, activation, output_activation,
                                                              y_stride, y0_exp);
    else if (num_hidden == 1) MLP32_FWD(1);
    else if (num_hidden == 2) MLP32_FWD(2);
    else MLP32_FWD(3);
#undef MLP32_FWD
#undef MLP32_FWD2
    ENERF_LAUNCH_CHECK("mlp32_forward");
    return 0;
}

int enerf_mlp32_forward(const float* X, const float* W, uint32_t B, uint32_t in_dim, uint32_t out_dim,
                        uint32_t num_hidden, uint32_t activation, uint32_t output_activation, float* fb, float* Y,
                        uint32_t x_layout, uint32_t y_stride, float* y0_exp, enerf_stream_t stream) {
    if (num_hidden < 1 || num_hidden > 3) ENERF_BADARG("mlp32: num_hidden must be 1..3, got %u", num_hidden);
    return mlp32_forward_impl(X, blob_src(W, num_hidden), B, in_dim, out_dim, num_hidden, activation, output_activation,
                              fb, Y, x_layout, y_stride, y0_exp, nullptr, stream);
}

int enerf_mlp32_forward_sh(const float* X, const float* W, uint32_t B, uint32_t in_dim, uint32_t out_dim,
                           uint32_t num_hidden, uint32_t activation, uint32_t output_activation, float* fb, float* Y,
                           uint32_t x_layout, uint32_t y_stride, float* y0_exp, const float* sh_dirs,
                           enerf_stream_t stream) {
    if (num_hidden < 1 || num_hidden > 3) ENERF_BADARG("mlp32: num_hidden must be 1..3, got %u", num_hidden);
    return mlp32_forward_impl(X, blob_src(W, num_hidden), B, in_dim, out_dim, num_hidden, activation, output_activation,
                              fb, Y, x_layout, y_stride, y0_exp, sh_dirs, stream);
}

// The same forward with the weights where the caller keeps them: wseg = {first layer, hidden 0, hidden 1, output layer}
// (unused hidden slots NULL); w0_cols / nerf_perm as in WSrc above.
int enerf_mlp32_forward_p(const float* X, const float* const* wseg, uint32_t w0_cols, uint32_t nerf_perm, uint32_t B,
                          uint32_t in_dim, uint32_t out_dim, uint32_t num_hidden, uint32_t activation,
                          uint32_t output_activation, float* fb, float* Y, uint32_t x_layout, uint32_t y_stride,
                          float* y0_exp, const float* sh_dirs, enerf_stream_t stream) {
    if (num_hidden < 1 || num_hidden > 3) ENERF_BADARG("mlp32: num_hidden must be 1..3, got %u", num_hidden);
    if (int e = segs_ok((const void* const*)wseg, num_hidden, w0_cols, nerf_perm, "mlp32_forward_p")) return e;
    WSrc w;
    for (int k = 0; k < 4; k++) w.seg[k] = wseg[k];
    if (num_hidden < 3) w.seg[2] = nullptr;
    if (num_hidden < 2) w.seg[1] = nullptr;
    w.w0_cols = w0_cols;
    w.nerf_perm = nerf_perm;
    w.valid_rows = g_valid_rows;
    w.valid_base = g_valid_base;
    w.valid_cap = g_valid_cap;
    return mlp32_forward_impl(X, w, B, in_dim, out_dim, num_hidden, activation, output_activation, fb, Y, x_layout,
                              y_stride, y0_exp, sh_dirs, stream);
}

// dY [B,out_dim], fb from the forward; bb [num_hidden,Bp,64] scratch (written); dX NULL, [B,32] (x_layout 0) or
// [16,Bp,2] (x_layout 1, pad rows written as zeros); dW (fp32 blob) is ACCUMULATED into (+=).
static int mlp32_backward_impl(const float* dY, const float* X, WSrc W, const float* fb, uint32_t B, uint32_t in_dim,
                               uint32_t out_dim, uint32_t num_hidden, uint32_t activation, float* bb, float* dX,
                               WDst dW, uint32_t x_layout, uint32_t dy_stride, const float* y_sigmoid,
                               uint32_t y_sigmoid_stride, const float* dsigma, const float* h0, uint32_t h0_stride,
                               enerf_stream_t stream) {
    if (B == 0) return 0;
    if (in_dim != IN) ENERF_BADARG("mlp32: in_dim must be 32, got %u", in_dim);
    if (out_dim == 0 || out_dim > 32) ENERF_BADARG("mlp32: out_dim must be in [1, 32], got %u", out_dim);
    if (num_hidden < 1 || num_hidden > 3) ENERF_BADARG("mlp32: num_hidden must be 1..3, got %u", num_hidden);
    if (x_layout > 1) ENERF_BADARG("mlp32: x_layout must be 0 (row-major) or 1 (level-major), got %u", x_layout);
    if ((dsigma == nullptr) != (h0 == nullptr)) ENERF_BADARG("mlp32: dsigma and h0 go together");
    if (int eg = single_device_guard("mlp32_backward")) return eg;
    DySource dys;
    dys.dY = dY;
    dys.stride = dy_stride ? dy_stride : out_dim;
    dys.y_sig = y_sigmoid;
    dys.y_sig_stride = y_sigmoid_stride ? y_sigmoid_stride : out_dim;
    dys.dsigma = dsigma;
    dys.h0 = h0;
    dys.h0_stride = h0_stride ? h0_stride : 1u;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t NW = HID * IN + (num_hidden - 1) * HID * HID + out_dim * HID;
    const size_t lds = sizeof(float) * NW;
    const size_t lds_w = sizeof(float) * (((NW + 3u) & ~3u) + (x_layout == 1 ? 4 * 16 * XT_LD : 0));
    // the split / bf16 kernels: one or two hidden layers in either mode, three (the FFMLP colour net, row-major input) with
    // bf16 operands
    const bool split_bwd = split_bwd_shape(num_hidden, out_dim, x_layout);
    // timing (enerf_prof_*): the split kernel by its own begin / end stamps, the weight-gradient reduce launch as a family
    // of its own (ENERF_K_MLP_REDUCE); the fp32 MFMA route between two event packets around all of its launches
    ProfScope prof(ENERF_K_FFMLP_BWD, s, split_bwd);
    prof.units((double)B);
    if (ops16() && !split_bwd)
        ENERF_BADARG("mlp32_backward: 16-bit operands (precision 2 / 3) need out_dim <= 16 and at most three hidden layers");
    const bool fused = split_bwd || (g_fused_bwd && num_hidden <= 2);
    if (!fused && W.valid_rows)
        ENERF_BADARG("mlp32_backward: enerf_mlp32_valid_rows needs the fused backward (num_hidden <= 2)");
    const uint32_t grid = pgrid(B, 1024);
    const uint32_t wgrid = fused ? pgrid(B, g_bwd_blocks ? g_bwd_blocks : 256)
                                 : pgrid(B, g_wgrad_blocks ? g_wgrad_blocks : (num_hidden == 1 ? 768u : 512u));
    // fused epilogue gradients are evaluated once, by the dgrad kernel, which leaves the effective dL/dY in the
    // workspace for the weight-gradient kernel (whose inner loop is load-bound)
    const bool fused_dy = !fused && (y_sigmoid != nullptr || dsigma != nullptr || dys.stride != out_dim);
    const size_t part_bytes = sizeof(float) * (size_t)wgrid * NW;
    // one-shot deferral (enerf_mlp32_defer_reduce): this call's partial sums wait, in their own workspace, for the next
    // call's reduce launch
    const bool defer = g_defer_next && !g_have_pending;
    g_defer_next = false;
    if (int ew = workspace_family_enter(1, s)) return ew;
    float* partial = (float*)workspace(defer ? WS_MLP32_DEFER : WS_FFMLP,
                                       part_bytes + (fused_dy ? sizeof(float) * (size_t)B * out_dim : 0));
    if (!partial) return ENERF_E_NOMEM;
    float* dy_eff = fused_dy ? partial + (size_t)wgrid * NW : nullptr;
    DySource dys_w = dys;
    if (fused_dy) {
        dys_w.dY = dy_eff;
        dys_w.stride = out_dim;
        dys_w.y_sig = nullptr;
        dys_w.dsigma = nullptr;
        dys_w.h0 = nullptr;
    }
#define MLP32_BA(NHV, KPOV, XLV) \
    k_mlp32_bwd_act<NHV, KPOV, XLV><<<grid, 256, lds, s>>>(dys, W, fb, bb, dX, B, out_dim, activation, dy_eff)
#define MLP32_BWD2(NHV, XLV)                                                                      \
    do {                                                                                          \
        if (out_dim <= 4) MLP32_BA(NHV, 2, XLV);                                                  \
        else if (out_dim <= 16) MLP32_BA(NHV, 8, XLV);                                            \
        else MLP32_BA(NHV, 16, XLV);                                                              \
        k_mlp32_bwd_w<NHV, XLV><<<wgrid, 256, lds_w, s>>>(dys_w, X, fb, bb, partial, B, out_dim); \
    } while (0)
#define MLP32_BWD(NHV)                          \
    do {                                        \
        if (x_layout == 0) MLP32_BWD2(NHV, 0);  \
        else MLP32_BWD2(NHV, 1);                \
    } while (0)
#define MLP32_BF(NHV, KPOV, XLV) \
    k_mlp32_bwd_fused<NHV, KPOV, XLV><<<wgrid, 256, 0, s>>>(dys, X, W, fb, dX, partial, B, out_dim, activation)
#define MLP32_BF2(NHV, XLV)                      \
    do {                                         \
        if (out_dim <= 4) MLP32_BF(NHV, 2, XLV); \
        else if (out_dim <= 16) MLP32_BF(NHV, 8, XLV); \
        else MLP32_BF(NHV, 16, XLV);             \
    } while (0)
    if (split_bwd) {
        (void)bb;
        (g_precision == 3 ? mlp32s_f16_launch_bwd : mlp32s_launch_bwd)(
            g_precision == 1 ? 3 : 1, num_hidden, x_layout, dys, X, W, fb, dX, partial, B, out_dim, activation, wgrid, s,
            prof.start(), prof.stop(), recompute_for(num_hidden) || g_io16, g_io16);
    } else if (fused) {
        (void)bb;
        if (num_hidden == 1) { if (x_layout == 0) MLP32_BF2(1, 0); else MLP32_BF2(1, 1); }
        else { if (x_layout == 0) MLP32_BF2(2, 0); else MLP32_BF2(2, 1); }
    } else if (num_hidden == 1) MLP32_BWD(1);
    else if (num_hidden == 2) MLP32_BWD(2);
    else MLP32_BWD(3);
#undef MLP32_BF2
#undef MLP32_BF
#undef MLP32_BWD
#undef MLP32_BWD2
#undef MLP32_BA
    if (defer) {
        g_pending = ReduceJob{partial, wgrid, NW, dW, 0};
        g_have_pending = true;
    } else {
        // enerf_mlp32_signal_next_reduce: the reduce launch carries the signal event as its stop event -- a
        // kernel-attached completion signal costs the stream nothing, where an event record after the launch is a
        // packet of its own that the next kernel waits for (tools/launch_chain.hip: +0 against +3 us per kernel)
        hipEvent_t sig = nullptr;
        if (g_signal_armed) {
            g_signal_armed = false;
            // (tried and dropped, tools/step_timeline.py: a device-scope release event -- same ~6 us before the next
            // kernel of this stream; a generation number published by the reduce launch's last workgroup + a sleeping
            // wait kernel on the other stream -- the gap goes, the reduce launch grows by 3 us, the step does not move)
            if (!g_signal_event && hipEventCreateWithFlags(&g_signal_event, kSignalEventFlags) != hipSuccess)
                g_signal_event = nullptr;
            sig = g_signal_event;
            g_signal_recorded = sig != nullptr;
        }
        ProfScope prof_reduce(ENERF_K_MLP_REDUCE, s);
        if (g_have_pending) {
            g_have_pending = false;
            hipExtLaunchKernelGGL(k_mlp32_reduce_w2, dim3(div_up(g_pending.NW, 64) + div_up(NW, 64)), dim3(1024), 0, s,
                                  nullptr, sig, 0, g_pending, ReduceJob{partial, wgrid, NW, dW, 0}, amp_state().found_inf);
        } else {
            hipExtLaunchKernelGGL(k_mlp32_reduce_w, dim3(div_up(NW, 64)), dim3(1024), 0, s, nullptr, sig, 0, partial,
                                  wgrid, NW, dW, amp_state().found_inf);
        }
    }
    ENERF_LAUNCH_CHECK("mlp32_backward");
    return 0;
}

// ---- nerf/network.py's two nets as one launch per direction (csrc/nerf_mlp.hip) -------------------------------------
static bool g_nerf_fused = true;                 // enerf_debug_nerf_mlp_fused
static const float* g_nerf_built[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
static uint32_t g_nerf_built_cols = 0, g_nerf_built_out = 0;
static uint64_t g_nerf_built_gen = 0;
static uint32_t g_nerf_map_cols = 0, g_nerf_map_out = 0;      // geometry nerf_launch_frag_map's table was made for
static uint64_t g_nerf_map_gen = ~0ull;

// 1 when enerf_nerf_mlp_forward / _backward serve the process's current arithmetic (split-bf16, recomputing backward)
int enerf_nerf_mlp_available(void) { return (g_nerf_fused && g_precision == 1 && g_recompute && g_fused_bwd) ? 1 : 0; }
// testing aid: 0 = callers that ask enerf_nerf_mlp_available() fall back to one launch per net; returns the previous value
int enerf_debug_nerf_mlp_fused(int on) {
    const int prev = g_nerf_fused ? 1 : 0;
    if (on >= 0) g_nerf_fused = on != 0;
    return prev;
}

static int nerf_args_ok(const float* const* wseg_s, const float* const* wseg_c, uint32_t w0_cols_c, uint32_t out_c,
                        const char* what) {
    if (!wseg_s || !wseg_c || !wseg_s[0] || !wseg_s[3] || !wseg_c[0] || !wseg_c[1] || !wseg_c[3]) {
        set_error("%s: weight segments {first layer, hidden 0, -, output layer} of both nets are required", what);
        return ENERF_E_BADARG;
    }
    if (w0_cols_c != 31 && w0_cols_c != 32) {
        set_error("%s: the colour net's first-layer rows are 31 (or 32, padded) floats [SH 16 | geo_feat 15]", what);
        return ENERF_E_BADARG;
    }
    if (out_c == 0 || out_c > 16) {
        set_error("%s: colour outputs must be in [1, 16], got %u", what, out_c);
        return ENERF_E_BADARG;
    }
    if (g_precision != 1) {
        set_error("%s: serves the split-bf16 arithmetic (enerf_mlp32_precision 1) only; ask enerf_nerf_mlp_available()", what);
        return ENERF_E_BADARG;
    }
    return 0;
}

// the operand fragments of the five matrices (flags bit 0: the caller vouches that they were built for these very
// weights -- same pointers, values unchanged since -- by an earlier call; anything else rebuilds them: one 44-wave launch)
static uint32_t* nerf_frags(const float* const* wseg_s, const float* const* wseg_c, uint32_t w0_cols_c, uint32_t out_c,
                            uint32_t flags, hipStream_t s) {
    uint32_t* frags = (uint32_t*)workspace(WS_NERF_FRAGS, kNerfWsBytes);
    if (!frags) return nullptr;
    const float* w[5] = {wseg_s[0], wseg_s[3], wseg_c[0], wseg_c[1], wseg_c[3]};
    bool same = (flags & 1u) && g_nerf_built_cols == w0_cols_c && g_nerf_built_out == out_c &&
                g_nerf_built_gen == workspace_generation();
    for (int k = 0; k < 5; k++) same = same && g_nerf_built[k] == w[k];
    if (!same) {
        nerf_launch_frags(w[0], w[1], w[2], w[3], w[4], w0_cols_c, out_c, frags, s);
        for (int k = 0; k < 5; k++) g_nerf_built[k] = w[k];
        g_nerf_built_cols = w0_cols_c;
        g_nerf_built_out = out_c;
        g_nerf_built_gen = workspace_generation();
    }
    return frags;
}

// feats [16, Bp, 2] (grid_encode_forward's level-major layout 2), dirs [B, 3] -> sigma [B] = exp(h0), rgb [B, out_c] =
// sigmoid(colour net([h0 | geo_feat | SH(dirs)])).  Honours enerf_mlp32_valid_rows.
int enerf_nerf_mlp_forward(const float* feats, const float* dirs, const float* const* wseg_s, const float* const* wseg_c,
                           uint32_t w0_cols_c, uint32_t B, uint32_t out_c, float* sigma, float* rgb, uint32_t flags,
                           enerf_stream_t stream) {
    if (B == 0) return 0;
    if (int e = nerf_args_ok(wseg_s, wseg_c, w0_cols_c, out_c, "nerf_mlp_forward")) return e;
    if (!feats || !dirs || !sigma || !rgb) ENERF_BADARG("nerf_mlp_forward: feats, dirs, sigma and rgb are required");
    if (int eg = single_device_guard("nerf_mlp_forward")) return eg;
    hipStream_t s = (hipStream_t)stream;
    if (int ew = workspace_family_enter(1, s)) return ew;
    uint32_t* frags = nerf_frags(wseg_s, wseg_c, w0_cols_c, out_c, flags, s);
    if (!frags) return ENERF_E_NOMEM;
    ProfScope prof(ENERF_K_FFMLP_FWD, s, true);
    prof.units((double)B);
    nerf_launch_fwd(feats, dirs, frags, sigma, rgb, B, out_c, g_valid_rows, g_valid_base, g_valid_cap,
                    pgrid(B, g_fwd_blocks ? g_fwd_blocks : kNerfFwdPerCu * num_cus()), s, prof.start(), prof.stop());
    ENERF_LAUNCH_CHECK("nerf_mlp_forward");
    return 0;
}

// Gradients of both nets' weights (dwseg_*: overwritten when `overwrite`, else added to) and of feats (dfeat [16, Bp, 2])
// given g_rgb [B, out_c], g_sigma [B] (multiplied by sigma_scale on the fly) and the forward's rgb.  Honours
// enerf_mlp32_valid_rows and enerf_mlp32_signal_next_reduce (the reduce launch carries the signal).
int enerf_nerf_mlp_backward(const float* g_rgb, const float* g_sigma, float sigma_scale, const float* feats,
                            const float* dirs, const float* rgb, const float* const* wseg_s, const float* const* wseg_c,
                            float* const* dwseg_s, float* const* dwseg_c, uint32_t w0_cols_c, uint32_t overwrite,
                            uint32_t B, uint32_t out_c, float* dfeat, uint32_t flags, enerf_stream_t stream) {
    if (B == 0) return 0;
    if (int e = nerf_args_ok(wseg_s, wseg_c, w0_cols_c, out_c, "nerf_mlp_backward")) return e;
    if (!dwseg_s || !dwseg_c || !dwseg_s[0] || !dwseg_s[3] || !dwseg_c[0] || !dwseg_c[1] || !dwseg_c[3])
        ENERF_BADARG("nerf_mlp_backward: gradient segments of both nets are required");
    if (!g_rgb || !g_sigma || !feats || !dirs || !rgb || !dfeat)
        ENERF_BADARG("nerf_mlp_backward: g_rgb, g_sigma, feats, dirs, rgb and dfeat are required");
    if (int eg = single_device_guard("nerf_mlp_backward")) return eg;
    hipStream_t s = (hipStream_t)stream;
    if (int ew = workspace_family_enter(1, s)) return ew;
    uint32_t* frags = nerf_frags(wseg_s, wseg_c, w0_cols_c, out_c, flags, s);
    if (!frags) return ENERF_E_NOMEM;
    const uint32_t grid = pgrid(B, g_bwd_blocks ? g_bwd_blocks : 256);
    float* partial = (float*)workspace(WS_NERF_PART, sizeof(float) * (size_t)grid * kNerfPartialStride);
    if (!partial) return ENERF_E_NOMEM;
    hipEvent_t sig = nullptr;
    bool sig_sent = false;
    if (g_signal_armed) {
        g_signal_armed = false;
        if (!g_signal_event && hipEventCreateWithFlags(&g_signal_event, kSignalEventFlags) != hipSuccess)
            g_signal_event = nullptr;
        sig = g_signal_event;
        g_signal_recorded = sig != nullptr;
    }
    {
        ProfScope prof(ENERF_K_FFMLP_BWD, s, true);
        prof.units((double)B);
        // (flags bit 1: no reduce launch follows to carry the signal -- it leaves with this launch, unless a timed interval
        //  needs the launch's stop event)
        hipEvent_t stop = prof.stop();
        if ((flags & 2u) && sig && !stop) {
            stop = sig;
            sig_sent = true;
        }
        nerf_launch_bwd(feats, dirs, g_rgb, rgb, g_sigma, sigma_scale, frags, dfeat, partial, B, out_c, g_valid_rows,
                        g_valid_base, g_valid_cap, grid, s, prof.start(), stop);
    }
    WDst ds, dc;
    for (int k = 0; k < 4; k++) {
        ds.seg[k] = (k == 0 || k == 3) ? dwseg_s[k] : nullptr;
        dc.seg[k] = (k == 0 || k == 1 || k == 3) ? dwseg_c[k] : nullptr;
    }
    ds.w0_cols = IN; ds.nerf_perm = 0; ds.overwrite = overwrite;
    dc.w0_cols = w0_cols_c; dc.nerf_perm = 1; dc.overwrite = overwrite;
    if (flags & 2u) {
        // the caller sums the partial sums itself (enerf::nerf_mlp_partial_job): no reduce launch; the signal leaves on a
        // marker of its own
        if (!overwrite) ENERF_BADARG("nerf_mlp_backward: flags bit 1 serves overwrite != 0");
        if (sig && !sig_sent) hipExtLaunchKernelGGL(k_signal_mark, dim3(1), dim3(64), 0, s, nullptr, sig, 0);
        ENERF_LAUNCH_CHECK("nerf_mlp_backward");
        return 0;
    }
    const uint32_t nw_c = HID * IN + HID * HID + out_c * HID;
    ProfScope prof_reduce(ENERF_K_MLP_REDUCE, s);
    hipExtLaunchKernelGGL(k_mlp32_reduce_w2, dim3(div_up(kNerfSigmaWords, 64) + div_up(nw_c, 64)), dim3(1024), 0, s, nullptr,
                          sig, 0, ReduceJob{partial, grid, kNerfSigmaWords, ds, kNerfPartialStride},
                          ReduceJob{partial + kNerfSigmaWords, grid, nw_c, dc, kNerfPartialStride}, amp_state().found_inf);
    ENERF_LAUNCH_CHECK("nerf_mlp_backward");
    return 0;
}

int enerf_mlp32_signal_next_reduce(int on) {
    g_signal_armed = on != 0;
    if (on) g_signal_recorded = false;
    return 0;
}

int enerf_stream_wait_mlp32_signal(enerf_stream_t stream) {
    if (!g_signal_recorded || !g_signal_event)
        ENERF_BADARG("stream_wait_mlp32_signal: no reduce launch has carried the signal since it was armed");
    return check_hip(hipStreamWaitEvent((hipStream_t)stream, g_signal_event, 0), "stream_wait_mlp32_signal");
}

int enerf_mlp32_defer_reduce(int on) {
    g_defer_next = on != 0;
    g_have_pending = false;        // a pair always starts here: sums left behind by a pair that never completed are dropped
    return 0;
}

int enerf_mlp32_backward(const float* dY, const float* X, const float* W, const float* fb, uint32_t B, uint32_t in_dim,
                         uint32_t out_dim, uint32_t num_hidden, uint32_t activation, float* bb, float* dX, float* dW,
                         uint32_t x_layout, uint32_t dy_stride, const float* y_sigmoid, uint32_t y_sigmoid_stride,
                         const float* dsigma, const float* h0, uint32_t h0_stride, enerf_stream_t stream) {
    if (num_hidden < 1 || num_hidden > 3) ENERF_BADARG("mlp32: num_hidden must be 1..3, got %u", num_hidden);
    const WSrc w = blob_src(W, num_hidden);
    WDst d;
    for (int k = 0; k < 4; k++) d.seg[k] = w.seg[k] ? dW + (w.seg[k] - W) : nullptr;
    d.w0_cols = IN;
    d.nerf_perm = 0;
    d.overwrite = 0;
    return mlp32_backward_impl(dY, X, w, fb, B, in_dim, out_dim, num_hidden, activation, bb, dX, d, x_layout, dy_stride,
                               y_sigmoid, y_sigmoid_stride, dsigma, h0, h0_stride, stream);
}

// The backward with weights and weight gradients where the caller keeps them (see enerf_mlp32_forward_p); overwrite != 0:
// dwseg receives the gradient (no zero-filled accumulator needed), else it is added to.
int enerf_mlp32_backward_p(const float* dY, const float* X, const float* const* wseg, float* const* dwseg,
                           uint32_t w0_cols, uint32_t nerf_perm, uint32_t overwrite, const float* fb, uint32_t B,
                           uint32_t in_dim, uint32_t out_dim, uint32_t num_hidden, uint32_t activation, float* bb,
                           float* dX, uint32_t x_layout, uint32_t dy_stride, const float* y_sigmoid,
                           uint32_t y_sigmoid_stride, const float* dsigma, const float* h0, uint32_t h0_stride,
                           enerf_stream_t stream) {
    if (num_hidden < 1 || num_hidden > 3) ENERF_BADARG("mlp32: num_hidden must be 1..3, got %u", num_hidden);
    if (int e = segs_ok((const void* const*)wseg, num_hidden, w0_cols, nerf_perm, "mlp32_backward_p")) return e;
    if (int e = segs_ok((const void* const*)dwseg, num_hidden, w0_cols, nerf_perm, "mlp32_backward_p (gradients)")) return e;
    WSrc w;
    WDst d;
    for (int k = 0; k < 4; k++) {
        const bool used = k == 0 || k == 3 || (uint32_t)k < num_hidden;
        w.seg[k] = used ? wseg[k] : nullptr;
        d.seg[k] = used ? dwseg[k] : nullptr;
    }
    w.w0_cols = d.w0_cols = w0_cols;
    w.nerf_perm = d.nerf_perm = nerf_perm;
    w.valid_rows = g_valid_rows;
    w.valid_base = g_valid_base;
    w.valid_cap = g_valid_cap;
    d.overwrite = overwrite;
    return mlp32_backward_impl(dY, X, w, fb, B, in_dim, out_dim, num_hidden, activation, bb, dX, d, x_layout, dy_stride,
                               y_sigmoid, y_sigmoid_stride, dsigma, h0, h0_stride, stream);
}

}  // extern "C"

// testing aid: the 90 112 bytes of operand fragments as the last build left them, copied to `dst` (device memory) on `stream`
extern "C" int enerf_debug_nerf_frags_copy(void* dst, enerf_stream_t stream) {
    if (!dst) ENERF_BADARG("debug_nerf_frags_copy: dst is required");
    if (g_nerf_built_gen != workspace_generation() || !g_nerf_built[0]) ENERF_BADARG("debug_nerf_frags_copy: no fragments built");
    const void* frags = workspace(WS_NERF_FRAGS, kNerfWsBytes);
    if (!frags) return ENERF_E_NOMEM;
    if (hipMemcpyAsync(dst, frags, kNerfFragBytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
        ENERF_BADARG("debug_nerf_frags_copy: copy failed");
    return 0;
}

// common.h: the weight-gradient sums of the backward call that follows, left to the optimizer's launch
static const void* g_part_key[12] = {};
static uint64_t g_part_gen = ~0ull;
int enerf::nerf_mlp_partial_job(float* const* dwseg_s, float* const* dwseg_c, uint32_t w0_cols_c, uint32_t out_c,
                                const float* const* small_g, const uint32_t* small_n, uint32_t n_small, uint32_t B,
                                hipStream_t s, PartialSums* job) {
    if (n_small != 5 || !small_g || !small_n || !dwseg_s || !dwseg_c || amp_state().scale || B == 0) return 1;
    if ((w0_cols_c != 31 && w0_cols_c != 32) || out_c == 0 || out_c > 16) return 1;
    // the five gradient matrices must be the optimizer's five small tensors, whole (then every element gets exactly one sum)
    const float* seg[5] = {dwseg_s[0], dwseg_s[3], dwseg_c[0], dwseg_c[1], dwseg_c[3]};
    const uint32_t len[5] = {HID * IN, 16 * HID, HID * w0_cols_c, HID * HID, out_c * HID};
    uint32_t seen = 0;
    for (int k = 0; k < 5; k++) {
        if (!seg[k]) return 1;
        for (uint32_t j = 0; j < 5; j++)
            if (small_g[j] == seg[k] && small_n[j] == len[k]) seen |= 1u << j;
    }
    if (seen != 31u) return 1;
    if (int ew = workspace_family_enter(1, s)) return ew;
    uint32_t* ws = (uint32_t*)workspace(WS_NERF_FRAGS, kNerfWsBytes);
    const uint32_t grid = pgrid(B, g_bwd_blocks ? g_bwd_blocks : 256);
    float* partial = (float*)workspace(WS_NERF_PART, sizeof(float) * (size_t)grid * kNerfPartialStride);
    if (!ws || !partial) return ENERF_E_NOMEM;
    uint32_t* map = ws + (kNerfFragBytes + kNerfMapBytes) / 4;
    const uint32_t nw_c = HID * IN + HID * HID + out_c * HID;
    const void* key[12] = {seg[0], seg[1], seg[2], seg[3], seg[4], small_g[0], small_g[1], small_g[2], small_g[3], small_g[4],
                           (const void*)(uintptr_t)w0_cols_c, (const void*)(uintptr_t)out_c};
    bool same = g_part_gen == workspace_generation();
    for (int k = 0; k < 12; k++) same = same && g_part_key[k] == key[k];
    if (!same) {
        WDst ds, dc;
        for (int k = 0; k < 4; k++) {
            ds.seg[k] = (k == 0 || k == 3) ? dwseg_s[k] : nullptr;
            dc.seg[k] = (k == 0 || k == 1 || k == 3) ? dwseg_c[k] : nullptr;
        }
        ds.w0_cols = IN; ds.nerf_perm = 0; ds.overwrite = 1;
        dc.w0_cols = w0_cols_c; dc.nerf_perm = 1; dc.overwrite = 1;
        SmallDst sm;
        for (int k = 0; k < 5; k++) { sm.g[k] = small_g[k]; sm.n[k] = small_n[k]; }
        hipLaunchKernelGGL(k_nerf_partial_map, dim3(div_up(kNerfSigmaWords + nw_c, 256)), dim3(256), 0, s, ds, dc,
                           kNerfSigmaWords, nw_c, sm, map);
        for (int k = 0; k < 12; k++) g_part_key[k] = key[k];
        g_part_gen = workspace_generation();
    }
    job->partial = partial;
    job->map = map;
    job->parts = grid;
    job->stride = kNerfPartialStride;
    job->n = kNerfSigmaWords + nw_c;
    return 0;
}

// common.h: the fragments' build handed to a launch that runs before the MLP forward on `s`
int enerf::nerf_mlp_frag_job(const float* const* wseg_s, const float* const* wseg_c, uint32_t w0_cols_c, uint32_t out_c,
                             hipStream_t s, SplitJob* job) {
    if (int e = nerf_args_ok(wseg_s, wseg_c, w0_cols_c, out_c, "nerf_mlp_frag_job")) return e;
    if (int ew = workspace_family_enter(1, s)) return ew;
    uint32_t* frags = (uint32_t*)workspace(WS_NERF_FRAGS, kNerfWsBytes);
    if (!frags) return ENERF_E_NOMEM;
    uint32_t* map = frags + kNerfFragBytes / 4;
    if (g_nerf_map_cols != w0_cols_c || g_nerf_map_out != out_c || g_nerf_map_gen != workspace_generation()) {
        nerf_launch_frag_map(w0_cols_c, out_c, map, s);
        g_nerf_map_cols = w0_cols_c;
        g_nerf_map_out = out_c;
        g_nerf_map_gen = workspace_generation();
    }
    const float* w[5] = {wseg_s[0], wseg_s[3], wseg_c[0], wseg_c[1], wseg_c[3]};
    for (int k = 0; k < 5; k++) {
        job->src[k] = w[k];
        g_nerf_built[k] = w[k];
    }
    g_nerf_built_cols = w0_cols_c;
    g_nerf_built_out = out_c;
    g_nerf_built_gen = workspace_generation();
    job->map = map;
    job->out = frags;
    job->threads = kNerfFragBytes / 2048 * 64;
    return 0;
}

void enerf::nerf_mlp_frags_invalidate() {
    for (int k = 0; k < 5; k++) g_nerf_built[k] = nullptr;
    g_nerf_built_gen = 0;
}

namespace enerf_mlp32 {

// The reference's FFMLP entry points on this file's data flow (mlp32_common.h).  The arithmetic mode and the 16-bit I/O
// flag are set for the duration of the call.
int ffmlp16_forward(int dtype, const void* X, const float* W32, uint32_t B, uint32_t num_hidden, uint32_t activation,
                    void* Y, hipStream_t s) {
    if (num_hidden < 2 || num_hidden > 3) ENERF_BADARG("ffmlp16_forward: two or three hidden layers");
    const int prev = g_precision;
    g_precision = dtype == ENERF_BF16 ? 2 : 3;
    g_io16 = true;
    const int rc = mlp32_forward_impl(reinterpret_cast<const float*>(X), blob_src(W32, num_hidden), B, IN, 16, num_hidden,
                                      activation, 6, nullptr, reinterpret_cast<float*>(Y), 0, 16, nullptr, nullptr,
                                      (enerf_stream_t)s);
    g_io16 = false;
    g_precision = prev;
    return rc;
}

int ffmlp16_backward(int dtype, const void* dY, const void* X, const float* W32, uint32_t B, uint32_t num_hidden,
                     uint32_t activation, void* dX, float* dW32, hipStream_t s) {
    if (num_hidden < 2 || num_hidden > 3) ENERF_BADARG("ffmlp16_backward: two or three hidden layers");
    const WSrc w = blob_src(W32, num_hidden);
    WDst d;
    for (int k = 0; k < 4; k++) d.seg[k] = w.seg[k] ? dW32 + (w.seg[k] - W32) : nullptr;
    d.w0_cols = IN;
    d.nerf_perm = 0;
    d.overwrite = 1;
    const int prev = g_precision;
    g_precision = dtype == ENERF_BF16 ? 2 : 3;
    g_io16 = true;
    const int rc = mlp32_backward_impl(reinterpret_cast<const float*>(dY), reinterpret_cast<const float*>(X), w, nullptr, B,
                                       IN, 16, num_hidden, activation, nullptr, reinterpret_cast<float*>(dX), d, 0, 16,
                                       nullptr, 0, nullptr, nullptr, 0, (enerf_stream_t)s);
    g_io16 = false;
    g_precision = prev;
    return rc;
}

}  // namespace enerf_mlp32


#ifdef ENERF_MLP_TIMING
extern "C" int enerf_debug_mlp_phases(unsigned long long* out, int reset) {
    if (reset) {
        unsigned long long z[16] = {0};
        return hipMemcpyToSymbol(HIP_SYMBOL(g_mlp_phase), z, sizeof(z)) == hipSuccess ? 0 : -1;
    }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mlp_phase), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
