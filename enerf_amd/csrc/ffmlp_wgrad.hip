// ffmlp_wgrad.hip -- weight gradients of the fully fused MLP (see ffmlp.hip for the overall design).
// Separate translation unit: these kernels keep up to 192 accumulator registers per lane and want them in AGPRs,
// while ffmlp.hip is built with -amdgpu-mfma-vgpr-form.
#include "ffmlp_common.h"

using namespace enerf_ffmlp;

namespace {

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
__global__ void __launch_bounds__(1024) k_ffmlp_reduce_w(const float* __restrict__ partial, uint32_t nblocks, uint32_t NW,
                                                         E* __restrict__ gw) {
    // 64 weights per workgroup; each of the 16 waves sums every 16th partial block with four independent chains (a
    // single chain of 64 dependent loads per thread took 20 us), then the waves' sums are combined in a fixed order:
    // deterministic
    __shared__ float acc[16][64];
    const uint32_t i = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t part = threadIdx.x >> 6;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (i < NW) {
        uint32_t b = part;
        for (; b + 48 < nblocks; b += 64) {
            s0 += partial[(size_t)b * NW + i];
            s1 += partial[(size_t)(b + 16) * NW + i];
            s2 += partial[(size_t)(b + 32) * NW + i];
            s3 += partial[(size_t)(b + 48) * NW + i];
        }
        for (; b < nblocks; b += 16) s0 += partial[(size_t)b * NW + i];
    }
    acc[part][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    // grad_weights arrives zero-filled; accumulate like the reference's beta = 0/1 GEMMs
    if (part == 0 && i < NW) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < 16; w++) t += acc[w][threadIdx.x];
        gw[i] = (E)((float)gw[i] + t);
    }
}


}  // namespace

namespace enerf_ffmlp {

int ffmlp_wgrad_launch(int dtype, const void* dY, const void* X, const void* fb, const void* bb, uint32_t B,
                       uint32_t input_dim, uint32_t num_layers, void* grad_weights, hipStream_t s) {
    const uint32_t NWn = HID * (input_dim + HID * (num_layers - 1) + OUT);
    uint32_t wgrid = div_up(B / 32, 4);
    if (wgrid > 256u) wgrid = 256u;
    if (int ew = enerf::workspace_family_enter(1, s)) return ew;
    float* partial = (float*)workspace(WS_FFMLP, sizeof(float) * (size_t)wgrid * NWn);
    if (!partial) return ENERF_E_NOMEM;
    if (dtype == ENERF_BF16) {
        using E = __bf16;
        FFMLP_DISPATCH(E, (k_ffmlp_bwd_w<E, KB, NL><<<wgrid, 256, 0, s>>>((const E*)dY, (const E*)X, (const E*)fb,
                                                                           (const E*)bb, partial, B)));
        k_ffmlp_reduce_w<E><<<div_up(NWn, 64), 1024, 0, s>>>(partial, wgrid, NWn, (E*)grad_weights);
    } else {
        using E = _Float16;
        FFMLP_DISPATCH(E, (k_ffmlp_bwd_w<E, KB, NL><<<wgrid, 256, 0, s>>>((const E*)dY, (const E*)X, (const E*)fb,
                                                                           (const E*)bb, partial, B)));
        k_ffmlp_reduce_w<E><<<div_up(NWn, 64), 1024, 0, s>>>(partial, wgrid, NWn, (E*)grad_weights);
    }
    return 0;
}

}  // namespace enerf_ffmlp
