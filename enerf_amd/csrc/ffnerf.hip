// ffnerf.hip -- the whole fully-fused NeRF network of nerf/network_ff.py as ONE inference kernel on the CDNA4 matrix cores:
//
//     hash-grid features -> sigma_net (FFMLP 32 -> 64 -> 64 -> 16) -> sigma = exp(h[0]), geo_feat = h[1:16]
//     dirs -> SH(4) ; [SH 16 | geo_feat 15 | 0] -> color_net (FFMLP 32 -> 64 -> 64 -> 64 -> 16) -> rgb = sigmoid(h[0:3])
//     (network_ff.py:60-88)
//
// The op-by-op route spends 5 % of a 640x480 frame in the two FFMLP kernels and 40 % in what surrounds them: fp32 -> 16
// bit conversions, the 128-row padding cats, the [SH | geo | pad] cat, exp / sigmoid / slicing launches and a row-major
// encoder output.  Here a wavefront takes 32 samples through both nets in registers (v_mfma_f32_32x32x16_{bf16,f16}):
//  * the first layer's B operand is read straight from the grid encoder's level-major fp32 output [16, Mp, 2] (four
//    coalesced 8-byte loads per 16 input features) and converted in registers;
//  * as in ffmlp.hip everything is computed transposed, so a layer's D tile is the next layer's B operand once the
//    weight fragments use the matching permuted K order -- that includes the hand-over between the nets: rows 0..15 of
//    the sigma net's output tile ARE K-block 1 of the colour net's first layer, with the colour weights' columns
//    permuted to [none | geo 0..14] in D-tile order; K-block 0 is the SH basis evaluated in registers;
//  * weights are read as fp32 parameters and converted while they are staged into LDS (no per-call conversion launch).
// Roundings follow the op-by-op route (16-bit net outputs, exp / sigmoid evaluated in fp32 on the rounded value and
// rounded again), fp32 accumulation throughout.
#define FFMLP_OPERAND_BARRIER          // (see ffmlp_common.h: operands_ready)
#include "ffmlp_common.h"
#include "sh_basis.h"

using namespace enerf_ffmlp;

namespace {

constexpr uint32_t kSigmaW = HID * (32 + HID + OUT);          // 7168:  W0 64x32 | Wh 64x64 | Wout 16x64
constexpr uint32_t kColorW = HID * (32 + 2 * HID + OUT);      // 11264: W0 64x32 | Wh 2 x 64x64 | Wout 16x64

// ---- weight fragments in LDS, fragment-major: fragment f = 64 lanes x 8 elements, contiguous (a conflict-free
// ds_read_b128 per lane).  Order: sigma W0 [ob][kb] 0..3 | sigma Wh [ob][blk] 4..11 | sigma Wout [blk] 12..15 |
// colour W0 [ob][kb] 16..19 | colour Wh [l][ob][blk] 20..35 | colour Wout [blk] 36..39.
constexpr int kFrags = 40;
constexpr int F_SW0 = 0, F_SWH = 4, F_SWO = 12, F_CW0 = 16, F_CWH = 20, F_CWO = 36;

// sigma-net output row held by D-tile slot (lane half h, element e) of K-block 0
__device__ __forceinline__ int drow(int h, int e) { return 4 * h + (e & 3) + 8 * (e >> 2); }
// natural / permuted K order of ffmlp_common.h: column of element e of K-block kb (natural) or blk (permuted)
__device__ __forceinline__ int col_nat(int kb, int h, int e) { return 16 * kb + 8 * h + e; }
__device__ __forceinline__ int col_perm(int blk, int h, int e) { return 16 * blk + drow(h, e); }

// element e of fragment f for `lane`, read from the fp32 parameter blobs
__device__ __forceinline__ float frag_source(const float* __restrict__ Ws, const float* __restrict__ Wc, int f, int lane,
                                             int e) {
    const int j = lane & 31, h = lane >> 5;
    if (f < F_SWH) {                                     // sigma W0 [64 x 32], natural order
        const int ob = (f - F_SW0) >> 1, kb = (f - F_SW0) & 1;
        return Ws[(32 * ob + j) * 32 + col_nat(kb, h, e)];
    }
    if (f < F_SWO) {                                     // sigma Wh [64 x 64], permuted order
        const int ob = (f - F_SWH) >> 2, blk = (f - F_SWH) & 3;
        return Ws[HID * 32 + (32 * ob + j) * HID + col_perm(blk, h, e)];
    }
    if (f < F_CW0)                                       // sigma Wout [16 x 64]: rows 16..31 of the tile are zero
        return j < OUT ? Ws[HID * 32 + HID * HID + j * HID + col_perm(f - F_SWO, h, e)] : 0.0f;
    if (f < F_CWH) {                                     // colour W0 [64 x 32]: K-block 0 = SH (natural order),
        const int ob = (f - F_CW0) >> 1, kb = (f - F_CW0) & 1;       // K-block 1 = [none | geo 0..14] in D-tile order
        if (kb == 0) return Wc[(32 * ob + j) * 32 + col_nat(0, h, e)];
        const int r = drow(h, e);
        return r == 0 ? 0.0f : Wc[(32 * ob + j) * 32 + 16 + r - 1];
    }
    if (f < F_CWO) {                                     // colour Wh [2][64 x 64]
        const int l = (f - F_CWH) >> 3, ob = ((f - F_CWH) >> 2) & 1, blk = (f - F_CWH) & 3;
        return Wc[HID * 32 + l * HID * HID + (32 * ob + j) * HID + col_perm(blk, h, e)];
    }
    return j < OUT ? Wc[HID * 32 + 2 * HID * HID + j * HID + col_perm(f - F_CWO, h, e)] : 0.0f;
}

// the same as an address: -> pointer to read (always readable) and whether the element is a structural zero.  Branch-free
// in the lane, so that the staging loop below can issue every load of a thread before its first LDS store.
__device__ __forceinline__ const float* frag_address(const float* __restrict__ Ws, const float* __restrict__ Wc, int f, int lane,
                                                     int e, bool& zero) {
    const int j = lane & 31, h = lane >> 5;
    const int jo = j < OUT ? j : 0;
    zero = false;
    if (f < F_SWH) return Ws + (32 * ((f - F_SW0) >> 1) + j) * 32 + col_nat((f - F_SW0) & 1, h, e);
    if (f < F_SWO) return Ws + HID * 32 + (32 * ((f - F_SWH) >> 2) + j) * HID + col_perm((f - F_SWH) & 3, h, e);
    if (f < F_CW0) {
        zero = j >= OUT;
        return Ws + HID * 32 + HID * HID + jo * HID + col_perm(f - F_SWO, h, e);
    }
    if (f < F_CWH) {
        const int ob = (f - F_CW0) >> 1, kb = (f - F_CW0) & 1;
        const int r = drow(h, e);
        zero = kb == 1 && r == 0;
        return Wc + (32 * ob + j) * 32 + (kb == 0 ? col_nat(0, h, e) : (r == 0 ? 0 : 16 + r - 1));
    }
    if (f < F_CWO) {
        const int l = (f - F_CWH) >> 3, ob = ((f - F_CWH) >> 2) & 1, blk = (f - F_CWH) & 3;
        return Wc + HID * 32 + l * HID * HID + (32 * ob + j) * HID + col_perm(blk, h, e);
    }
    zero = j >= OUT;
    return Wc + HID * 32 + 2 * HID * HID + jo * HID + col_perm(f - F_CWO, h, e);
}

struct TileIn {
    float2 f[2][4];
    float dx, dy, dz;
};
__device__ __forceinline__ TileIn load_inputs(const float2* __restrict__ f2, const float* __restrict__ dirs, uint32_t tile,
                                              int j, int h, uint32_t M, uint32_t Mp) {
    TileIn t;
    const size_t s = (size_t)tile * 32 + j;
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int q = 0; q < 4; q++) t.f[kb][q] = f2[(size_t)(8 * kb + 4 * h + q) * Mp + s];
    const size_t sd = s < M ? s : (size_t)M - 1;
    t.dx = dirs[sd * 3]; t.dy = dirs[sd * 3 + 1]; t.dz = dirs[sd * 3 + 2];
    return t;
}

template <typename E>
__global__ void __launch_bounds__(256, 2) k_ffnerf_infer(const float* __restrict__ feats, const float* __restrict__ dirs,
                                                         const float* __restrict__ Ws, const float* __restrict__ Wc,
                                                         uint32_t M, ShNorm4 nrm, float* __restrict__ sigma,
                                                         float* __restrict__ rgb) {
    using x8 = typename V<E>::x8;
    __shared__ __attribute__((aligned(16))) E wl[kFrags * 64 * 8];
    {
        // a thread builds whole 16-byte fragment pieces (fragment f, lane): 10 of them, 80 gathered floats, every load
        // issued before the first store -- as the element-by-element loop it was one L2 round trip per element, 80 in a row
        // (rocprofv3, round 5: ~10 % of the kernel's 0.5 ms per frame were this set-up); f is uniform in a wavefront
        constexpr int PIECES = kFrags * 64 / 256;
        static_assert(kFrags * 64 % 256 == 0, "pieces per thread");
        float v[PIECES][8];
        bool z[PIECES][8];
#pragma unroll
        for (int k = 0; k < PIECES; k++) {
            const int piece = (int)threadIdx.x + 256 * k;
#pragma unroll
            for (int e = 0; e < 8; e++) v[k][e] = *frag_address(Ws, Wc, piece >> 6, piece & 63, e, z[k][e]);
        }
#pragma unroll
        for (int k = 0; k < PIECES; k++) {
            const int piece = (int)threadIdx.x + 256 * k;
            x8 q;
#pragma unroll
            for (int e = 0; e < 8; e++) q[e] = (E)(z[k][e] ? 0.0f : v[k][e]);
            reinterpret_cast<x8*>(wl)[piece] = q;
        }
    }
    __syncthreads();
    const int lane = lane_id(), j = lane & 31, h = lane >> 5;
    const x8* fr = reinterpret_cast<const x8*>(wl) + lane;             // fragment f of this lane: fr[f * 64]

    // resident fragments: first and last layers (64 VGPRs); the hidden layers' 24 fragments are re-read from LDS per
    // tile (24 KB per tile and wave, conflict-free) -- held in registers as well the kernel needs 280 of them
    x8 s_w0[2][2], s_wo[4], c_w0[2][2], c_wo[4];
#pragma unroll
    for (int ob = 0; ob < 2; ob++)
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
            s_w0[ob][kb] = fr[(F_SW0 + 2 * ob + kb) * 64];
            c_w0[ob][kb] = fr[(F_CW0 + 2 * ob + kb) * 64];
        }
#pragma unroll
    for (int blk = 0; blk < 4; blk++) {
        s_wo[blk] = fr[(F_SWO + blk) * 64];
        c_wo[blk] = fr[(F_CWO + blk) * 64];
    }

    const uint32_t Mp = (M + 31u) & ~31u;
    const uint32_t ntiles = Mp / 32;
    // T tiles (64 samples) per iteration: two independent MFMA dependency chains per layer, and every hidden-layer
    // fragment read from LDS feeds T MFMAs
    constexpr int T = 2;
    const uint32_t ngroups = div_up(ntiles, (uint32_t)T);
    const uint32_t gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t nw = gridDim.x * (blockDim.x >> 6);
    const float2* f2 = reinterpret_cast<const float2*>(feats);

    // inputs of a tile: features of levels 8*kb + 4*h .. +3 (pad rows of the level-major tensor are zero) + direction;
    // a group's second tile may lie past the end: it is computed on the last tile's data and not stored
    TileIn in[T];
#pragma unroll
    for (int t = 0; t < T; t++) in[t] = TileIn{};
    if (gw < ngroups) {
#pragma unroll
        for (int t = 0; t < T; t++) in[t] = load_inputs(f2, dirs, min(gw * T + t, ntiles - 1), j, h, M, Mp);
    }

    for (uint32_t grp = gw; grp < ngroups; grp += nw) {
        size_t s[T];
        bool live[T];
#pragma unroll
        for (int t = 0; t < T; t++) {
            s[t] = (size_t)(grp * T + t) * 32 + j;
            live[t] = s[t] < M && h == 0;
        }
        // an opaque zero keeps the hidden-layer fragment reads inside the loop (they are loop invariant otherwise)
        uint32_t opaque;
        asm volatile("s_mov_b32 %0, 0" : "=s"(opaque));
        const x8* frh = fr + opaque;
        x8 xb[T][2];
        float dx[T], dy[T], dz[T];
#pragma unroll
        for (int t = 0; t < T; t++) {
#pragma unroll
            for (int kb = 0; kb < 2; kb++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    xb[t][kb][2 * q] = (E)in[t].f[kb][q].x;
                    xb[t][kb][2 * q + 1] = (E)in[t].f[kb][q].y;
                }
            operands_ready(xb[t]);                  // (the operand barrier of ffmlp_common.h / mlp32s_ops.h)
            dx[t] = in[t].dx; dy[t] = in[t].dy; dz[t] = in[t].dz;
        }
        if (grp + nw < ngroups) {                                  // next group's loads fly during this one's MFMAs
#pragma unroll
            for (int t = 0; t < T; t++)
                in[t] = load_inputs(f2, dirs, min((grp + nw) * T + t, ntiles - 1), j, h, M, Mp);
        }

        // ---- sigma net
        f32x16 acc[T][2];
        x8 hb[T][2][2];
#pragma unroll
        for (int ob = 0; ob < 2; ob++)
#pragma unroll
            for (int t = 0; t < T; t++) {
                acc[t][ob] = (f32x16)(0.0f);
#pragma unroll
                for (int kb = 0; kb < 2; kb++) acc[t][ob] = mma(s_w0[ob][kb], xb[t][kb], acc[t][ob]);
            }
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int ob = 0; ob < 2; ob++) act_to_frags<E, 0>(acc[t][ob], 0, hb[t][ob]);
#pragma unroll
        for (int ob = 0; ob < 2; ob++) {
#pragma unroll
            for (int t = 0; t < T; t++) acc[t][ob] = (f32x16)(0.0f);
#pragma unroll
            for (int blk = 0; blk < 4; blk++) {
                const x8 w = frh[(F_SWH + 4 * ob + blk) * 64];
#pragma unroll
                for (int t = 0; t < T; t++) acc[t][ob] = mma(w, hb[t][blk >> 1][blk & 1], acc[t][ob]);
            }
        }
#pragma unroll
        for (int t = 0; t < T; t++) {
            x8 nb[2][2];
#pragma unroll
            for (int ob = 0; ob < 2; ob++) act_to_frags<E, 0>(acc[t][ob], 0, nb[ob]);
#pragma unroll
            for (int ob = 0; ob < 2; ob++)
#pragma unroll
                for (int kbb = 0; kbb < 2; kbb++) hb[t][ob][kbb] = nb[ob][kbb];
        }
        x8 geo[T][2];
#pragma unroll
        for (int t = 0; t < T; t++) {
            f32x16 so = (f32x16)(0.0f);
#pragma unroll
            for (int blk = 0; blk < 4; blk++) so = mma(s_wo[blk], hb[t][blk >> 1][blk & 1], so);
            tile_to_frags<E>(so, geo[t]);           // geo[0]: rows 0..15 = [sigma_raw | geo_feat] in D-tile order
            if (live[t]) sigma[s[t]] = (float)(E)expf((float)geo[t][0][0]);         // trunc_exp forward
        }

        // ---- colour net
#pragma unroll
        for (int t = 0; t < T; t++) {
            float Y[16];
            sh4(dx[t], dy[t], dz[t], nrm, Y);
            x8 xsh;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                // lower lane half takes Y[e], upper Y[8 + e]: a bit select (left as `h ? :` the compiler spills Y to
                // scratch and indexes it by lane)
                const uint32_t m = 0u - (uint32_t)h;
                xsh[e] = (E)__uint_as_float((__float_as_uint(Y[e]) & ~m) | (__float_as_uint(Y[8 + e]) & m));
            }
            {
                x8 pair[2] = {xsh, xsh};
                operands_ready(pair);
                xsh = pair[0];
            }
#pragma unroll
            for (int ob = 0; ob < 2; ob++) {
                acc[t][ob] = mma(c_w0[ob][0], xsh, (f32x16)(0.0f));
                acc[t][ob] = mma(c_w0[ob][1], geo[t][0], acc[t][ob]);
            }
        }
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int ob = 0; ob < 2; ob++) act_to_frags<E, 0>(acc[t][ob], 0, hb[t][ob]);
#pragma unroll
        for (int l = 0; l < 2; l++) {
#pragma unroll
            for (int ob = 0; ob < 2; ob++) {
#pragma unroll
                for (int t = 0; t < T; t++) acc[t][ob] = (f32x16)(0.0f);
#pragma unroll
                for (int blk = 0; blk < 4; blk++) {
                    const x8 w = frh[(F_CWH + 8 * l + 4 * ob + blk) * 64];
#pragma unroll
                    for (int t = 0; t < T; t++) acc[t][ob] = mma(w, hb[t][blk >> 1][blk & 1], acc[t][ob]);
                }
            }
#pragma unroll
            for (int t = 0; t < T; t++) {
                x8 nb[2][2];
#pragma unroll
                for (int ob = 0; ob < 2; ob++) act_to_frags<E, 0>(acc[t][ob], 0, nb[ob]);
#pragma unroll
                for (int ob = 0; ob < 2; ob++)
#pragma unroll
                    for (int kbb = 0; kbb < 2; kbb++) hb[t][ob][kbb] = nb[ob][kbb];
            }
        }
#pragma unroll
        for (int t = 0; t < T; t++) {
            f32x16 co = (f32x16)(0.0f);
#pragma unroll
            for (int blk = 0; blk < 4; blk++) co = mma(c_wo[blk], hb[t][blk >> 1][blk & 1], co);
            if (live[t]) {
                // rows 0, 1, 2 of the output tile sit in elements 0, 1, 2 of the lower lane half
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float o = (float)(E)co[c];
                    rgb[s[t] * 3 + c] = (float)(E)(1.0f / (1.0f + expf(-o)));
                }
            }
        }
    }
}

}  // namespace

extern "C" {

int enerf_ffnerf_inference(const float* feats, const float* dirs, const float* w_sigma, const float* w_color, uint32_t M,
                           int dtype, float* sigma, float* rgb, enerf_stream_t stream) {
    if (M == 0) return 0;
    if (dtype != ENERF_BF16 && dtype != ENERF_F16) ENERF_BADARG("ffnerf_inference: dtype must be bf16 or f16");
    if ((((uintptr_t)w_sigma | (uintptr_t)w_color | (uintptr_t)feats) & 15) != 0)
        ENERF_BADARG("ffnerf_inference: feats / weights must be 16-byte aligned");
    const ShNorm4 nrm = make_sh_norm4();
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(ENERF_K_FFMLP_FWD, s);
    const uint32_t tiles = div_up(M, 32);
    uint32_t blocks = div_up(div_up(tiles, 2u), 4);
    if (blocks > 512u) blocks = 512u;
    if (dtype == ENERF_BF16)
        k_ffnerf_infer<__bf16><<<blocks, 256, 0, s>>>(feats, dirs, w_sigma, w_color, M, nrm, sigma, rgb);
    else
        k_ffnerf_infer<_Float16><<<blocks, 256, 0, s>>>(feats, dirs, w_sigma, w_color, M, nrm, sigma, rgb);
    ENERF_LAUNCH_CHECK("ffnerf_inference");
    return 0;
}

}  // extern "C"
