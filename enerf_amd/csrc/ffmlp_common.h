// ffmlp_common.h -- types, MFMA wrappers, fragment builders and tile I/O shared by ffmlp.hip (forward + dgrad, MFMA
// accumulators in arch VGPRs) and ffmlp_wgrad.hip (weight gradients, accumulators in AGPRs).  See ffmlp.hip for the
// design notes.
#pragma once
#include "mfma_guard.h"
#include <hip/hip_runtime.h>

#include "common.h"

namespace enerf_ffmlp {
using namespace enerf;

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

// (ENERF_MFMA_GUARD: mfma_guard.h -- a zero-initialised MFMA's result must not land on its own operands)
__device__ __forceinline__ f32x16 mma(bf16x8 a, bf16x8 b, f32x16 c) {
    f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    ENERF_MFMA_GUARD(d, a, b, c);
    return d;
}
__device__ __forceinline__ f32x16 mma(f16x8 a, f16x8 b, f32x16 c) {
    f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    ENERF_MFMA_GUARD(d, a, b, c);
    return d;
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

// Whole-tile activation.  relu / none (the only ones the NeRF nets use) are inline; every other activation goes through
// a NOINLINE function: left inline, hipcc if-converts the uniform switch into straight-line code that evaluates every
// transcendental variant for every element and selects (seen in the ISA: 384 v_exp + 128 v_sin + 768 v_div_scale per
// tile), which made the kernel 3x slower than its MFMA + memory time.
// (by value in / out: a by-reference accumulator would be forced into scratch memory by the call ABI)
static __device__ __attribute__((noinline)) f32x16 apply_act_generic(f32x16 t, uint32_t a) {
    f32x16 r;
#pragma unroll
    for (int q = 0; q < 16; q++) r[q] = act_fwd(t[q], a);
    return r;
}
static __device__ __attribute__((noinline)) f32x16 apply_act_bwd_generic(f32x16 g, f32x16 fw, uint32_t a) {
    f32x16 r;
#pragma unroll
    for (int q = 0; q < 16; q++) r[q] = act_bwd(g[q], fw[q], a);
    return r;
}
__device__ __forceinline__ void apply_act(f32x16& t, uint32_t a) {
    if (a == 0) {
        // relu as a signed-integer max on the bit pattern: one v_max_i32 per element, no fmaxf canonicalisation pass
        // (negative floats, -0.0 and negative NaNs are negative integers -> +0.0; everything else is unchanged)
#pragma unroll
        for (int q = 0; q < 16; q++) t[q] = __int_as_float(max(__float_as_int(t[q]), 0));
    } else if (a != 6) {
        t = apply_act_generic(t, a);
    }
}
__device__ __forceinline__ void apply_act_bwd(f32x16& g, const float (&fw)[16], uint32_t a) {
    if (a == 0) {
#pragma unroll
        for (int q = 0; q < 16; q++) g[q] = fw[q] > 0.0f ? g[q] : 0.0f;
    } else if (a != 6) {
        f32x16 f;
#pragma unroll
        for (int q = 0; q < 16; q++) f[q] = fw[q];
        g = apply_act_bwd_generic(g, f, a);
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

// A freshly converted pair of 16-bit operands against the MFMAs that read it: the barrier of mlp32s_ops.h (operand_ready,
// where the hazard is described) -- every register of both fragments is written, in program order, before any reader, and
// FFMLP_READY_NOPS + 1 idle cycles follow.  A translation unit asks for it with FFMLP_OPERAND_BARRIER: ffnerf.hip does (the
// compiler put conversions two slots in front of their MFMAs there); ffmlp.hip's kernels keep the conversions of a tile
// seven and more slots away from the MFMAs of the next layer as compiled, which tests/test_mfma_operand_overlap.py checks on
// the assembly of every build -- the barrier would cost ffmlp_inference 8 % (0.437 -> 0.402 of the bf16 peak).
#ifndef FFMLP_READY_NOPS
#define FFMLP_READY_NOPS 1
#endif
#define FFMLP_STR2(x) #x
#define FFMLP_STR(x) FFMLP_STR2(x)
template <typename X8>
__device__ __forceinline__ void operands_barrier(X8 (&out)[2]) {
    typedef int i32x4_ __attribute__((ext_vector_type(4)));
    i32x4_ a = __builtin_bit_cast(i32x4_, out[0]), b = __builtin_bit_cast(i32x4_, out[1]);
    asm volatile("s_nop " FFMLP_STR(FFMLP_READY_NOPS) : "+v"(a), "+v"(b));
    out[0] = __builtin_bit_cast(X8, a);
    out[1] = __builtin_bit_cast(X8, b);
}
// (the ReLU path of act_to_frags -- the inference kernels' -- only where the translation unit asks: see above)
template <typename X8>
__device__ __forceinline__ void operands_ready(X8 (&out)[2]) {
#ifdef FFMLP_OPERAND_BARRIER
    operands_barrier(out);
#endif
}

// D tile (fp32, lane = sample) -> two permuted-order K-blocks of 16-bit operands, with an elementwise map
template <typename E>
__device__ __forceinline__ void tile_to_frags(const f32x16& acc, typename V<E>::x8 (&out)[2]) {
#pragma unroll
    for (int kbb = 0; kbb < 2; kbb++)
#pragma unroll
        for (int e = 0; e < 8; e++) out[kbb][e] = (E)acc[8 * kbb + e];
    operands_barrier(out);                       // (plain conversions: the compiler does place them right in front of the MFMAs)
}

// activation + conversion of a hidden layer's D tile.  ReLU commutes with the (sign-symmetric, monotone) rounding to
// 16 bits, so it is applied to the packed result: a signed 16-bit max with 0 on the bit patterns (negative values,
// -0.0 and negative NaNs are negative integers -> +0.0), one v_pk_max_i16 per two elements instead of one v_max per
// element before the conversion.
// ACT: 0 = ReLU, 1 = none, 2 = any other activation (runtime code `a`, out-of-line).  The hidden activation is a
// kernel template parameter: a runtime switch per tile puts a control-flow diamond (and its register copies) around
// every one of the dozen tiles a wavefront processes per iteration.
template <typename E, int ACT>
__device__ __forceinline__ void act_to_frags(f32x16& acc, uint32_t a, typename V<E>::x8 (&out)[2]) {
    if (ACT == 1) {
        tile_to_frags<E>(acc, out);
    } else if (ACT == 0) {
        // convert pairs (one v_cvt_pk per two elements), clamp the packed pair, assemble the 16-byte fragment
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef E e16x2 __attribute__((ext_vector_type(2)));
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int kbb = 0; kbb < 2; kbb++) {
            i32x4 r;
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const f32x2 f = {acc[8 * kbb + 2 * p], acc[8 * kbb + 2 * p + 1]};
                s16x2 v = __builtin_bit_cast(s16x2, __builtin_convertvector(f, e16x2));
                v = __builtin_elementwise_max(v, (s16x2)(0));
                r[p] = __builtin_bit_cast(int, v);
            }
            out[kbb] = __builtin_bit_cast(typename V<E>::x8, r);
        }
        operands_ready(out);
    } else {
        acc = apply_act_generic(acc, a);
        tile_to_frags<E>(acc, out);
    }
}
template <int ACT>
__device__ __forceinline__ void act_bwd_t(f32x16& g, const float (&fw)[16], uint32_t a) {
    if (ACT == 0) {
#pragma unroll
        for (int q = 0; q < 16; q++) g[q] = fw[q] > 0.0f ? g[q] : 0.0f;
    } else if (ACT == 2) {
        f32x16 f;
#pragma unroll
        for (int q = 0; q < 16; q++) f[q] = fw[q];
        g = apply_act_bwd_generic(g, f, a);
    }
}
__host__ __device__ inline int act_class(uint32_t a) { return a == 0 ? 0 : (a == 6 ? 1 : 2); }

// store / load one [32 samples][32 neurons] D-tile-shaped block of a row-major [B,64] 16-bit buffer.
// Lane (j, h) owns neurons 32*ib + 8*g + 4*h + r (g = 0..3, r = 0..3): four 8-byte pieces of row j, 16 bytes apart.
// The two lanes of a sample first trade halves (v_permlane32_swap: the upper 32 lanes of one register <-> the lower 32
// lanes of another), after which the lower lane holds neurons 0..15 and the upper lane 16..31 of the block: two 16-byte
// accesses per lane instead of four 8-byte ones (the forward / dgrad kernels are bound by exactly these accesses).
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void trade_halves(u32x4_t& lo, u32x4_t& hi) {
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const auto r = __builtin_amdgcn_permlane32_swap(lo[d], hi[d], false, false);
        lo[d] = r[0];
        hi[d] = r[1];
    }
}
template <typename E>
__device__ __forceinline__ void store_tile(E* rowptr /* row of sample j */, int ib, int h,
                                           const typename V<E>::x8 (&fr)[2]) {
    u32x4_t a = __builtin_bit_cast(u32x4_t, fr[0]), b = __builtin_bit_cast(u32x4_t, fr[1]);
    trade_halves(a, b);       // lower lane: a = own groups 0/1, b = partner's; upper lane: a = partner's groups 2/3, b = own
    E* p = rowptr + 32 * ib + 16 * h;
    *reinterpret_cast<u32x4_t*>(p) = u32x4_t{a[0], a[1], b[0], b[1]};
    *reinterpret_cast<u32x4_t*>(p + 8) = u32x4_t{a[2], a[3], b[2], b[3]};
}
template <typename E>
__device__ __forceinline__ void load_tile_f32(const E* rowptr, int ib, int h, float (&v)[16]) {
    const E* p = rowptr + 32 * ib + 16 * h;
    const u32x4_t p0 = *reinterpret_cast<const u32x4_t*>(p), p1 = *reinterpret_cast<const u32x4_t*>(p + 8);
    u32x4_t a = {p0[0], p0[1], p1[0], p1[1]}, b = {p0[2], p0[3], p1[2], p1[3]};
    trade_halves(a, b);       // back to the D-tile order: a = K-block 0 of this lane, b = K-block 1
    const typename V<E>::x8 f0 = __builtin_bit_cast(typename V<E>::x8, a), f1 = __builtin_bit_cast(typename V<E>::x8, b);
#pragma unroll
    for (int e = 0; e < 8; e++) {
        v[e] = (float)f0[e];
        v[8 + e] = (float)f1[e];
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
// The same with the element count known at compile time (256 threads): every load of the thread is issued before its
// first LDS store, so the copy costs one memory latency instead of one per pass (9 passes for the largest net).
template <uint32_t N, typename E>
__device__ __forceinline__ void stage_weights_n(E* wl, const E* __restrict__ w) {
    constexpr uint32_t V = N / 8, PASSES = (V + 255u) / 256u;
    const uint4* src = reinterpret_cast<const uint4*>(w);
    uint4* dst = reinterpret_cast<uint4*>(wl);
    uint4 v[PASSES];
#pragma unroll
    for (uint32_t k = 0; k < PASSES; k++) {
        const uint32_t i = threadIdx.x + k * 256u;
        v[k] = src[i < V ? i : V - 1u];
    }
#pragma unroll
    for (uint32_t k = 0; k < PASSES; k++) {
        const uint32_t i = threadIdx.x + k * 256u;
        if (i < V) dst[i] = v[k];
    }
    __syncthreads();
}

// The forward's copy: rows padded by 8 elements (16 B).  Its fragment builders read ROWS across lanes (lane = output
// neuron): with the blob's own 128-byte rows every lane of a half-wave hit the same LDS bank -- a 32-way conflict on each of
// the ~50 reads of the set-up, ~10 % of k_ffmlp_fwd's time at 2 M samples (round 5); 144-byte rows spread them over the banks.
// Matrix m of the blob (W0 [64, IN] | Wh [64, 64] x (NL-1) | Wout [16, 64]) starts at padded_base<IN>(m).
constexpr uint32_t kRowPad = 8;
template <int IN>
__host__ __device__ constexpr uint32_t padded_base(int m) {      // m = 0: W0, 1..: hidden matrices, then Wout
    return m == 0 ? 0u : HID * (IN + kRowPad) + (uint32_t)(m - 1) * HID * (HID + kRowPad);
}
template <int IN, int NL>
__host__ __device__ constexpr uint32_t padded_size() { return padded_base<IN>(NL) + OUT * (HID + kRowPad); }
template <int IN, int NL, typename E>
__device__ __forceinline__ void stage_weights_padded(E* wl, const E* __restrict__ w) {
    constexpr uint32_t N = HID * (IN + HID * (NL - 1) + OUT);
    constexpr uint32_t V = N / 8, PASSES = (V + 255u) / 256u;
    constexpr uint32_t C0 = IN / 8, V0 = HID * C0;                 // 16-byte chunks per row / in all of W0
    const uint4* src = reinterpret_cast<const uint4*>(w);
    uint4 v[PASSES];
#pragma unroll
    for (uint32_t k = 0; k < PASSES; k++) {
        const uint32_t i = threadIdx.x + k * 256u;
        v[k] = src[i < V ? i : V - 1u];
    }
#pragma unroll
    for (uint32_t k = 0; k < PASSES; k++) {
        const uint32_t i = threadIdx.x + k * 256u;
        if (i < V) {
            uint32_t at;                                           // element offset in the padded copy
            if (i < V0) at = (i / C0) * (IN + kRowPad) + (i % C0) * 8u;
            else {
                const uint32_t r = (i - V0) / (HID / 8), c = (i - V0) % (HID / 8);      // rows of the 64-wide matrices, numbered through
                at = HID * (IN + kRowPad) + r * (HID + kRowPad) + c * 8u;
            }
            *reinterpret_cast<uint4*>(wl + at) = v[k];
        }
    }
    __syncthreads();
}

#define FFMLP_DISPATCH_ACT(E, CALL)                                                 \
    switch (act_class(act)) {                                                       \
        case 0: { constexpr int ACT = 0; FFMLP_DISPATCH(E, CALL); } break;          \
        case 1: { constexpr int ACT = 1; FFMLP_DISPATCH(E, CALL); } break;          \
        default: { constexpr int ACT = 2; FFMLP_DISPATCH(E, CALL); } break;         \
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


// weight-gradient pass (ffmlp_wgrad.hip): dW from dY / X / forward_buffer / backward_buffer, accumulated into the
// caller's zero-filled 16-bit grad_weights
int ffmlp_wgrad_launch(int dtype, const void* dY, const void* X, const void* fb, const void* bb, uint32_t B,
                       uint32_t input_dim, uint32_t num_layers, void* grad_weights, hipStream_t s);

}  // namespace enerf_ffmlp
