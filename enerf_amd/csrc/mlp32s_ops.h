// mlp32s_ops.h -- operand types and register-level building blocks of the split-bf16 / 16-bit MFMA kernels: hi / lo
// splitting, the three-product multiply, tile flips by the matrix pipe.  Shared by mlp32s.hip (one net per launch) and
// nerf_mlp.hip (sigma + colour net of nerf/network.py in one launch); both are compiled with -amdgpu-mfma-vgpr-form.
// The comments on the arithmetic are at the top of mlp32s.hip.
#pragma once
#include "mfma_guard.h"
#include "mlp32_common.h"

namespace enerf_mlp32 {
#ifdef ENERF_MLP32S_F16
typedef _Float16 elem16;
#define MLP32S_MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16
#define MLP32S_LAUNCH_FWD mlp32s_f16_launch_fwd
#define MLP32S_LAUNCH_BWD mlp32s_f16_launch_bwd
// (sigma = trunc_exp(h) is evaluated in fp32 under the reference's autocast: activation.py's cast_inputs=torch.float)
constexpr bool kRoundExp = false;
#else
typedef __bf16 elem16;
#define MLP32S_MFMA __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define MLP32S_LAUNCH_FWD mlp32s_launch_fwd
#define MLP32S_LAUNCH_BWD mlp32s_launch_bwd
constexpr bool kRoundExp = true;
#endif
typedef elem16 bf16x8 __attribute__((ext_vector_type(8)));
typedef elem16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// P = 3: hi + lo (fp32 operands, three products); P = 1: hi only -- the operand IS a bf16 number (the FFMLP nets of
// nerf/network_ff.py: 16-bit weights, activations rounded to 16 bits between layers, fp32 accumulation), one product.
template <int P>
struct FragT {
    bf16x8 hi, lo;
};
template <>
struct FragT<1> {
    bf16x8 hi;
};

__device__ __forceinline__ f32x16 mmab(bf16x8 a, bf16x8 b, f32x16 c) {
    f32x16 d = MLP32S_MFMA(a, b, c, 0, 0, 0);
    ENERF_MFMA_GUARD(d, a, b, c);
    return d;
}
__device__ __forceinline__ f32x16 mmap(const FragT<3>& a, const FragT<3>& b, f32x16 c) {
    c = mmab(a.lo, b.hi, c);
    c = mmab(a.hi, b.lo, c);
    return mmab(a.hi, b.hi, c);
}
__device__ __forceinline__ f32x16 mmap(const FragT<1>& a, const FragT<1>& b, f32x16 c) { return mmab(a.hi, b.hi, c); }

__device__ __forceinline__ float bf16r(float x) { return (float)(elem16)x; }      // round to nearest 16-bit operand value, back to fp32

// The conversions that make a 16-bit MFMA operand (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 and the v_pk_add_f32 of the low
// halves) against the MFMA that reads it: on gfx950 the matrix pipe can read the operand registers BEFORE a conversion
// issued a few instructions earlier has written lanes 16..31 / 48..63 of them -- the compiler separates the two by the two
// wait states of an ordinary VALU result, which holds while a SIMD runs one or two of these wavefronts and fails with
// three (the wavefronts' conversions queue up in a unit they share): samples 16..31 of a 32-sample tile then saw a
// stale operand, 10^-3..10^-2 in the output, in 5 % of the launches of the fused forward -- or in every launch, depending on
// how the scheduler happened to place the conversions (profiles/r05_mfma_operand_hazard.txt).  The operand therefore passes
// through this barrier: all of its registers are written, in program order, before anything that reads them, and four
// idle cycles follow.  (Found with tools/nerf_fwd_residency.py: padded builds fail in 300 of 300 launches without it, 0
// of 300 with it; the empty barrier alone already clears them, the wait states are the margin.)
#ifndef MLP32S_READY_NOPS
#define MLP32S_READY_NOPS 3
#endif
#define MLP32S_STR2(x) #x
#define MLP32S_STR(x) MLP32S_STR2(x)
#ifdef MLP32S_NO_OPERAND_BARRIER             // (the reproducer's build: tools/nerf_fwd_residency.sh)
__device__ __forceinline__ void operand_ready(i32x4&) {}
__device__ __forceinline__ void operand_ready(i32x4&, i32x4&) {}
#else
__device__ __forceinline__ void operand_ready(i32x4& r) { asm volatile("s_nop " MLP32S_STR(MLP32S_READY_NOPS) : "+v"(r)); }
__device__ __forceinline__ void operand_ready(i32x4& r, i32x4& q) {
    asm volatile("s_nop " MLP32S_STR(MLP32S_READY_NOPS) : "+v"(r), "+v"(q));
}
#endif

template <int P>
__device__ __forceinline__ FragT<P> split8(const float (&v)[8]) {
    i32x4 rh, rl;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const f32x2 f = {v[2 * p], v[2 * p + 1]};
        const bf16x2 h2 = __builtin_convertvector(f, bf16x2);              // v_cvt_pk_bf16_f32 (RNE)
        rh[p] = __builtin_bit_cast(int, h2);
        if (P == 3) {
            const f32x2 rest = f - __builtin_convertvector(h2, f32x2);     // exact in fp32
            rl[p] = __builtin_bit_cast(int, __builtin_convertvector(rest, bf16x2));
        }
    }
#ifdef MLP32S_READY_SPLIT
    operand_ready(rh);                       // (the hi half on its own: the first product of a triple needs only it)
    if constexpr (P == 3) operand_ready(rl);
#else
    if constexpr (P == 3) operand_ready(rh, rl);
    else operand_ready(rh);
#endif
    FragT<P> r;
    r.hi = __builtin_bit_cast(bf16x8, rh);
    if constexpr (P == 3) r.lo = __builtin_bit_cast(bf16x8, rl);
    return r;
}
// accumulator registers 8t .. 8t+7 of a D tile as the operand of K-step t
template <int P>
__device__ __forceinline__ void split_tile(const f32x16& a, FragT<P> (&f)[2]) {
#pragma unroll
    for (int t = 0; t < 2; t++) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = a[8 * t + e];
        f[t] = split8<P>(v);
    }
}
// registers of a tile whose values ARE bf16 numbers (a flipped tile) -> operand halves, exactly
__device__ __forceinline__ bf16x8 exact8(const f32x16& d, int t) {
    i32x4 r;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const f32x2 f = {d[8 * t + 2 * p], d[8 * t + 2 * p + 1]};
        r[p] = __builtin_bit_cast(int, __builtin_convertvector(f, bf16x2));
    }
    operand_ready(r);
    return __builtin_bit_cast(bf16x8, r);
}
// selection matrices B[k][c] of the flips, as B operands of lane (c, h): k = 8h + e
//   kind 0: k == c            (an operand in natural order: dL/dY, outputs 8h + e)
//   kind 1: nrow(e, h) == c   (registers 0..7 of a D tile: neurons 0..15 of its block)
//   kind 2: 16 + nrow(e, h) == c   (registers 8..15: neurons 16..31)
__device__ __forceinline__ bf16x8 selector(int c, int h, int kind) {
    bf16x8 f;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int k = kind == 0 ? 8 * h + e : nrow(e, h) + (kind == 2 ? 16 : 0);
        f[e] = k == c ? (elem16)1.0f : (elem16)0.0f;
    }
    return f;
}
// D tile (as its two K-step operands) -> the same 32 x 32 block with the neuron on the lanes: lane (c, h) gets
// T[c][sample nrow(q, h)], q = 0..15, again as two K-step operands (contraction over samples)
template <int P>
__device__ __forceinline__ void flip_tile(const FragT<P> (&f)[2], bf16x8 selA, bf16x8 selB, FragT<P> (&out)[2]) {
    f32x16 dh = (f32x16)(0.0f);
    dh = mmab(f[0].hi, selA, dh);
    dh = mmab(f[1].hi, selB, dh);
#pragma unroll
    for (int t = 0; t < 2; t++) out[t].hi = exact8(dh, t);
    if constexpr (P == 3) {
        f32x16 dl = (f32x16)(0.0f);
        dl = mmab(f[0].lo, selA, dl);
        dl = mmab(f[1].lo, selB, dl);
#pragma unroll
        for (int t = 0; t < 2; t++) out[t].lo = exact8(dl, t);
    }
}
template <int P>
__device__ __forceinline__ void flip_natural(const FragT<P>& f, bf16x8 selN, FragT<P> (&out)[2]) {
    f32x16 dh = mmab(f.hi, selN, (f32x16)(0.0f));
#pragma unroll
    for (int t = 0; t < 2; t++) out[t].hi = exact8(dh, t);
    if constexpr (P == 3) {
        f32x16 dl = mmab(f.lo, selN, (f32x16)(0.0f));
#pragma unroll
        for (int t = 0; t < 2; t++) out[t].lo = exact8(dl, t);
    }
}

}  // namespace enerf_mlp32
