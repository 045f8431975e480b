// shencoder.hip -- real spherical-harmonics direction encoding (degree 1..8) for gfx950.
//
// Replaces shencoder/src/shencoder.cu of the reference (sh_encode_forward / sh_encode_backward).
// The reference writes the 64 basis polynomials and their 192 partial derivatives out as literals; they are
//     Y_l^m(x,y,z) = N_l^m * Q_l^m(z) * Re/Im (x + i y)^|m|,   index l*l + l + m,
// with Q_l^m(z) = P_l^m(z) / (1 - z^2)^(m/2) (Condon-Shortley phase) a polynomial in z, and x, y, z treated as
// independent variables (inputs are not normalised).  Here the same polynomials are produced by fully unrolled
// compile-time recurrences, one thread per direction; normalisation constants come from the host in a by-value
// table.  Pure ALU work: 12 B in, 4*C^2 B out per point.
#include <hip/hip_fp16.h>
#include <math.h>

#include "common.h"

using namespace enerf;

namespace {

struct ShNorm {
    float n[8][8];  // n[l][m], m <= l, includes sqrt(2) for m > 0
};

__device__ __forceinline__ float ld_f(const float* p) { return *p; }
__device__ __forceinline__ float ld_f(const __half* p) { return __half2float(*p); }
__device__ __forceinline__ void st_f(float* p, float v) { *p = v; }
__device__ __forceinline__ void st_f(__half* p, float v) { *p = __float2half(v); }

template <typename T, int DEG, bool JAC>
__global__ void __launch_bounds__(256) k_sh_fwd(const T* __restrict__ inputs, T* __restrict__ outputs, uint32_t B,
                                                T* __restrict__ dy_dx, ShNorm nrm, uint32_t out_stride) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    constexpr int C2 = DEG * DEG;
    const float x = ld_f(inputs + (size_t)b * 3), y = ld_f(inputs + (size_t)b * 3 + 1),
                z = ld_f(inputs + (size_t)b * 3 + 2);

    // (x + i y)^m = A[m] + i Bm[m]
    float A[DEG], Bm[DEG];
    A[0] = 1.0f;
    Bm[0] = 0.0f;
#pragma unroll
    for (int m = 1; m < DEG; m++) {
        A[m] = x * A[m - 1] - y * Bm[m - 1];
        Bm[m] = x * Bm[m - 1] + y * A[m - 1];
    }

    float Y[C2];
    float dX[JAC ? C2 : 1], dY[JAC ? C2 : 1], dZ[JAC ? C2 : 1];

#pragma unroll
    for (int m = 0; m < DEG; m++) {
        // Q_m^m = (-1)^m (2m-1)!!
        float qmm = 1.0f;
#pragma unroll
        for (int k = 1; k <= m; k++) qmm *= -(2.0f * k - 1.0f);
        float Q[DEG], dQ[DEG];
        Q[m] = qmm;
        dQ[m] = 0.0f;
        if (m + 1 < DEG) {
            Q[m + 1] = (2.0f * m + 1.0f) * z * qmm;
            dQ[m + 1] = (2.0f * m + 1.0f) * qmm;
        }
#pragma unroll
        for (int l = m + 2; l < DEG; l++) {
            const float inv = 1.0f / (float)(l - m);
            Q[l] = ((2.0f * l - 1.0f) * z * Q[l - 1] - (float)(l + m - 1) * Q[l - 2]) * inv;
            if (JAC) dQ[l] = ((2.0f * l - 1.0f) * (Q[l - 1] + z * dQ[l - 1]) - (float)(l + m - 1) * dQ[l - 2]) * inv;
        }
        // partials of (x+iy)^m
        const float Ax = m ? (float)m * A[m ? m - 1 : 0] : 0.0f;
        const float Ay = m ? -(float)m * Bm[m ? m - 1 : 0] : 0.0f;
        const float Bx = m ? (float)m * Bm[m ? m - 1 : 0] : 0.0f;
        const float By = m ? (float)m * A[m ? m - 1 : 0] : 0.0f;
#pragma unroll
        for (int l = m; l < DEG; l++) {
            const float nq = nrm.n[l][m] * Q[l];
            const int ip = l * l + l + m, in_ = l * l + l - m;
            Y[ip] = nq * A[m];
            if (JAC) {
                dX[ip] = nq * Ax;
                dY[ip] = nq * Ay;
                dZ[ip] = nrm.n[l][m] * dQ[l] * A[m];
            }
            if (m) {
                Y[in_] = nq * Bm[m];
                if (JAC) {
                    dX[in_] = nq * Bx;
                    dY[in_] = nq * By;
                    dZ[in_] = nrm.n[l][m] * dQ[l] * Bm[m];
                }
            }
        }
    }

    T* out = outputs + (size_t)b * out_stride;
    if constexpr (sizeof(T) == 4 && (C2 % 4) == 0) {
#pragma unroll
        for (int i = 0; i < C2; i += 4)
            *reinterpret_cast<float4*>(out + i) = make_float4(Y[i], Y[i + 1], Y[i + 2], Y[i + 3]);
    } else {
#pragma unroll
        for (int i = 0; i < C2; i++) st_f(out + i, Y[i]);
    }
    if (JAC) {
        T* j = dy_dx + (size_t)b * 3 * C2;
#pragma unroll
        for (int i = 0; i < C2; i++) {
            st_f(j + i, dX[i]);
            st_f(j + C2 + i, dY[i]);
            st_f(j + 2 * C2 + i, dZ[i]);
        }
    }
}

// grad_inputs[b,d] += sum_ch grad[b,ch] * dy_dx[b,d,ch]      (shencoder.cu:359-383)
template <typename T>
__global__ void __launch_bounds__(256) k_sh_bwd(const T* __restrict__ grad, uint32_t B, uint32_t D, uint32_t C2,
                                                const T* __restrict__ dy_dx, T* grad_inputs) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = t / D;
    if (b >= B) return;
    const uint32_t d = t - b * D;
    const T* g = grad + (size_t)b * C2;
    const T* j = dy_dx + ((size_t)b * D + d) * C2;
    float acc = ld_f(grad_inputs + t);
    for (uint32_t ch = 0; ch < C2; ch++) acc = fmaf(ld_f(g + ch), ld_f(j + ch), acc);
    st_f(grad_inputs + t, acc);
}

void fill_norm(ShNorm& nrm) {
    for (int l = 0; l < 8; l++)
        for (int m = 0; m < 8; m++) {
            if (m > l) {
                nrm.n[l][m] = 0.0f;
                continue;
            }
            double ratio = 1.0;
            for (int k = l - m + 1; k <= l + m; k++) ratio /= (double)k;
            double v = sqrt((2.0 * l + 1.0) / (4.0 * M_PI) * ratio);
            if (m) v *= sqrt(2.0);
            nrm.n[l][m] = (float)v;
        }
}

template <typename T, int DEG>
void launch_sh(const T* in, T* out, uint32_t B, bool jac, T* dy_dx, const ShNorm& nrm, uint32_t out_stride,
               hipStream_t s) {
    const uint32_t stride = out_stride ? out_stride : (uint32_t)(DEG * DEG);
    if (jac)
        k_sh_fwd<T, DEG, true><<<div_up(B, 256), 256, 0, s>>>(in, out, B, dy_dx, nrm, stride);
    else
        k_sh_fwd<T, DEG, false><<<div_up(B, 256), 256, 0, s>>>(in, out, B, dy_dx, nrm, stride);
}

template <typename T>
int dispatch_sh(const T* in, T* out, uint32_t B, uint32_t C, bool jac, T* dy_dx, uint32_t out_stride, hipStream_t s) {
    ShNorm nrm;
    fill_norm(nrm);
    switch (C) {
        case 1: launch_sh<T, 1>(in, out, B, jac, dy_dx, nrm, out_stride, s); break;
        case 2: launch_sh<T, 2>(in, out, B, jac, dy_dx, nrm, out_stride, s); break;
        case 3: launch_sh<T, 3>(in, out, B, jac, dy_dx, nrm, out_stride, s); break;
        case 4: launch_sh<T, 4>(in, out, B, jac, dy_dx, nrm, out_stride, s); break;
        case 5: launch_sh<T, 5>(in, out, B, jac, dy_dx, nrm, out_stride, s); break;
        case 6: launch_sh<T, 6>(in, out, B, jac, dy_dx, nrm, out_stride, s); break;
        case 7: launch_sh<T, 7>(in, out, B, jac, dy_dx, nrm, out_stride, s); break;
        case 8: launch_sh<T, 8>(in, out, B, jac, dy_dx, nrm, out_stride, s); break;
        default: ENERF_BADARG("SH encoder only supports degree in [1, 8], got %u", C);
    }
    return 0;
}

}  // namespace

extern "C" {

int enerf_sh_encode_forward(const void* inputs, void* outputs, uint32_t B, uint32_t D, uint32_t C, int calc_grad_inputs,
                            void* dy_dx, int dtype, enerf_stream_t stream) {
    if (B == 0) return 0;
    if (D != 3) ENERF_BADARG("SH encoder only support input dim == 3, got %u", D);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(ENERF_K_SH_FWD, s);
    int rc;
    if (dtype == ENERF_F32) rc = dispatch_sh<float>((const float*)inputs, (float*)outputs, B, C, calc_grad_inputs != 0, (float*)dy_dx, 0, s);
    else if (dtype == ENERF_F16) rc = dispatch_sh<__half>((const __half*)inputs, (__half*)outputs, B, C, calc_grad_inputs != 0, (__half*)dy_dx, 0, s);
    else ENERF_BADARG("SH encoder: dtype must be f32 or f16");
    if (rc) return rc;
    ENERF_LAUNCH_CHECK("sh_encode_forward");
    return 0;
}

// extension: fp32 forward without Jacobian into rows of `out_stride` floats (e.g. columns 16..31 of a [B,32] buffer)
int enerf_sh_encode_forward_strided(const float* inputs, float* outputs, uint32_t B, uint32_t C, uint32_t out_stride,
                                    enerf_stream_t stream) {
    if (B == 0) return 0;
    if (out_stride < C * C) ENERF_BADARG("SH encoder: out_stride %u < degree^2", out_stride);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(ENERF_K_SH_FWD, s);
    const int rc = dispatch_sh<float>(inputs, outputs, B, C, false, nullptr, out_stride, s);
    if (rc) return rc;
    ENERF_LAUNCH_CHECK("sh_encode_forward_strided");
    return 0;
}

int enerf_sh_encode_backward(const void* grad, const void* inputs, uint32_t B, uint32_t D, uint32_t C, const void* dy_dx,
                             void* grad_inputs, int dtype, enerf_stream_t stream) {
    (void)inputs;
    if (B == 0) return 0;
    if (D != 3) ENERF_BADARG("SH encoder only support input dim == 3, got %u", D);
    if (C < 1 || C > 8) ENERF_BADARG("SH encoder only supports degree in [1, 8], got %u", C);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == ENERF_F32)
        k_sh_bwd<float><<<div_up(B * D, 256), 256, 0, s>>>((const float*)grad, B, D, C * C, (const float*)dy_dx, (float*)grad_inputs);
    else if (dtype == ENERF_F16)
        k_sh_bwd<__half><<<div_up(B * D, 256), 256, 0, s>>>((const __half*)grad, B, D, C * C, (const __half*)dy_dx, (__half*)grad_inputs);
    else ENERF_BADARG("SH encoder: dtype must be f32 or f16");
    ENERF_LAUNCH_CHECK("sh_encode_backward");
    return 0;
}

}  // extern "C"
