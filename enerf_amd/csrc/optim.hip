// optim.hip -- fused Adam step (main_nerf.py:211: Adam(betas=(0.9, 0.99), eps=1e-15), no weight decay / amsgrad).
//
// The step right after the hot path.  torch's foreach Adam makes ~7 passes over the 52 MB hash table and its two
// moment buffers; this kernel makes one: read p, g, m, v -> write p, m, v (28 B/element, HBM-bound), 16-byte
// accesses, grid-stride over <= 2048 workgroups.
#include "common.h"

using namespace enerf;

namespace {

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float b1, float b2, float eps,
                                      float step_size, float inv_bc2_sqrt) {
    m = fmaf(g - m, 1.0f - b1, m);                    // exp_avg.lerp_(grad, 1 - beta1)
    v = fmaf((1.0f - b2) * g, g, v * b2);            // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
    p = p - step_size * (m / denom);                  // param.addcdiv_(exp_avg, denom, value = -step_size)
}

__global__ void __launch_bounds__(256) k_adam(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, size_t n, float b1, float b2, float eps,
                                              float step_size, float inv_bc2_sqrt, int zero_grad) {
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 P = reinterpret_cast<float4*>(p)[i];
        const float4 G = reinterpret_cast<const float4*>(g)[i];
        float4 M = reinterpret_cast<float4*>(m)[i];
        float4 V = reinterpret_cast<float4*>(v)[i];
        adam1(P.x, G.x, M.x, V.x, b1, b2, eps, step_size, inv_bc2_sqrt);
        adam1(P.y, G.y, M.y, V.y, b1, b2, eps, step_size, inv_bc2_sqrt);
        adam1(P.z, G.z, M.z, V.z, b1, b2, eps, step_size, inv_bc2_sqrt);
        adam1(P.w, G.w, M.w, V.w, b1, b2, eps, step_size, inv_bc2_sqrt);
        reinterpret_cast<float4*>(p)[i] = P;
        reinterpret_cast<float4*>(m)[i] = M;
        reinterpret_cast<float4*>(v)[i] = V;
        if (zero_grad) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // tail
    const size_t t = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        float P = p[t], M = m[t], V = v[t];
        adam1(P, g[t], M, V, b1, b2, eps, step_size, inv_bc2_sqrt);
        p[t] = P; m[t] = M; v[t] = V;
        if (zero_grad) g[t] = 0.0f;
    }
}

}  // namespace

extern "C" {

// One Adam update of a contiguous fp32 parameter: `step` is the 1-based step count used for bias correction.
// p, g, m, v must be 16-byte aligned.  zero_grad != 0 also clears g in the same pass.
int enerf_adam_step(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                    uint32_t step, int zero_grad, enerf_stream_t stream) {
    if (n == 0) return 0;
    if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) != 0)
        ENERF_BADARG("adam_step: p/g/m/v must be 16-byte aligned");
    if (step == 0) ENERF_BADARG("adam_step: step counts from 1");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    const size_t n4 = n / 4 ? n / 4 : 1;
    uint32_t blocks = (uint32_t)((n4 + 255) / 256);
    if (blocks > 2048u) blocks = 2048u;
    k_adam<<<blocks, 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, beta1, beta2, eps, step_size, inv_bc2_sqrt, zero_grad);
    ENERF_LAUNCH_CHECK("adam_step");
    return 0;
}

}  // extern "C"
