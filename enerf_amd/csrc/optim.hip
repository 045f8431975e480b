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

// All parameters of a model in one launch: the table travels as a kernel argument (no device-side descriptor to keep
// in sync); workgroup b works on tensor t with first_block[t] <= b < first_block[t + 1].
constexpr int kMaxAdamTensors = 16;
struct AdamTable {
    float* p[kMaxAdamTensors];
    float* g[kMaxAdamTensors];
    float* m[kMaxAdamTensors];
    float* v[kMaxAdamTensors];
    unsigned long long n[kMaxAdamTensors];
    float step_size[kMaxAdamTensors];
    float inv_bc2_sqrt[kMaxAdamTensors];
    uint32_t first_block[kMaxAdamTensors + 1];
    uint32_t count;
};

__global__ void __launch_bounds__(256) k_adam_multi(AdamTable tab, float b1, float b2, float eps, int zero_grad) {
    uint32_t t = 0;
    while (t + 1 < tab.count && blockIdx.x >= tab.first_block[t + 1]) t++;
    float* __restrict__ p = tab.p[t];
    float* __restrict__ g = tab.g[t];
    float* __restrict__ m = tab.m[t];
    float* __restrict__ v = tab.v[t];
    const size_t n = tab.n[t];
    const float step_size = tab.step_size[t], inv_bc2_sqrt = tab.inv_bc2_sqrt[t];
    const uint32_t nblk = tab.first_block[t + 1] - tab.first_block[t];
    const uint32_t blk = blockIdx.x - tab.first_block[t];
    const size_t n4 = n / 4;
    const size_t stride = (size_t)nblk * blockDim.x;
    for (size_t i = (size_t)blk * blockDim.x + threadIdx.x; i < n4; i += stride) {
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
    const size_t tl = n4 * 4 + (size_t)blk * blockDim.x + threadIdx.x;
    if (tl < n) {
        float P = p[tl], M = m[tl], V = v[tl];
        adam1(P, g[tl], M, V, b1, b2, eps, step_size, inv_bc2_sqrt);
        p[tl] = P; m[tl] = M; v[tl] = V;
        if (zero_grad) g[tl] = 0.0f;
    }
}

// torch.amp.GradScaler.update() (torch._amp_update_scale_) + the bookkeeping of a skipped step, one thread
__global__ void k_amp_update(float* scale, int32_t* growth_tracker, uint32_t* found_inf, uint32_t* skipped,
                             float growth_factor, float backoff_factor, int32_t growth_interval) {
    if (found_inf[0]) {
        scale[0] = scale[0] * backoff_factor;
        growth_tracker[0] = 0;
        skipped[0] += 1u;
        found_inf[0] = 0u;
    } else {
        const int32_t ok = growth_tracker[0] + 1;
        if (ok == growth_interval) {
            const float grown = scale[0] * growth_factor;
            if (fabsf(grown) <= 3.402823466e38f) scale[0] = grown;
            growth_tracker[0] = 0;
        } else {
            growth_tracker[0] = ok;
        }
    }
}

AmpState g_amp = {nullptr, nullptr, nullptr};
float* g_amp_scale = nullptr;
int32_t* g_amp_tracker = nullptr;
uint32_t* g_amp_skipped = nullptr;

}  // namespace

namespace enerf {
AmpState amp_state() { return g_amp; }
}  // namespace enerf

extern "C" {

// Loss scaling of the fp16 regime, the device side of torch.amp.GradScaler around a closed-form step (nerf/utils.py:964-975:
// scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()).  `scale` (fp32) and `growth_tracker` (int32)
// are the GradScaler's own device tensors, `found_inf` / `skipped` two uint32 words of the caller's (zero at first).
// enerf_amp_begin arms the kernels named at AmpState (common.h) for the calls that follow; enerf_amp_end queues the
// scale update on `stream` (backoff on a step with a non-finite weight gradient, which the optimizer launch has left
// unapplied; growth after `growth_interval` clean steps in a row) and disarms them.  No host synchronisation anywhere.
int enerf_amp_begin(float* scale, int32_t* growth_tracker, uint32_t* found_inf, uint32_t* skipped) {
    if (!scale || !growth_tracker || !found_inf || !skipped) ENERF_BADARG("amp_begin: four device pointers are required");
    g_amp = AmpState{scale, found_inf, skipped};
    g_amp_scale = scale;
    g_amp_tracker = growth_tracker;
    g_amp_skipped = skipped;
    return 0;
}

int enerf_amp_end(float growth_factor, float backoff_factor, int32_t growth_interval, enerf_stream_t stream) {
    if (!g_amp.scale) return 0;
    k_amp_update<<<1, 1, 0, (hipStream_t)stream>>>(g_amp_scale, g_amp_tracker, g_amp.found_inf, g_amp_skipped, growth_factor,
                                                   backoff_factor, growth_interval);
    g_amp = AmpState{nullptr, nullptr, nullptr};
    ENERF_LAUNCH_CHECK("amp_end");
    return 0;
}

// (an aborted step: disarm without an update)
// 1 between enerf_amp_begin and enerf_amp_end / _cancel (callers whose optimizer path is not AMP-aware must refuse to run then)
int enerf_amp_armed(void) { return g_amp.scale ? 1 : 0; }

int enerf_amp_cancel(void) {
    g_amp = AmpState{nullptr, nullptr, nullptr};
    return 0;
}

// The same update for up to 16 parameters in one launch (per-tensor lr and step count; shared betas / eps).
int enerf_adam_step_multi(uint32_t count, float* const* p, float* const* g, float* const* m, float* const* v,
                          const size_t* n, const float* lr, const uint32_t* step, float beta1, float beta2, float eps,
                          int zero_grad, enerf_stream_t stream) {
    if (count == 0) return 0;
    if (count > (uint32_t)kMaxAdamTensors) ENERF_BADARG("adam_step_multi: at most %d tensors per call, got %u", kMaxAdamTensors, count);
    AdamTable tab;
    uint32_t blocks = 0;
    for (uint32_t t = 0; t < count; t++) {
        if ((((uintptr_t)p[t] | (uintptr_t)g[t] | (uintptr_t)m[t] | (uintptr_t)v[t]) & 15) != 0)
            ENERF_BADARG("adam_step_multi: p/g/m/v must be 16-byte aligned");
        if (step[t] == 0) ENERF_BADARG("adam_step_multi: step counts from 1");
        if (n[t] == 0) ENERF_BADARG("adam_step_multi: empty tensor");
        tab.p[t] = p[t]; tab.g[t] = g[t]; tab.m[t] = m[t]; tab.v[t] = v[t];
        tab.n[t] = n[t];
        const double bc1 = 1.0 - pow((double)beta1, (double)step[t]);
        const double bc2 = 1.0 - pow((double)beta2, (double)step[t]);
        tab.step_size[t] = (float)((double)lr[t] / bc1);
        tab.inv_bc2_sqrt[t] = (float)(1.0 / sqrt(bc2));
        const size_t n4 = n[t] / 4 ? n[t] / 4 : 1;
        uint32_t b = (uint32_t)((n4 + 255) / 256);
        if (b > 2048u) b = 2048u;
        tab.first_block[t] = blocks;
        blocks += b;
    }
    tab.first_block[count] = blocks;
    tab.count = count;
    k_adam_multi<<<blocks, 256, 0, (hipStream_t)stream>>>(tab, beta1, beta2, eps, zero_grad);
    ENERF_LAUNCH_CHECK("adam_step_multi");
    return 0;
}


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
