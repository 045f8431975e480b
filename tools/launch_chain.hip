// Microbenchmark: the bubble between dependent kernels of one HIP stream on MI355X, and what it costs to replace the
// stream's barrier by a device-side flag (kernels launched with hipExtAnyOrderLaunch start while their predecessor
// drains, set up, then wait for the predecessor's completion counter).
// hipcc --offload-arch=gfx950 -O3 tools/launch_chain.hip -o /tmp/launch_chain && timeout 60 /tmp/launch_chain
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdint.h>
#include <chrono>

// every workgroup: optional wait for `wait_for` (counter reaching `expect`), then `work` rounds of a dependent FMA chain
// over a small buffer, then signal `done`
__global__ void __launch_bounds__(256) k_stage(float* buf, uint32_t work, const uint32_t* wait_for, uint32_t expect,
                                               uint32_t* done, uint32_t setup) {
    __shared__ float lds[4096];
    // "set-up": something a real kernel does before it needs its predecessor's output
    for (uint32_t r = 0; r < setup; r++)
        for (uint32_t i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)(i + r);
    __syncthreads();
    if (wait_for) {
        if (threadIdx.x == 0) {
            while (__hip_atomic_load(wait_for, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < expect)
                __builtin_amdgcn_s_sleep(4);
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    float x = buf[tid] + lds[threadIdx.x];
    for (uint32_t r = 0; r < work; r++) x = fmaf(x, 1.0000001f, 0.5f);
    buf[tid] = x;
    if (done) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main() {
    const uint32_t nwg = 512, chain = 8, iters = 200;
    float* buf;
    uint32_t* cnt;
    hipMalloc(&buf, nwg * 256 * sizeof(float));
    hipMemset(buf, 0, nwg * 256 * sizeof(float));
    hipMalloc(&cnt, sizeof(uint32_t) * chain * (iters + 8));
    hipStream_t s;
    hipStreamCreate(&s);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEvent_t evs[16];
    for (int k = 0; k < 16; k++) hipEventCreate(&evs[k]);
    for (uint32_t work : {200u, 4000u}) {
        for (int mode = 0; mode < 6; mode++) {
            if (mode == 1 || mode == 2) continue;      // (any-order + counters: measured once, 6x slower)
            // mode 0: plain stream order; 1: any-order launches + counters (no set-up); 2: same with 2 us of set-up;
            // 3: every kernel carries a stop event (hipExtLaunchKernelGGL); 4: hipEventRecord after every kernel;
            // 5: start + stop events on every kernel
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                hipMemsetAsync(cnt, 0, sizeof(uint32_t) * chain * (iters + 8), s);
                hipStreamSynchronize(s);
                hipEventRecord(e0, s);
                for (uint32_t it = 0; it < iters; it++)
                    for (uint32_t k = 0; k < chain; k++) {
                        const uint32_t idx = it * chain + k;
                        if (mode == 3 || mode == 5) {
                            hipExtLaunchKernelGGL(k_stage, dim3(nwg), dim3(256), 0, s, mode == 5 ? evs[2 * k] : nullptr,
                                                  evs[2 * k + 1], 0, buf, work, (const uint32_t*)nullptr, 0u,
                                                  (uint32_t*)nullptr, 8u);
                        } else if (mode == 4) {
                            hipLaunchKernelGGL(k_stage, dim3(nwg), dim3(256), 0, s, buf, work, (const uint32_t*)nullptr, 0u,
                                               (uint32_t*)nullptr, 8u);
                            hipEventRecord(evs[2 * k + 1], s);
                        } else if (mode == 0) {
                            hipLaunchKernelGGL(k_stage, dim3(nwg), dim3(256), 0, s, buf, work, (const uint32_t*)nullptr, 0u,
                                               (uint32_t*)nullptr, 8u);
                        } else {
                            hipExtLaunchKernelGGL(k_stage, dim3(nwg), dim3(256), 0, s, nullptr, nullptr,
                                                  hipExtAnyOrderLaunch, buf, work,
                                                  idx ? (const uint32_t*)(cnt + idx - 1) : (const uint32_t*)nullptr, nwg,
                                                  cnt + idx, 8u);
                        }
                    }
                hipEventRecord(e1, s);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("work %6u  mode %d  %.2f us per kernel\n", work, mode, best * 1e3f / (iters * chain));
        }
    }
    return 0;
}
