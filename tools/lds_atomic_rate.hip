// Microbenchmark: LDS atomic throughput per CU on MI355X (random addresses in a 128 KiB tile, 1024 threads / WG).
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lds_atomic_rate.hip -o tools/lds_atomic_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// MODE 0: ds_add_f32   1: ds_add_u32   2: plain read+add+write (racy, rate only)   3: ds_add_f32 on float2 (both)
// 4: ds_add_rtn_f32 (returning)  5: ds_pk_add? n/a -> ds_add_f64
template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, uint32_t iters, uint32_t mask) {
    __shared__ __attribute__((aligned(16))) float acc[32768];
    for (uint32_t i = threadIdx.x; i < 32768; i += 1024) acc[i] = 0.f;
    __syncthreads();
    uint32_t s = hash32(blockIdx.x * 1024 + threadIdx.x);
    float r = 0.f;
    for (uint32_t i = 0; i < iters; i++) {
        s = s * 1664525u + 1013904223u;
        const uint32_t a = (s >> 8) & mask;
        if (MODE == 0) atomicAdd(&acc[a], 1.0f);
        else if (MODE == 1) atomicAdd(reinterpret_cast<uint32_t*>(&acc[a]), 1u);
        else if (MODE == 2) acc[a] += 1.0f;
        else if (MODE == 3) { atomicAdd(&acc[a & ~1u], 1.0f); atomicAdd(&acc[a | 1u], 2.0f); }
        else if (MODE == 4) r += atomicAdd(&acc[a], 1.0f);
        else if (MODE == 5) atomicAdd(reinterpret_cast<double*>(&acc[a & ~1u]), 1.0);
    }
    __syncthreads();
    float t = r;
    for (uint32_t i = threadIdx.x; i < 32768; i += 1024) t += acc[i];
    if (t == 123.456f) out[0] = t;
}

template <int MODE>
void run(const char* name, float* out, uint32_t mask) {
    const uint32_t iters = 256, blocks = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 1024>>>(out, iters, mask);
    hipEventRecord(a);
    k<MODE><<<blocks, 1024>>>(out, iters, mask);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double ops = (double)blocks * 1024 * iters * (MODE == 3 ? 2 : 1);
    printf("%-28s mask %5x: %7.3f ms  %7.1f G lane-ops/s chip  (%.2f cycles/lane-op/CU at 2.4 GHz)\n", name, mask, ms,
           ops / ms / 1e6, 2.4e9 / (ops / blocks / (ms * 1e-3)));
}

int main() {
    float* out; hipMalloc(&out, 64);
    for (uint32_t mask : {0x7fffu, 0xffu}) {
        run<0>("ds_add_f32", out, mask);
        run<1>("ds_add_u32", out, mask);
        run<2>("plain read-add-write", out, mask);
        run<3>("ds_add_f32 x2 (pair)", out, mask);
        run<4>("ds_add_rtn_f32", out, mask);
        run<5>("ds_add_f64", out, mask);
    }
    return 0;
}
