// Microbenchmark: float atomic-add throughput on MI355X by memory scope and address pattern.
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_rate.hip -o /tmp/atomic_rate && /tmp/atomic_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int SCOPE>   // 0 = agent (atomicAdd default), 1 = workgroup scope, 2 = agent + xcd-partitioned addresses
__global__ void __launch_bounds__(256) k_atomics(float* buf, uint32_t mask, uint32_t per_thread, uint32_t xcd_part) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t xcc = 0;
    if (xcd_part) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
    }
    for (uint32_t i = 0; i < per_thread; i++) {
        uint32_t a = hash32(tid * per_thread + i) & mask;
        if (xcd_part) a = (a & (mask >> 3)) | (xcc * ((mask + 1) >> 3));   // disjoint 1/8 of the buffer per XCD
        float* p = buf + 2 * a;
        if (SCOPE == 1) {
            __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(p + 1, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(p + 1, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ void k_xcc_census(uint32_t* hist) {
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) atomicAdd(&hist[(xcc & 7u) * 2 + ((blockIdx.x & 7u) == (xcc & 7u) ? 0 : 1)], 1u);
}

int main() {
    const uint32_t nblocks = 4096, per_thread = 16;
    const double natom = 2.0 * nblocks * 256.0 * per_thread;
    float* buf;
    hipMalloc(&buf, sizeof(float) * 2 * (1u << 24));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    uint32_t* hist; hipMalloc(&hist, 64); hipMemset(hist, 0, 64);
    k_xcc_census<<<4096, 64>>>(hist);
    uint32_t h[16]; hipMemcpy(h, hist, 64, hipMemcpyDeviceToHost);
    printf("XCC census (blocks with bid%%8==xcc / others):");
    for (int i = 0; i < 8; i++) printf(" [%u/%u]", h[2 * i], h[2 * i + 1]);
    printf("\n");
    for (uint32_t lg : {12u, 16u, 19u, 22u}) {       // distinct float2 rows: 4K, 64K, 512K (= one hashed level), 4M
        const uint32_t mask = (1u << lg) - 1;
        for (int mode = 0; mode < 4; mode++) {
            hipMemset(buf, 0, sizeof(float) * 2 * (1u << 24));
            float best = 1e9;
            for (int rep = 0; rep < 4; rep++) {
                hipEventRecord(a);
                if (mode == 0) k_atomics<0><<<nblocks, 256>>>(buf, mask, per_thread, 0);
                if (mode == 1) k_atomics<1><<<nblocks, 256>>>(buf, mask, per_thread, 0);
                if (mode == 2) k_atomics<0><<<nblocks, 256>>>(buf, mask, per_thread, 1);
                if (mode == 3) k_atomics<1><<<nblocks, 256>>>(buf, mask, per_thread, 1);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            // verify total (sum of buffer == natom*4 reps) only meaningful for coherent modes
            std::vector<float> hbuf(2 * (size_t)(mask + 1));
            hipMemcpy(hbuf.data(), buf, sizeof(float) * hbuf.size(), hipMemcpyDeviceToHost);
            double sum = 0; for (float v : hbuf) sum += v;
            const char* names[4] = {"agent scope, any XCD      ", "workgroup scope, any XCD  ", "agent scope, XCD-owned    ",
                                    "workgroup scope, XCD-owned"};
            printf("rows=2^%-2u %s: %7.3f ms  %6.1f G atomics/s   sum/expected = %.4f\n", lg, names[mode], best,
                   natom / best / 1e6, sum / (natom * 4));
        }
    }
    return 0;
}
