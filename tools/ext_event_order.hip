// Does hipStreamWaitEvent on the STOP event of a hipExtLaunchKernelGGL launch order another stream after that kernel?
// (enerf_mlp32_signal_next_reduce / enerf_stream_wait_mlp32_signal rely on it.)  Kernel A spins ~2 ms on stream 1 and
// then writes a word; stream 2 waits for A's stop event and runs kernel B, which copies the word.  B must see it, every
// time, also when the event object is reused from iteration to iteration.
// hipcc --offload-arch=gfx950 -O3 tools/ext_event_order.hip -o /tmp/ext_event_order && timeout 60 /tmp/ext_event_order
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdint.h>

__global__ void k_slow(uint32_t* word, uint32_t value, uint64_t spin) {
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) *word = value;
}
__global__ void k_copy(const uint32_t* word, uint32_t* out) { *out = *word; }

int main() {
    uint32_t *word, *out;
    hipMalloc(&word, 4);
    hipMalloc(&out, 4);
    hipStream_t s1, s2;
    hipStreamCreate(&s1);
    hipStreamCreate(&s2);
    hipEvent_t ev;
    hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    int bad = 0;
    for (uint32_t it = 1; it <= 200; it++) {
        // 100 MHz wall clock: 200000 ticks = 2 ms
        hipExtLaunchKernelGGL(k_slow, dim3(1), dim3(64), 0, s1, nullptr, ev, 0, word, it, (uint64_t)200000);
        hipStreamWaitEvent(s2, ev, 0);
        hipLaunchKernelGGL(k_copy, dim3(1), dim3(1), 0, s2, (const uint32_t*)word, out);
        uint32_t h = 0;
        hipMemcpyAsync(&h, out, 4, hipMemcpyDeviceToHost, s2);
        hipStreamSynchronize(s2);
        if (h != it) bad++;
    }
    printf("stop-event ordering: %d of 200 iterations saw a stale word\n", bad);
    hipDeviceSynchronize();
    return bad != 0;
}
