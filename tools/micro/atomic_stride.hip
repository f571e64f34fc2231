// micro-benchmark (round 4, scan): does the rate of random returning device-scope atomics on 1.1e6 counters depend on how the counters are
// laid out -- packed (16 per 64-byte line), one per 16 / 64 / 128 bytes?   hipcc --offload-arch=gfx950 -O3 atomic_stride.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16; return x; }
__global__ void k(uint32_t* cur, uint32_t nb, uint32_t stride, uint32_t per_thread, uint32_t* sink) {
    uint32_t h = mix(blockIdx.x * 1024u + threadIdx.x), acc = 0;
    for (uint32_t i = 0; i < per_thread; i++) {
        h = mix(h + i);
        const uint32_t b = (uint32_t)(((uint64_t)h * nb) >> 32);
        acc += atomicAdd(&cur[(size_t)b * stride], 1u);
    }
    if (acc == 0xdeadbeefu) *sink = acc;
}
int main() {
    const uint32_t nb = 1100000, per_thread = 64;
    uint32_t *cur, *sink;
    hipMalloc(&cur, (size_t)nb * 32 * 4 + 4096); hipMalloc(&sink, 4);
    hipMemset(cur, 0, (size_t)nb * 32 * 4);
    const uint32_t blocks = 256 * 16 * 8, threads = 256;          // 5.4e8 atomics
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (uint32_t stride : {1u, 4u, 16u, 32u}) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a);
            k<<<blocks, threads>>>(cur, nb, stride, per_thread, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("one counter per %3u bytes: %.2f ms  %.1f G atomics/s\n", stride * 4, best, (double)blocks * threads * per_thread / best / 1e6);
    }
    return 0;
}
