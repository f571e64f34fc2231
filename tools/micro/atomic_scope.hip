// micro-benchmark: random returning atomics on 1.1e6 32-bit counters, agent scope vs workgroup scope (XCD-local L2),
// with the counter array either shared by all XCDs or private per XCD.   hipcc --offload-arch=gfx950 -O3 atomic_scope.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16; return x; }
template <int SCOPE, bool PER_XCD>
__global__ void k(uint32_t* cur, uint32_t nb, uint32_t per_thread, uint32_t* sink) {
    uint32_t xcd = 0;
    if (PER_XCD) { asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcd)); xcd &= 7u; }
    uint32_t* c = cur + (size_t)xcd * nb;
    uint32_t h = mix(blockIdx.x * 1024u + threadIdx.x), acc = 0;
    for (uint32_t i = 0; i < per_thread; i++) {
        h = mix(h + i);
        const uint32_t b = (uint32_t)(((uint64_t)h * nb) >> 32);
        acc += __hip_atomic_fetch_add(&c[b], 1u, __ATOMIC_RELAXED, SCOPE);
    }
    if (acc == 0xdeadbeefu) *sink = acc;
}
int main() {
    const uint32_t nb = 1100000, per_thread = 64;
    uint32_t *cur, *sink;
    hipMalloc(&cur, (size_t)8 * nb * 4); hipMalloc(&sink, 4);
    hipMemset(cur, 0, (size_t)8 * nb * 4);
    const uint32_t blocks = 256 * 16 * 8, threads = 256;          // 8.4M threads x 64 = 5.4e8 atomics
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char* name, auto kern) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a);
            kern<<<blocks, threads>>>(cur, nb, per_thread, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep == 2) printf("%-40s %.2f ms  %.1f G atomics/s\n", name, ms, (double)blocks * threads * per_thread / ms / 1e6);
        }
    };
    run("agent scope, shared counters", k<__HIP_MEMORY_SCOPE_AGENT, false>);
    run("workgroup scope, per-XCD counters", k<__HIP_MEMORY_SCOPE_WORKGROUP, true>);
    run("agent scope, per-XCD counters", k<__HIP_MEMORY_SCOPE_AGENT, true>);
    run("wavefront scope, per-XCD counters", k<__HIP_MEMORY_SCOPE_WAVEFRONT, true>);
    // check the per-XCD workgroup-scope result adds up after kernel end
    return 0;
}
