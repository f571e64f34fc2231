#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16; return x; }
__global__ void k(uint64_t* slab, uint64_t n_rec, uint32_t per_thread) {
    uint32_t h = mix(blockIdx.x * 1024u + threadIdx.x);
    for (uint32_t i = 0; i < per_thread; i++) {
        h = mix(h + i);
        const uint32_t h2 = mix(h ^ 0x9E3779B9u);
        const uint64_t r = (uint64_t)(((unsigned __int128)(((uint64_t)h << 32) | h2) * n_rec) >> 64);
        uint64_t* o = slab + r * 3;
        o[0] = h; o[1] = h2; o[2] = i;
    }
}
int main() {
    const size_t bytes = 26ull << 30;
    const uint64_t n_rec = bytes / 24;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    uint64_t* s[5];
    for (int v = 0; v < 5; v++) if (hipMalloc((void**)&s[v], bytes) != hipSuccess) { s[v] = nullptr; (void)hipGetLastError(); }
    for (int round = 0; round < 2; round++)
    for (int v = 0; v < 5; v++) {
        if (!s[v]) continue;
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a);
            k<<<256 * 16 * 8, 256>>>(s[v], n_rec, 16);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("block %d (%p): %.2f ms\n", v, (void*)s[v], best);
    }
    return 0;
}
