// micro-benchmark (round 4, dense path): random device-scope atomics WITH and WITHOUT a returned value, 64-bit adds on a table of
// 2^22 (33 MB: k = 11) and 2^30 (8.6 GB: k = 15) entries, and the same with a plain load of the entry next to a non-returning add
// (what the dense path would do to learn which Exts bits are still missing).   hipcc --offload-arch=gfx950 -O3 atomic_noret.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16; return x; }
template <int MODE>   // 0 returning add, 1 non-returning add, 2 non-returning add + plain load
__global__ void k(unsigned long long* tab, uint32_t bits, uint32_t per_thread, unsigned long long* sink) {
    uint32_t h = mix(blockIdx.x * 1024u + threadIdx.x);
    unsigned long long acc = 0;
    for (uint32_t i = 0; i < per_thread; i++) {
        h = mix(h + i);
        const uint32_t b = h >> (32 - bits);
        if (MODE == 0) acc += atomicAdd(&tab[b], 1ull);
        else {
            if (MODE == 2) acc += __builtin_nontemporal_load(&tab[b]);
            atomicAdd(&tab[b], 1ull);
        }
    }
    if (acc == 0xdeadbeefull) *sink = acc;
}
int main() {
    unsigned long long *tab, *sink;
    hipMalloc(&tab, (size_t)8 << 30); hipMalloc(&sink, 8);
    hipMemset(tab, 0, (size_t)8 << 30);
    const uint32_t blocks = 256 * 16 * 4, threads = 256, per_thread = 128;     // 5.4e8 atomics
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (uint32_t bits : {22u, 30u}) {
        auto run = [&](const char* name, auto kern) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(a);
                kern<<<blocks, threads>>>(tab, bits, per_thread, sink);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            printf("2^%u entries  %-34s %.2f ms  %.1f G/s\n", bits, name, best, (double)blocks * threads * per_thread / best / 1e6);
        };
        run("returning add", k<0>);
        run("non-returning add", k<1>);
        run("non-returning add + plain load", k<2>);
    }
    return 0;
}
