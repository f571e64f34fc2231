// Map of block speeds: 120 blocks of 2 GB allocated one after another, each probed with random 24-byte writes (whole block) and a
// streaming copy (first half -> second half).  Which fraction of the memory is of the fast kind, and does it come in runs?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16; return x; }
__global__ void k_rand(uint64_t* slab, uint64_t n_rec, uint32_t per_thread) {
    uint32_t h = mix(blockIdx.x * 1024u + threadIdx.x);
    for (uint32_t i = 0; i < per_thread; i++) {
        h = mix(h + i);
        const uint32_t h2 = mix(h ^ 0x9E3779B9u);
        const uint64_t r = (uint64_t)(((unsigned __int128)(((uint64_t)h << 32) | h2) * n_rec) >> 64);
        uint64_t* o = slab + r * 3;
        o[0] = h; o[1] = h2; o[2] = i;
    }
}
__global__ void k_copy(const uint4* __restrict__ in, uint4* __restrict__ out, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = in[i];
}
int main() {
    const size_t bytes = 2ull << 30;
    const int NB = 120;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    static uint64_t* s[NB];
    for (int v = 0; v < NB; v++) if (hipMalloc((void**)&s[v], bytes) != hipSuccess) { s[v] = nullptr; (void)hipGetLastError(); }
    auto timeit = [&](auto fn) { float best = 1e9; for (int rep = 0; rep < 3; rep++) { hipEventRecord(a); fn(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; } return best; };
    printf("block: random-write ms / copy TB/s\n");
    for (int v = 0; v < NB; v++) {
        if (!s[v]) { printf("%d: -\n", v); continue; }
        const uint64_t half16 = bytes / 32;
        const float t1 = timeit([&] { k_rand<<<256 * 16 * 2, 256>>>(s[v], bytes / 24, 16); });
        const float t2 = timeit([&] { k_copy<<<256 * 32, 256>>>((const uint4*)s[v], (uint4*)s[v] + half16, half16); });
        printf("%3d %p: %.2f / %.2f%s", v, (void*)s[v], t1, 2.0 * 1.0737 / t2, (v % 3 == 2) ? "\n" : "   ");
    }
    printf("\n");
    return 0;
}
