// Do the fast / slow kinds of a block (slab_probe2.hip) show in other access patterns?  Per block: random 24-byte writes, a
// streaming copy inside the block, and 256-byte runs written at random places (the radix scatter's pattern).
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
__global__ void k_runs(const uint4* __restrict__ in, uint4* __restrict__ out, uint64_t n_runs) {      // 16 lanes x 16 B = one 256-byte run per 16 lanes
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, run = t >> 4, l = t & 15;
    if (run >= n_runs) return;
    const uint32_t h = mix((uint32_t)run), h2 = mix(h ^ 0x85EBCA6Bu);
    const uint64_t dst = (uint64_t)(((unsigned __int128)(((uint64_t)h << 32) | h2) * n_runs) >> 64);
    out[dst * 16 + l] = in[run * 16 + l];
}
int main() {
    const size_t bytes = 26ull << 30;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    uint64_t* s[5];
    for (int v = 0; v < 5; v++) if (hipMalloc((void**)&s[v], bytes) != hipSuccess) { s[v] = nullptr; (void)hipGetLastError(); }
    auto timeit = [&](auto fn) { float best = 1e9; for (int rep = 0; rep < 3; rep++) { hipEventRecord(a); fn(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; } return best; };
    for (int v = 0; v < 5; v++) {
        if (!s[v]) continue;
        const uint64_t half16 = bytes / 32;                       // uint4 elements in half the block
        const float t1 = timeit([&] { k_rand<<<256 * 16 * 8, 256>>>(s[v], bytes / 24, 16); });
        const float t2 = timeit([&] { k_copy<<<256 * 32, 256>>>((const uint4*)s[v], (uint4*)s[v] + half16, half16 / 2); });     // 6.5 GB read + 6.5 GB written
        const float t3 = timeit([&] { k_runs<<<(uint32_t)((half16 / 2 + 255) / 256), 256>>>((const uint4*)s[v], (uint4*)s[v] + half16, half16 / 2 / 16); });
        printf("block %d: random 24 B writes %.2f ms | copy 6.5 GB %.2f ms (%.2f TB/s) | 256 B runs scattered %.2f ms\n", v, t1, t2, 13.0 * 1.0737 / t2, t3);
    }
    return 0;
}
