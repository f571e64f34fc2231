// micro-benchmark: random 24-byte record writes (+ one returning atomic per record on a 1.1e6-entry cursor array) into a
// 26 GB "slab" buffer -- the memory behaviour of the super-k-mer scan -- as a function of HOW the buffer was allocated
// (plain hipMalloc / fine-grained / physically contiguous) and of WHICH 2 GB window of it is written.
//   hipcc --offload-arch=gfx950 -O3 slab_probe.hip -o slab_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16; return x; }
__global__ void k(uint64_t* slab, uint64_t rec_lo, uint64_t n_rec, uint32_t* cur, uint32_t nb, uint32_t per_thread, int with_atomic) {
    uint32_t h = mix(blockIdx.x * 1024u + threadIdx.x), acc = 0;
    for (uint32_t i = 0; i < per_thread; i++) {
        h = mix(h + i);
        const uint32_t h2 = mix(h ^ 0x9E3779B9u);
        if (with_atomic) acc += atomicAdd(&cur[(uint32_t)(((uint64_t)h2 * nb) >> 32)], 1u);
        const uint64_t r = rec_lo + (uint64_t)(((unsigned __int128)(((uint64_t)h << 32) | h2) * n_rec) >> 64);
        uint64_t* o = slab + r * 3;
        o[0] = h; o[1] = h2 + acc; o[2] = i;
    }
}
int main(int argc, char** argv) {
    const size_t bytes = 26ull << 30;
    const uint64_t n_rec = bytes / 24;
    const uint32_t nb = 1100000, per_thread = 16;
    const uint32_t blocks = 256 * 16 * 8, threads = 256;          // 8.4M threads x 16 = 1.3e8 records per launch
    uint32_t* cur; hipMalloc(&cur, (size_t)nb * 4); hipMemset(cur, 0, (size_t)nb * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const char* names[] = {"plain", "finegrained", "contiguous"};
    const unsigned flags[] = {0, hipDeviceMallocFinegrained, hipDeviceMallocContiguous};
    for (int v = 0; v < 3; v++) {
        uint64_t* slab = nullptr;
        hipError_t e = flags[v] ? hipExtMallocWithFlags((void**)&slab, bytes, flags[v]) : hipMalloc((void**)&slab, bytes);
        if (e != hipSuccess) { printf("%s: allocation failed\n", names[v]); (void)hipGetLastError(); continue; }
        auto run = [&](uint64_t lo, uint64_t n, int at) {
            float best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(a);
                k<<<blocks, threads>>>(slab, lo, n, cur, nb, per_thread, at);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            return best;
        };
        printf("%-12s whole 26 GB: %.2f ms with atomics, %.2f ms writes only (1.34e8 records)\n", names[v], run(0, n_rec, 1), run(0, n_rec, 0));
        printf("%-12s 2 GB windows (writes only, ms):", names[v]);
        const uint64_t wrec = (2ull << 30) / 24;
        for (int w = 0; w < 13; w++) printf(" %.2f", run((uint64_t)w * wrec, wrec, 0));
        printf("\n");
        hipFree(slab);
    }
    return 0;
}
