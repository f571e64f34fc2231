// Round 6: where does a block's placement kind come from?  One 99 x 256 MB block is mapped; the random-write probe then runs (a) over the
// whole block, (b) over each 256 MB handle alone (same number of writes confined to that handle), (c) over the first / second half, the even /
// odd handles.  If the handles differ from one another, a slab could be assembled from the good ones.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16; return x; }
// writes land in pieces [first, first + count) step `step` of the block
__global__ void k(uint64_t* slab, uint64_t rec_per_piece, uint32_t first, uint32_t count, uint32_t step, uint32_t per_thread) {
    uint32_t h = mix(blockIdx.x * 1024u + threadIdx.x);
    for (uint32_t i = 0; i < per_thread; i++) {
        h = mix(h + i);
        const uint32_t h2 = mix(h ^ 0x9E3779B9u);
        const uint32_t piece = first + step * (uint32_t)(((uint64_t)h * count) >> 32);
        const uint64_t r = (uint64_t)piece * rec_per_piece + (uint64_t)(((uint64_t)h2 * rec_per_piece) >> 32);
        uint64_t* o = slab + r * 3;
        o[0] = h; o[1] = h2; o[2] = i;
    }
}
static hipEvent_t ea, eb;
static float probe(uint64_t* p, uint64_t rpp, uint32_t first, uint32_t count, uint32_t step, uint32_t blocks) {
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(ea);
        k<<<blocks, 256>>>(p, rpp, first, count, step, 16);
        (void)hipEventRecord(eb); (void)hipEventSynchronize(eb);
        float ms; (void)hipEventElapsedTime(&ms, ea, eb);
        if (ms < best) best = ms;
    }
    return best;
}
int main(int argc, char** argv) {
    const size_t piece = (argc > 1 ? (size_t)atoi(argv[1]) : 256) << 20;
    const uint32_t np = (uint32_t)((99ull * (256ull << 20)) / piece);
    (void)hipEventCreate(&ea); (void)hipEventCreate(&eb);
    const size_t pre_sizes[] = {3750000016ull, 800000000ull, 400000000ull, 100000000ull, 2ull << 20, 4400000ull, 4400000ull, 8, 8, 1200000000ull, 300000000ull};
    for (size_t s : pre_sizes) { void* p = nullptr; (void)hipMalloc(&p, s); }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    for (int blk = 0; blk < 3; blk++) {
        void* va = nullptr;
        if (hipMemAddressReserve(&va, piece * np, 0, nullptr, 0) != hipSuccess) return 1;
        for (uint32_t i = 0; i < np; i++) {
            hipMemGenericAllocationHandle_t h;
            if (hipMemCreate(&h, piece, &prop, 0) != hipSuccess || hipMemMap((char*)va + (size_t)i * piece, piece, 0, h, 0) != hipSuccess) { printf("map failed\n"); return 1; }
        }
        hipMemAccessDesc acc = {};
        acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
        if (hipMemSetAccess(va, piece * np, &acc, 1) != hipSuccess) return 1;
        uint64_t* p = (uint64_t*)va;
        const uint64_t rpp = piece / 24;
        const uint32_t full = 256 * 16 * 8;
        printf("block %d (%u handles of %zu MB): whole %.2f ms | first half %.2f, second half %.2f | even %.2f, odd %.2f | thirds %.2f %.2f %.2f\n", blk, np, piece >> 20,
               probe(p, rpp, 0, np, 1, full), probe(p, rpp, 0, np / 2, 1, full), probe(p, rpp, np / 2, np - np / 2, 1, full),
               probe(p, rpp, 0, (np + 1) / 2, 2, full), probe(p, rpp, 1, np / 2, 2, full),
               probe(p, rpp, 0, np / 3, 1, full), probe(p, rpp, np / 3, np / 3, 1, full), probe(p, rpp, 2 * (np / 3), np / 3, 1, full));
        if (np <= 128) {
            std::vector<float> t(np);
            for (uint32_t i = 0; i < np; i++) t[i] = probe(p, rpp, i, 1, 1, full / 8);       // 2^24 writes into one handle
            printf("  per handle (2^24 writes each), ms:");
            for (uint32_t i = 0; i < np; i++) printf(" %.3f", t[i]);
            std::vector<float> s = t; std::sort(s.begin(), s.end());
            printf("\n  min %.3f  median %.3f  max %.3f\n", s[0], s[np / 2], s[np - 1]);
        }
    }
    return 0;
}
