// Round 6: slab_probe5 said "a range mapped from 256 MB handles is always of the fast kind" -- measured in a clean process.  Inside the
// bench process (torch's input tensors and the library's other buffers allocated first) the same mapping probes at 5.4 .. 7.2 ms and the
// scan follows (30.8 .. 33.7 ms, per block).  This probe repeats the measurement with allocations like the bench's made FIRST, over more
// handle sizes, with the virtual range aligned to the handle size or not, three blocks per kind, probed in two rounds.
// usage: slab_probe6 [pre_alloc 0|1] [blocks per kind]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <string>
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
static hipEvent_t ea, eb;
static float probe(uint64_t* p, size_t bytes) {
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(ea);
        k<<<256 * 16 * 8, 256>>>(p, bytes / 24, 16);
        (void)hipEventRecord(eb); (void)hipEventSynchronize(eb);
        float ms; (void)hipEventElapsedTime(&ms, ea, eb);
        if (ms < best) best = ms;
    }
    return best;
}
#include <chrono>
static double g_last_map_s = 0;
static uint64_t* vmm_block_(size_t bytes, size_t piece, size_t va_align);
static uint64_t* vmm_block(size_t bytes, size_t piece, size_t va_align) {
    auto t0 = std::chrono::steady_clock::now();
    uint64_t* p = vmm_block_(bytes, piece, va_align);
    g_last_map_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return p;
}
static uint64_t* vmm_block_(size_t bytes, size_t piece, size_t va_align) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) return nullptr;
    piece = (piece + gran - 1) / gran * gran;
    bytes = (bytes + piece - 1) / piece * piece;
    void* va = nullptr;
    if (hipMemAddressReserve(&va, bytes, va_align, nullptr, 0) != hipSuccess) { printf("reserve failed (align %zu)\n", va_align); (void)hipGetLastError(); return nullptr; }
    for (size_t o = 0; o < bytes; o += piece) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, piece, &prop, 0) != hipSuccess) { printf("hipMemCreate failed at %zu\n", o); (void)hipGetLastError(); return nullptr; }
        if (hipMemMap((char*)va + o, piece, 0, h, 0) != hipSuccess) { printf("hipMemMap failed\n"); (void)hipGetLastError(); return nullptr; }
    }
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(va, bytes, &acc, 1) != hipSuccess) { printf("hipMemSetAccess failed\n"); (void)hipGetLastError(); return nullptr; }
    return (uint64_t*)va;
}
int main(int argc, char** argv) {
    const int pre = argc > 1 ? atoi(argv[1]) : 1, nb = argc > 2 ? atoi(argv[2]) : 3;
    const size_t bytes = 99ull * (256ull << 20);
    (void)hipEventCreate(&ea); (void)hipEventCreate(&eb);
    if (pre) {                                   // what the bench process holds before the first slab is mapped
        const size_t pre_sizes[] = {3750000016ull, 800000000ull, 400000000ull, 100000000ull, 2ull << 20, 4400000ull, 4400000ull, 8, 8, 1200000000ull, 300000000ull};
        for (size_t s : pre_sizes) { void* p = nullptr; if (hipMalloc(&p, s) != hipSuccess) printf("pre-allocation of %zu failed\n", s); }
    }
    size_t gran = 0;
    { hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
      (void)hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended); size_t gmin = 0;
      (void)hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum); printf("granularity: recommended %zu, minimum %zu\n", gran, gmin); }
    struct B { std::string kind; uint64_t* p; };
    std::vector<B> bs;
    struct K { const char* name; size_t piece; size_t align; };
    const int set = argc > 3 ? atoi(argv[3]) : 0;
    const K kinds_a[] = {{"VMM 2 MB handles", 2ull << 20, 0}, {"VMM 32 MB handles", 32ull << 20, 0}, {"VMM 256 MB handles", 256ull << 20, 0},
                         {"VMM 256 MB handles, VA aligned 256 MB", 256ull << 20, 256ull << 20}, {"VMM 1 GB handles", 1ull << 30, 0},
                         {"VMM 1 GB handles, VA aligned 1 GB", 1ull << 30, 1ull << 30}, {"VMM 2 GB handles, VA aligned 2 GB", 2ull << 30, 2ull << 30}};
    const K kinds_b[] = {{"VMM 2 MB handles", 2ull << 20, 0}, {"VMM 4 MB handles", 4ull << 20, 0}, {"VMM 8 MB handles", 8ull << 20, 0}};
    std::vector<K> kinds;
    if (set == 0) kinds.assign(kinds_a, kinds_a + 7); else kinds.assign(kinds_b, kinds_b + 3);
    for (int i = 0; i < nb; i++) {               // kinds interleaved, as the blocks of a long-lived process would be
        for (const K& kd : kinds) { bs.push_back({kd.name, vmm_block(bytes, kd.piece, kd.align)}); printf("mapped %-40s in %.3f s\n", kd.name, g_last_map_s); }
        if (set == 0) { uint64_t* p = nullptr; if (hipMalloc((void**)&p, bytes) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
        bs.push_back({"hipMalloc", p}); }
    }
    for (int round = 0; round < 2; round++)
        for (auto& b : bs) {
            if (!b.p) { printf("%-40s -\n", b.kind.c_str()); continue; }
            printf("%-40s %p  %.2f ms\n", b.kind.c_str(), (void*)b.p, probe(b.p, bytes));
        }
    return 0;
}
