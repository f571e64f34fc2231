// Does the placement kind of a 26 GB slab (slab_probe2.hip: 5.9 or 7.0-7.5 ms for the same random 24-byte writes; counters in
// profiles/r05_slab_modes.txt: the slow kind shows +45 % DRAM write-credit stalls at the L2, same requests, same hit rates) depend
// on HOW the virtual range is backed?  Same probe on: plain hipMalloc blocks; one virtual reservation mapped from 2 GB physical
// handles (hipMemCreate / hipMemMap); from 256 MB handles; from a single 26 GB handle.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
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
static uint64_t* vmm_block(size_t bytes, size_t piece, int dev) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) { printf("granularity query failed\n"); return nullptr; }
    piece = (piece + gran - 1) / gran * gran;
    bytes = (bytes + piece - 1) / piece * piece;
    void* va = nullptr;
    if (hipMemAddressReserve(&va, bytes, 0, nullptr, 0) != hipSuccess) { printf("reserve failed\n"); (void)hipGetLastError(); return nullptr; }
    for (size_t o = 0; o < bytes; o += piece) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, piece, &prop, 0) != hipSuccess) { printf("hipMemCreate failed at %zu\n", o); (void)hipGetLastError(); return nullptr; }
        if (hipMemMap((char*)va + o, piece, 0, h, 0) != hipSuccess) { printf("hipMemMap failed\n"); (void)hipGetLastError(); return nullptr; }
    }
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice; acc.location.id = dev; acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(va, bytes, &acc, 1) != hipSuccess) { printf("hipMemSetAccess failed\n"); (void)hipGetLastError(); return nullptr; }
    return (uint64_t*)va;
}
int main() {
    const size_t bytes = 26ull << 30;
    (void)hipEventCreate(&ea); (void)hipEventCreate(&eb);
    struct B { const char* kind; uint64_t* p; };
    std::vector<B> bs;
    for (int i = 0; i < 2; i++) { uint64_t* p = nullptr; if (hipMalloc((void**)&p, bytes) != hipSuccess) { (void)hipGetLastError(); p = nullptr; } bs.push_back({"hipMalloc 26 GB", p}); }
    for (int i = 0; i < 2; i++) bs.push_back({"VMM, 13 x 2 GB handles", vmm_block(bytes, 2ull << 30, 0)});
    for (int i = 0; i < 2; i++) bs.push_back({"VMM, 104 x 256 MB handles", vmm_block(bytes, 256ull << 20, 0)});
    for (int i = 0; i < 2; i++) bs.push_back({"VMM, one 26 GB handle", vmm_block(bytes, bytes, 0)});
    for (int i = 0; i < 2; i++) { uint64_t* p = nullptr; if (hipMalloc((void**)&p, bytes) != hipSuccess) { (void)hipGetLastError(); p = nullptr; } bs.push_back({"hipMalloc 26 GB (after the others)", p}); }
    for (int round = 0; round < 2; round++)
        for (auto& b : bs) {
            if (!b.p) { printf("%-36s -\n", b.kind); continue; }
            printf("%-36s %p  %.2f ms\n", b.kind, (void*)b.p, probe(b.p, bytes));
        }
    return 0;
}
