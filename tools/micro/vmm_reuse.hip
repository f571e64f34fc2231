// Round 6: is it safe to release a piecewise-mapped block (hipMemUnmap / hipMemRelease / hipMemAddressFree) right after kernels wrote it
// at random, and map a fresh block of the same size next?  The slab tournament did exactly that, and one bench process in four then failed
// with "corrupt super-k-mer record".  Here: block A takes 2^27 random 24-byte writes, is released; block B is mapped (it usually gets A's
// virtual range back), every word of B is written with f(index) by one kernel and verified by the next.  Modes: 0 = release immediately;
// 1 = hipDeviceSynchronize before the release; 2 = keep A's virtual range reserved (no hipMemAddressFree); 3 = never release (control);
// 4 = release immediately, then sleep 300 ms after mapping B before the first kernel touches it; 5 = release immediately, B's fill + check
// repeated a second time 300 ms later (does a late writer undo the fill?).  The check reports how many mismatching words are ZERO.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <unistd.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16; return x; }
__global__ void rnd_write(uint64_t* slab, uint64_t n_rec, uint32_t per_thread, uint32_t salt) {
    uint32_t h = mix(blockIdx.x * 1024u + threadIdx.x + salt);
    for (uint32_t i = 0; i < per_thread; i++) {
        h = mix(h + i);
        const uint32_t h2 = mix(h ^ 0x9E3779B9u);
        const uint64_t r = (uint64_t)(((unsigned __int128)(((uint64_t)h << 32) | h2) * n_rec) >> 64);
        uint64_t* o = slab + r * 3;
        o[0] = h; o[1] = h2; o[2] = i;
    }
}
__global__ void fill(uint64_t* p, uint64_t n, uint64_t salt) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = i * 0x9E3779B97F4A7C15ull + salt;
}
__global__ void check(const uint64_t* p, uint64_t n, uint64_t salt, unsigned long long* bad) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        if (p[i] != i * 0x9E3779B97F4A7C15ull + salt) { atomicAdd(bad, 1ull); if (p[i] == 0) atomicAdd(bad + 1, 1ull); }
}
struct Blk { void* va; size_t bytes; std::vector<hipMemGenericAllocationHandle_t> hs; size_t piece; };
static Blk map_block(size_t bytes, size_t piece) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    Blk b{nullptr, bytes, {}, piece};
    CK(hipMemAddressReserve(&b.va, bytes, 0, nullptr, 0));
    for (size_t o = 0; o < bytes; o += piece) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, piece, &prop, 0));
        CK(hipMemMap((char*)b.va + o, piece, 0, h, 0));
        b.hs.push_back(h);
    }
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(b.va, bytes, &acc, 1));
    return b;
}
static void release(Blk& b, bool keep_va) {
    size_t o = 0;
    for (auto h : b.hs) { CK(hipMemUnmap((char*)b.va + o, b.piece)); CK(hipMemRelease(h)); o += b.piece; }
    if (!keep_va) CK(hipMemAddressFree(b.va, b.bytes));
}
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const int rounds = argc > 2 ? atoi(argv[2]) : 8;
    const size_t piece = 256ull << 20, bytes = 99 * piece;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned long long* bad; CK(hipMalloc((void**)&bad, 16)); CK(hipMemset(bad, 0, 16));
    Blk a = map_block(bytes, piece);
    unsigned long long total_bad = 0;
    for (int r = 0; r < rounds; r++) {
        rnd_write<<<256 * 16 * 8, 256, 0, s>>>((uint64_t*)a.va, bytes / 24, 16, r);
        CK(hipStreamSynchronize(s));
        Blk keep = a;
        if (mode == 1) CK(hipDeviceSynchronize());
        if (mode != 3) release(a, mode == 2);
        Blk b = map_block(bytes, piece);
        if (mode == 4) usleep(300000);
        fill<<<4096, 256, 0, s>>>((uint64_t*)b.va, bytes / 8, r);
        rnd_write<<<256 * 16, 256, 0, s>>>((uint64_t*)b.va, bytes / 24, 1, 77);       // (some scattered writes afterwards, like the scan's)
        fill<<<4096, 256, 0, s>>>((uint64_t*)b.va, bytes / 8, r);
        check<<<4096, 256, 0, s>>>((uint64_t*)b.va, bytes / 8, r, bad);
        CK(hipStreamSynchronize(s));
        unsigned long long hb[2] = {0, 0}; CK(hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost));
        printf("mode %d round %d: A %p -> B %p%s, mismatching words so far %llu (zero: %llu)\n", mode, r, keep.va, b.va, b.va == keep.va ? " (same range)" : "", hb[0], hb[1]);
        if (mode == 5) {
            usleep(300000);
            check<<<4096, 256, 0, s>>>((uint64_t*)b.va, bytes / 8, r, bad);
            CK(hipStreamSynchronize(s));
            unsigned long long h2[2]; CK(hipMemcpy(h2, bad, 16, hipMemcpyDeviceToHost));
            printf("   re-checked 300 ms later: +%llu mismatching words (zero: +%llu)\n", h2[0] - hb[0], h2[1] - hb[1]);
            hb[0] = h2[0];
        }
        total_bad = hb[0];
        if (mode == 3) release(keep, false);
        a = b;
    }
    printf("mode %d: %s\n", mode, total_bad ? "CORRUPTION" : "clean");
    return total_bad ? 2 : 0;
}
