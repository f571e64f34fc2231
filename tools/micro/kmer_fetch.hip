// Does a k-mer fetch with conditional second/third word loads ever differ from three unconditional loads?  Run several copies at once.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/kmer_fetch tools/micro/kmer_fetch.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "../../rust-debruijn_amd/csrc/dbg_device.hpp"
// the fetch the library's device code used until round 3 (now host-only in dbg_device.hpp): second and third word loaded only when needed
__device__ __forceinline__ K128 cond_get_kmer(const uint64_t* __restrict__ w, uint64_t o, int k) {
    uint64_t wi = o >> 5;
    int s = (int)(o & 31) * 2;
    int need = s + 2 * k;
    uint64_t w0 = w[wi];
    uint64_t w1 = need > 64 ? w[wi + 1] : 0;
    uint64_t w2 = need > 128 ? w[wi + 2] : 0;
    K128 top;
    if (s) { top.hi = (w0 << s) | (w1 >> (64 - s)); top.lo = (w1 << s) | (w2 >> (64 - s)); }
    else   { top.hi = w0; top.lo = w1; }
    return k128_shr(top, 128 - 2 * k);
}
__device__ __forceinline__ K128 get_kmer3(const uint64_t* __restrict__ w, uint64_t o, int k) {
    const uint64_t wi = o >> 5; const int s = (int)(o & 31) * 2;
    const uint64_t w0 = w[wi], w1 = w[wi + 1], w2 = w[wi + 2];
    K128 top;
    if (s) { top.hi = (w0 << s) | (w1 >> (64 - s)); top.lo = (w1 << s) | (w2 >> (64 - s)); } else { top.hi = w0; top.lo = w1; }
    return k128_shr(top, 128 - 2 * k);
}
__global__ void a_kernel(const uint64_t* __restrict__ words, const uint64_t* __restrict__ start, const uint32_t* __restrict__ length, uint32_t n, int k,
                         uint64_t* f_hi, uint64_t* f_lo, uint64_t* l_hi, uint64_t* l_lo) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    K128 f = cond_get_kmer(words, start[i], k);
    K128 l = cond_get_kmer(words, start[i] + length[i] - (uint32_t)k, k);
    f_hi[i] = f.hi; l_hi[i] = l.hi; f_lo[i] = f.lo; l_lo[i] = l.lo;
}
__global__ void a3_kernel(const uint64_t* __restrict__ words, const uint64_t* __restrict__ start, const uint32_t* __restrict__ length, uint32_t n, int k,
                          uint64_t* f_hi, uint64_t* f_lo, uint64_t* l_hi, uint64_t* l_lo) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    K128 f = get_kmer3(words, start[i], k);
    K128 l = get_kmer3(words, start[i] + length[i] - (uint32_t)k, k);
    f_hi[i] = f.hi; l_hi[i] = l.hi; f_lo[i] = f.lo; l_lo[i] = l.lo;
}
__global__ void b_kernel(const uint64_t* __restrict__ words, const uint64_t* __restrict__ start, const uint32_t* __restrict__ length, uint32_t n, int k,
                         const uint64_t* f_hi, const uint64_t* f_lo, const uint64_t* l_hi, const uint64_t* l_lo, unsigned int* bad, unsigned long long* ex) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    K128 f = get_kmer3(words, start[i], k);
    K128 l = get_kmer3(words, start[i] + length[i] - (uint32_t)k, k);
    if (f_hi[i] != f.hi || f_lo[i] != f.lo || l_hi[i] != l.hi || l_lo[i] != l.lo) {
        const unsigned int q = atomicAdd(bad, 1u);
        if (q < 4) { unsigned long long* e = ex + q * 12; e[0] = i; e[1] = start[i]; e[2] = length[i]; e[3] = (unsigned)k;
                     e[4] = f_hi[i]; e[5] = f_lo[i]; e[6] = f.hi; e[7] = f.lo; e[8] = l_hi[i]; e[9] = l_lo[i]; e[10] = l.hi; e[11] = l.lo; }
    }
}
int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? atoi(argv[1]) : 2000;
    const int iters = argc > 2 ? atoi(argv[2]) : 20000;
    const int mode = argc > 3 ? atoi(argv[3]) : 0;      // 1: both kernels fetch unconditionally; 2: no memset before the copies; 3: sync after the copies
    std::vector<uint64_t> words, start(n); std::vector<uint32_t> len(n);
    uint64_t x = 88172645463325252ull, pos = 0;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (uint32_t i = 0; i < n; i++) { len[i] = 64 + rnd() % 200; start[i] = pos; pos += len[i]; }
    words.resize(pos / 32 + 3);
    for (auto& w : words) w = rnd();
    uint64_t *d_w, *d_s, *fh, *fl, *lh, *ll; uint32_t* d_l; unsigned int* d_bad;
    hipMalloc(&d_w, words.size() * 8); hipMalloc(&d_s, n * 8); hipMalloc(&d_l, n * 4);
    hipMalloc(&fh, n * 8); hipMalloc(&fl, n * 8); hipMalloc(&lh, n * 8); hipMalloc(&ll, n * 8); hipMalloc(&d_bad, 4);
    unsigned long long* d_ex; hipMalloc(&d_ex, 48 * 8);
    hipMemset(d_bad, 0, 4);
    unsigned long long total_bad = 0;
    for (int it = 0; it < iters; it++) {
        const int k = 16 + it % 49;
        // the same sequence of stream operations as dev_graph_build: memset, pageable H2D copies, kernel
        if (mode != 2) hipMemsetAsync(d_w, 0, words.size() * 8, 0);
        hipMemcpyAsync(d_w, words.data(), (words.size() - 2) * 8, hipMemcpyHostToDevice, 0);
        hipMemcpyAsync(d_s, start.data(), n * 8, hipMemcpyHostToDevice, 0);
        hipMemcpyAsync(d_l, len.data(), n * 4, hipMemcpyHostToDevice, 0);
        if (mode == 3) hipStreamSynchronize(0);
        if (mode == 1) a3_kernel<<<(n + 255) / 256, 256>>>(d_w, d_s, d_l, n, k, fh, fl, lh, ll);
        else a_kernel<<<(n + 255) / 256, 256>>>(d_w, d_s, d_l, n, k, fh, fl, lh, ll);
        b_kernel<<<(n + 255) / 256, 256>>>(d_w, d_s, d_l, n, k, fh, fl, lh, ll, d_bad, d_ex);
        if (it % 500 == 499) { unsigned int b; hipMemcpy(&b, d_bad, 4, hipMemcpyDeviceToHost); total_bad += b; if (b) { printf("iter %d: %u mismatching nodes\n", it, b); unsigned long long e[48]; hipMemcpy(e, d_ex, sizeof(e), hipMemcpyDeviceToHost);
                for (unsigned q = 0; q < (b < 4 ? b : 4); q++) { unsigned long long* x = e + q * 12;
                    printf("  node %llu start %llu (s=%llu) len %llu k %llu\n   first got %016llx:%016llx want %016llx:%016llx\n   last  got %016llx:%016llx want %016llx:%016llx (last offset s=%llu)\n",
                           x[0], x[1], (x[1] & 31) * 2, x[2], x[3], x[4], x[5], x[6], x[7], x[8], x[9], x[10], x[11], ((x[1] + x[2] - x[3]) & 31) * 2); } } hipMemset(d_bad, 0, 4); }
    }
    hipDeviceSynchronize();
    printf("mode %d done: %llu mismatches\n", mode, total_bad);
    return 0;
}
