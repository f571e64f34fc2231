// PCIe between pinned host memory and HBM: what the host-boundary calls (dbg_filter_kmers: host arrays in, host table out) can reach.
// One hipMemcpyAsync, two on two streams, both directions at once, and a kernel that writes / reads the pinned block directly.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void copy_kernel(const ulonglong2* __restrict__ src, ulonglong2* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t bytes = 4ull << 30;
    void *h0, *h1, *d0, *d1;
    CK(hipHostMalloc(&h0, bytes, hipHostMallocDefault)); CK(hipHostMalloc(&h1, bytes, hipHostMallocDefault));
    CK(hipMalloc(&d0, bytes)); CK(hipMalloc(&d1, bytes));
    memset(h0, 1, bytes); memset(h1, 2, bytes);
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    auto t = [&](const char* name, double gb, auto fn) {
        double best = 1e9;
        for (int r = 0; r < 3; r++) { hipDeviceSynchronize(); const double a = now(); fn(); hipDeviceSynchronize(); best = std::min(best, now() - a); }
        printf("%-58s %7.1f GB/s\n", name, gb / best);
    };
    const double G = bytes / 1e9;
    t("H2D one copy", G, [&] { hipMemcpyAsync(d0, h0, bytes, hipMemcpyHostToDevice, s0); });
    t("D2H one copy", G, [&] { hipMemcpyAsync(h0, d0, bytes, hipMemcpyDeviceToHost, s0); });
    t("H2D two copies on two streams", 2 * G, [&] { hipMemcpyAsync(d0, h0, bytes, hipMemcpyHostToDevice, s0); hipMemcpyAsync(d1, h1, bytes, hipMemcpyHostToDevice, s1); });
    t("D2H two copies on two streams", 2 * G, [&] { hipMemcpyAsync(h0, d0, bytes, hipMemcpyDeviceToHost, s0); hipMemcpyAsync(h1, d1, bytes, hipMemcpyDeviceToHost, s1); });
    t("H2D + D2H at once (sum of both directions)", 2 * G, [&] { hipMemcpyAsync(d0, h0, bytes, hipMemcpyHostToDevice, s0); hipMemcpyAsync(h1, d1, bytes, hipMemcpyDeviceToHost, s1); });
    t("D2H in 64 MB pieces, one stream", G, [&] { for (size_t o = 0; o < bytes; o += 64u << 20) hipMemcpyAsync((char*)h0 + o, (char*)d0 + o, 64u << 20, hipMemcpyDeviceToHost, s0); });
    t("D2H by a kernel writing the pinned block (1024 x 256)", G, [&] { copy_kernel<<<1024, 256, 0, s0>>>((const ulonglong2*)d0, (ulonglong2*)h0, bytes / 16); });
    t("H2D by a kernel reading the pinned block (1024 x 256)", G, [&] { copy_kernel<<<1024, 256, 0, s0>>>((const ulonglong2*)h0, (ulonglong2*)d0, bytes / 16); });
    t("D2H kernel + D2H copy at once", 2 * G, [&] { copy_kernel<<<1024, 256, 0, s0>>>((const ulonglong2*)d0, (ulonglong2*)h0, bytes / 16); hipMemcpyAsync(h1, d1, bytes, hipMemcpyDeviceToHost, s1); });
    // pageable source staged by host threads into pinned chunks is what dbg_filter_kmers does on the way in: the memcpy rate alone
    std::vector<char> pg(bytes, 3);
    for (int nt : {1, 4, 8, 16}) {
        double best = 1e9;
        for (int r = 0; r < 2; r++) {
            const double a = now();
            std::vector<std::thread> th;
            for (int i = 0; i < nt; i++) th.emplace_back([&, i] { const size_t lo = bytes * i / nt, hi = bytes * (i + 1) / nt; memcpy((char*)h0 + lo, pg.data() + lo, hi - lo); });
            for (auto& x : th) x.join();
            best = std::min(best, now() - a);
        }
        printf("host memcpy pageable -> pinned, %2d threads                   %7.1f GB/s\n", nt, G / best);
    }
    return 0;
}
