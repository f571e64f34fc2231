// Micro-benchmark for the order-restoring sort: what does ONE scatter pass cost when a tile's 4096
// 16-byte records go to 4096 digit streams (a 12-bit digit: every record its own 16-byte run) instead of 256 streams (8-bit
// digit: runs of ~16 records = 256 B)?  Streams are append-only, so consecutive tiles write neighbouring slots of every stream:
// whether the 16-byte pieces of a 128-byte line meet in an L2 before they leave it depends on WHICH XCD the tiles run on
// (per-XCD L2s; block b runs on XCD b % 8 -- observed, for speed only).
//   mode 0: 256 streams, runs of 16 records (today's pass)
//   mode 1: 4096 streams, tile t = blockIdx.x (neighbouring slots written from 8 different XCDs)
//   mode 2: 4096 streams, the output cut into 8 chunks, chunk = blockIdx.x % 8, tile within the chunk = blockIdx.x / 8
//           (every stream segment is written by ONE XCD)
//   mode 3: as 2 with 64 chunks (chunk = blockIdx.x % 64: eight chunks per XCD)
// build: hipcc --offload-arch=gfx950 -O3 -o scatter12 scatter12.hip ; run: ./scatter12 [records]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
__global__ void __launch_bounds__(512) scatter_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t n_tiles, int mode) {
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    uint32_t nchunk = mode == 2 ? 8u : (mode == 3 ? 64u : 1u);
    const uint32_t chunk = b % nchunk, t = b / nchunk, tpc = (n_tiles + nchunk - 1) / nchunk;
    for (int i = 0; i < 8; i++) {
        const uint32_t r = tid + 512u * i;
        const uint4 v = in[(uint64_t)b * 4096 + r];
        uint64_t pos;
        if (mode == 0) {
            // tile-local rank r: stream = r / 16 (256 streams), 16 consecutive records per stream and tile
            const uint32_t d = (r >> 4) ^ ((b * 37u) & 255u);
            pos = ((uint64_t)d * n_tiles + b) * 16 + (r & 15u);
        } else {
            const uint32_t d = (r * 2654435761u + b * 40503u) >> 20;            // 4096 streams, one record per stream and tile (a permutation of r)
            pos = ((uint64_t)(chunk * 4096u + (d ^ (r & 0u))) * tpc + t);
        }
        out[pos] = v;
    }
}
int main(int argc, char** argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 500000000ull;
    const uint32_t n_tiles = (uint32_t)(n / 4096);
    uint4 *in, *out;
    hipMalloc(&in, (uint64_t)n_tiles * 4096 * 16);
    hipMalloc(&out, ((uint64_t)n_tiles + 64) * 4096 * 16 + (1 << 20));
    hipMemset(in, 1, (uint64_t)n_tiles * 4096 * 16);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 4; mode++) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a);
            scatter_kernel<<<n_tiles, 512>>>(in, out, n_tiles, mode);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("mode %d: %.2f ms for %.1f GB read + %.1f GB written (%.2f TB/s)\n", mode, best, n_tiles * 65536.0 / 1e9, n_tiles * 65536.0 / 1e9,
               2 * n_tiles * 65536.0 / best / 1e9);
    }
    return 0;
}
