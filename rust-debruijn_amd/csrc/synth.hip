// Deterministic synthetic reads, counter-based so that any read can be
// generated independently on the device or on the host with bit-identical results:
//   genome word w (32 bases)   = splitmix64 output #w of a stream seeded with genome_seed
//   read r: x0, x1             = outputs #0, #1 of a stream seeded with read_seed ^ (r+1)*GOLDEN
//       start  = x0 % (G - L + 1);  reverse strand iff !stranded && (x1 >> 63)
//       base i : u = 32-bit half (i&1) of output #(2 + i/2);
//                substitution iff (u >> 8) < floor(e * 2^24); replacement = (orig + 1 + (u & 0xff) % 3) & 3
//   colour D1 = r % n_colours (u8)
// Reads are packed back-to-back: start[r - first_read] = (r - first_read) * L, PackedDnaStringSet layout.
#include "dbg_internal.hpp"
#include <algorithm>
#include <cmath>

#define GOLDEN 0x9E3779B97F4A7C15ull

__host__ __device__ __forceinline__ uint64_t sm64_mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// i-th output (i = 0, 1, ...) of splitmix64 seeded with `seed`
__host__ __device__ __forceinline__ uint64_t sm64_at(uint64_t seed, uint64_t i) { return sm64_mix(seed + (i + 1) * GOLDEN); }

struct SynthP {
    uint64_t n_reads, genome_len, genome_seed, read_seed, first_read;
    uint32_t read_len, thr24, n_colours;
    int stranded;
};

__host__ __device__ __forceinline__ uint32_t synth_base(const SynthP& p, uint64_t r_local, uint32_t i) {
    uint64_t r = p.first_read + r_local;
    uint64_t rs = p.read_seed ^ ((r + 1) * GOLDEN);
    uint64_t x0 = sm64_at(rs, 0), x1 = sm64_at(rs, 1);
    uint64_t st = x0 % (p.genome_len - p.read_len + 1);
    bool rev = !p.stranded && (x1 >> 63);
    uint64_t g = rev ? st + (p.read_len - 1 - i) : st + i;
    uint64_t gw = sm64_at(p.genome_seed, g >> 5);
    uint32_t b = (uint32_t)((gw >> (62 - 2 * (g & 31))) & 3ull);
    if (rev) b = 3u - b;
    if (p.thr24) {
        uint64_t d = sm64_at(rs, 2 + (i >> 1));
        uint32_t u = (uint32_t)(d >> (32 * (i & 1)));
        if ((u >> 8) < p.thr24) b = (b + 1 + (u & 0xffu) % 3u) & 3u;
    }
    return b;
}

__host__ __device__ __forceinline__ uint64_t synth_word(const SynthP& p, uint64_t w) {
    uint64_t total = p.n_reads * p.read_len;
    uint64_t out = 0;
    uint64_t b0 = w * 32;
    for (int j = 0; j < 32; j++) {
        uint64_t b = b0 + j;
        if (b >= total) break;
        uint64_t r = b / p.read_len;
        uint32_t i = (uint32_t)(b - r * p.read_len);
        out |= (uint64_t)synth_base(p, r, i) << (62 - 2 * j);
    }
    return out;
}

namespace {
__global__ void synth_words_kernel(SynthP p, uint64_t n_words, uint64_t* __restrict__ words) {
    uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < n_words) words[w] = synth_word(p, w);
}
__global__ void synth_meta_kernel(SynthP p, uint64_t* __restrict__ start, uint32_t* __restrict__ length,
                                  uint8_t* __restrict__ data) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= p.n_reads) return;
    start[r] = r * p.read_len;
    length[r] = p.read_len;
    if (data) data[r] = p.n_colours ? (uint8_t)((p.first_read + r) % p.n_colours) : 0;
}

SynthP make_p(const dbg_synth_params* q) {
    SynthP p;
    p.n_reads = q->n_reads; p.read_len = q->read_len;
    p.genome_len = q->genome_len ? q->genome_len : std::max<uint64_t>(q->n_reads * q->read_len / 30, q->read_len);
    p.genome_seed = q->genome_seed; p.read_seed = q->read_seed; p.first_read = q->first_read;
    double t = q->error_rate * 16777216.0;
    p.thr24 = t <= 0 ? 0u : (t >= 16777215.0 ? 16777215u : (uint32_t)std::floor(t));
    p.n_colours = q->n_colours; p.stranded = q->stranded;
    return p;
}
}  // namespace

extern "C" uint64_t dbg_synth_words(const dbg_synth_params* q) {
    return (q->n_reads * q->read_len + 31) / 32 + 2;
}

int synth_reads_dev(dbg_ctx* ctx, const dbg_synth_params* q, uint64_t* words, uint64_t* start, uint32_t* length,
                    uint8_t* data) {
    SynthP p = make_p(q);
    if (p.genome_len < p.read_len) return ctx->fail(120, "synthetic genome shorter than a read");
    uint64_t nw = dbg_synth_words(q);
    synth_words_kernel<<<cdiv(nw, 256), 256, 0, ctx->stream>>>(p, nw, words);
    LAUNCH_CHECK(ctx, "synth_words");
    if (p.n_reads) {
        synth_meta_kernel<<<cdiv(p.n_reads, 256), 256, 0, ctx->stream>>>(p, start, length, data);
        LAUNCH_CHECK(ctx, "synth_meta");
    }
    return 0;
}

extern "C" int dbg_synth_reads_host(const dbg_synth_params* q, uint64_t* words, uint64_t* start, uint32_t* length,
                                    uint8_t* data) {
    SynthP p = make_p(q);
    if (p.genome_len < p.read_len) return 120;
    uint64_t nw = dbg_synth_words(q);
    for (uint64_t w = 0; w < nw; w++) words[w] = synth_word(p, w);
    for (uint64_t r = 0; r < p.n_reads; r++) {
        start[r] = r * p.read_len;
        length[r] = p.read_len;
        if (data) data[r] = p.n_colours ? (uint8_t)((p.first_read + r) % p.n_colours) : 0;
    }
    return 0;
}
