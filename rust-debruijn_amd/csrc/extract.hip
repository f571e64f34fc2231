// K-mer extraction from packed sequences: the fused GPU form of
//   Vmer::iter_kmer_exts / KmerExtsIter::next      (lib.rs:408-422, :812-841)
//   Kmer::min_rc_flip + Exts::rc when !stranded    (filter.rs:190-196, lib.rs:224-231, :729-748)
// One wavefront walks one sequence 64 k-mers at a time; each lane materialises its k-mer from
// <= 3 consecutive packed words (dna_string.rs:123-153) and writes one (key, payload) record.
#include "dbg_internal.hpp"
#include <algorithm>

namespace {

__global__ void kmer_counts_kernel(const uint32_t* __restrict__ length, uint64_t n, int k, uint32_t* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        uint32_t len = length[i];
        out[i] = len >= (uint32_t)(k - 1) ? len - (uint32_t)(k - 1) : 0u;      // saturating_sub (filter.rs:154)
    }
}

__device__ __forceinline__ uint32_t load_d1(const void* data, uint32_t width, uint64_t i) {
    if (!data) return 0;
    if (width == 1) return ((const uint8_t*)data)[i];
    if (width == 2) return ((const uint16_t*)data)[i];
    return ((const uint32_t*)data)[i];
}

template <bool STRANDED, bool HAS_HI>
__global__ void __launch_bounds__(256) extract_kernel(SeqDev s, const uint64_t* __restrict__ koff, int k,
                                                      uint64_t* __restrict__ out_hi, uint64_t* __restrict__ out_lo,
                                                      uint32_t* __restrict__ out_pay) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t si = wave; si < s.n; si += n_waves) {
        const uint32_t len = s.length[si];
        if (len < (uint32_t)k) continue;                                       // lib.rs:813 with pos = k
        const uint64_t st = s.start[si];
        const uint32_t nk = len - (uint32_t)k + 1;
        const uint32_t sexts = s.exts ? s.exts[si] : 0u;
        const uint32_t d1 = load_d1(s.data, s.data_width, si);
        const uint64_t obase = koff[si];
        for (uint32_t j = lane; j < nk; j += 64) {
            K128 km = packed_get_kmer(s.words, st + j, k, s.n_words ? s.n_words - 1 : 0);
            // lib.rs:820-832: interior exts from the neighbouring bases, boundary exts from seq_exts
            uint32_t left = j == 0 ? (sexts & 0x0fu) : (1u << packed_get(s.words, st + j - 1));
            uint32_t right = (j + (uint32_t)k == len) ? (sexts & 0xf0u) : (16u << packed_get(s.words, st + j + k));
            uint32_t ex = left | right;
            if (!STRANDED) {
                K128 rc = kmer_rc(km, k);
                if (!k128_lt(km, rc)) { km = rc; ex = exts_rc(ex); }             // ties flip (lib.rs:226-230)
            }
            uint64_t o = obase + j;
            if (HAS_HI) out_hi[o] = km.hi;
            out_lo[o] = km.lo;
            out_pay[o] = pay_make(ex, d1);
        }
    }
}
// ---- passes over key ranges (the device analogue of filter.rs:156-168: the reference re-scans the input once per range of
//      its 256 first-four-bases buckets when the k-mers do not fit `memory_size`; here a pass takes a range of the canonical
//      k-mer's top byte and the device decides the ranges from free memory) ----
__device__ __forceinline__ uint32_t key_top_byte(K128 km, int k) {
    const int bits = 2 * k;
    return bits >= 8 ? (uint32_t)k128_shr(km, bits - 8).lo & 0xffu : (uint32_t)(km.lo << (8 - bits)) & 0xffu;
}
template <bool STRANDED>
__device__ __forceinline__ K128 canon_kmer(const uint64_t* __restrict__ w, uint64_t last_word, uint64_t o, int k, bool* flipped) {
    K128 km = packed_get_kmer(w, o, k, last_word);
    *flipped = false;
    if (!STRANDED) {
        const K128 rc = kmer_rc(km, k);
        if (!k128_lt(km, rc)) { km = rc; *flipped = true; }
    }
    return km;
}

// histogram of the canonical k-mers' top bytes over all sequences
template <bool STRANDED>
__global__ void __launch_bounds__(256) top_byte_hist_kernel(SeqDev s, int k, unsigned long long* __restrict__ hist) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t si = wave; si < s.n; si += n_waves) {
        const uint32_t len = s.length[si];
        if (len < (uint32_t)k) continue;
        const uint64_t st = s.start[si];
        const uint32_t nk = len - (uint32_t)k + 1;
        for (uint32_t j = lane; j < nk; j += 64) { bool f; atomicAdd(&h[key_top_byte(canon_kmer<STRANDED>(s.words, s.n_words ? s.n_words - 1 : 0, st + j, k, &f), k)], 1u); }
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// k-mers of every sequence whose top byte lies in [b_lo, b_hi)
template <bool STRANDED>
__global__ void __launch_bounds__(256) kmer_counts_range_kernel(SeqDev s, int k, uint32_t b_lo, uint32_t b_hi, uint32_t* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t si = wave; si < s.n; si += n_waves) {
        const uint32_t len = s.length[si];
        uint32_t cnt = 0;
        if (len >= (uint32_t)k) {
            const uint64_t st = s.start[si];
            const uint32_t nk = len - (uint32_t)k + 1;
            for (uint32_t j0 = 0; j0 < nk; j0 += 64) {
                const uint32_t j = j0 + lane;
                bool in = false;
                if (j < nk) { bool f; const uint32_t b = key_top_byte(canon_kmer<STRANDED>(s.words, s.n_words ? s.n_words - 1 : 0, st + j, k, &f), k); in = b >= b_lo && b < b_hi; }
                cnt += (uint32_t)__popcll(__ballot(in));
            }
        }
        if (lane == 0) out[si] = cnt;
    }
}

template <bool STRANDED, bool HAS_HI>
__global__ void __launch_bounds__(256) extract_range_kernel(SeqDev s, const uint64_t* __restrict__ koff, int k, uint32_t b_lo, uint32_t b_hi,
                                                            uint64_t* __restrict__ out_hi, uint64_t* __restrict__ out_lo, uint32_t* __restrict__ out_pay) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t lt = lanemask_lt();
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t si = wave; si < s.n; si += n_waves) {
        const uint32_t len = s.length[si];
        if (len < (uint32_t)k) continue;
        const uint64_t st = s.start[si];
        const uint32_t nk = len - (uint32_t)k + 1;
        const uint32_t sexts = s.exts ? s.exts[si] : 0u;
        const uint32_t d1 = load_d1(s.data, s.data_width, si);
        uint64_t o = koff[si];                                                  // k-mers keep their order within the sequence
        for (uint32_t j0 = 0; j0 < nk; j0 += 64) {
            const uint32_t j = j0 + lane;
            bool in = false, flipped = false;
            K128 km{0, 0};
            if (j < nk) { km = canon_kmer<STRANDED>(s.words, s.n_words ? s.n_words - 1 : 0, st + j, k, &flipped); const uint32_t b = key_top_byte(km, k); in = b >= b_lo && b < b_hi; }
            const uint64_t m = __ballot(in);
            if (in) {
                uint32_t left = j == 0 ? (sexts & 0x0fu) : (1u << packed_get(s.words, st + j - 1));
                uint32_t right = (j + (uint32_t)k == len) ? (sexts & 0xf0u) : (16u << packed_get(s.words, st + j + k));
                uint32_t ex = left | right;
                if (flipped) ex = exts_rc(ex);
                const uint64_t q = o + (uint32_t)__popcll(m & lt);
                if (HAS_HI) out_hi[q] = km.hi;
                out_lo[q] = km.lo;
                out_pay[q] = pay_make(ex, d1);
            }
            o += (uint32_t)__popcll(m);
        }
    }
}
}  // namespace

int kmer_top_byte_hist(dbg_ctx* ctx, const SeqDev& s, int k, bool stranded, unsigned long long* hist_dev /* [256], zeroed here */) {
    HIP_TRY(ctx, hipMemsetAsync(hist_dev, 0, 256 * 8, ctx->stream));
    if (s.n == 0) return 0;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((s.n + 3) / 4, 256ull * 16);
    if (stranded) top_byte_hist_kernel<true><<<blocks, 256, 0, ctx->stream>>>(s, k, hist_dev);
    else top_byte_hist_kernel<false><<<blocks, 256, 0, ctx->stream>>>(s, k, hist_dev);
    LAUNCH_CHECK(ctx, "top_byte_hist");
    return 0;
}

int kmer_counts_range(dbg_ctx* ctx, const SeqDev& s, int k, bool stranded, uint32_t b_lo, uint32_t b_hi, uint32_t* kcount) {
    if (s.n == 0) return 0;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((s.n + 3) / 4, 256ull * 16);
    if (stranded) kmer_counts_range_kernel<true><<<blocks, 256, 0, ctx->stream>>>(s, k, b_lo, b_hi, kcount);
    else kmer_counts_range_kernel<false><<<blocks, 256, 0, ctx->stream>>>(s, k, b_lo, b_hi, kcount);
    LAUNCH_CHECK(ctx, "kmer_counts_range");
    return 0;
}

int extract_kmers_range(dbg_ctx* ctx, const SeqDev& s, const uint64_t* koff, uint64_t n_kmers, int k, bool stranded, uint32_t b_lo, uint32_t b_hi,
                        RecArrays out) {
    if (s.n == 0 || n_kmers == 0) return 0;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((s.n + 3) / 4, 256ull * 16);
    ctx->t_begin("extract_kmers", n_kmers);
    const bool has_hi = out.hi != nullptr;
#define GO(ST, HH) extract_range_kernel<ST, HH><<<blocks, 256, 0, ctx->stream>>>(s, koff, k, b_lo, b_hi, out.hi, out.lo, out.pay)
    if (stranded) { if (has_hi) GO(true, true); else GO(true, false); }
    else          { if (has_hi) GO(false, true); else GO(false, false); }
#undef GO
    ctx->t_end();
    LAUNCH_CHECK(ctx, "extract_kmers_range");
    return 0;
}

__global__ void __launch_bounds__(256) kmer_total_kernel(const uint32_t* __restrict__ length, uint64_t n, int k, unsigned long long* __restrict__ out) {
    unsigned long long v = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t len = length[i];
        v += len >= (uint32_t)(k - 1) ? len - (uint32_t)(k - 1) : 0u;          // saturating_sub (filter.rs:154)
    }
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(out, v);
}

// input_kmers (filter.rs:152-155) alone: one pass over the lengths
int kmer_total(dbg_ctx* ctx, const SeqDev& s, int k, uint64_t* n_out) {
    *n_out = 0;
    if (s.n == 0) return 0;
    DBuf<unsigned long long> d;
    ALLOC_OR_FAIL(ctx, d, 1);
    HIP_TRY(ctx, hipMemsetAsync(d.p, 0, 8, ctx->stream));
    kmer_total_kernel<<<(uint32_t)std::min<uint64_t>(cdiv(s.n, 256), 2048), 256, 0, ctx->stream>>>(s.length, s.n, k, d.p);
    LAUNCH_CHECK(ctx, "kmer_total");
    unsigned long long h = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&h, d.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = h;
    return 0;
}

int kmer_counts(dbg_ctx* ctx, const SeqDev& s, int k, uint32_t* kcount) {
    if (s.n == 0) return 0;
    kmer_counts_kernel<<<cdiv(s.n, 256), 256, 0, ctx->stream>>>(s.length, s.n, k, kcount);
    LAUNCH_CHECK(ctx, "kmer_counts");
    return 0;
}

int extract_kmers(dbg_ctx* ctx, const SeqDev& s, const uint64_t* koff, uint64_t n_kmers, int k, bool stranded,
                  RecArrays out) {
    if (s.n == 0 || n_kmers == 0) return 0;
    uint64_t waves_needed = s.n;
    uint32_t blocks = (uint32_t)std::min<uint64_t>((waves_needed + 3) / 4, 256ull * 16);
    ctx->t_begin("extract_kmers", n_kmers);
    bool has_hi = out.hi != nullptr;
#define GO(ST, HH) extract_kernel<ST, HH><<<blocks, 256, 0, ctx->stream>>>(s, koff, k, out.hi, out.lo, out.pay)
    if (stranded) { if (has_hi) GO(true, true); else GO(true, false); }
    else          { if (has_hi) GO(false, true); else GO(false, false); }
#undef GO
    ctx->t_end();
    LAUNCH_CHECK(ctx, "extract_kmers");
    return 0;
}
