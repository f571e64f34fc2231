// Stable LSD radix sort of k-mer records (key hi/lo + payload) in HBM, 8 bits per pass.
// This is the generic any-size path (fallback for partitions that do not fit the LDS sorter and
// for the final ordering of the output table).  Per pass: per-block digit histogram -> exclusive
// scan (digit-major) -> scatter with wave-level multisplit ranking (ballot match, wave64).
// It replaces the reference's per-bucket `sort_by_key` (filter.rs:206); stability keeps equal keys
// in input order exactly like Rust's stable merge sort.
#include "dbg_internal.hpp"
#include <algorithm>

namespace {
constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / DBG_WAVE;
constexpr int RS_ITEMS = 16;                        // rounds of 64 consecutive elements per wave
constexpr int RS_WAVE_CHUNK = RS_ITEMS * DBG_WAVE;  // 1024
constexpr int RS_TILE = RS_WAVES * RS_WAVE_CHUNK;   // 4096

// digit source: 0 = payload (u32), 1 = key lo, 2 = key hi
struct DigitSel {
    int src;
    int shift;
    uint32_t mask;
};

__device__ __forceinline__ uint32_t digit_of(const DigitSel& ds, uint64_t hi, uint64_t lo, uint32_t pay) {
    uint64_t v = ds.src == 0 ? (uint64_t)pay : (ds.src == 1 ? lo : hi);
    return (uint32_t)(v >> ds.shift) & ds.mask;
}

__global__ void __launch_bounds__(RS_THREADS) radix_hist_kernel(RecArrays in, uint32_t n, DigitSel ds,
                                                                uint32_t* __restrict__ hist, uint32_t nblocks) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t base = blockIdx.x * RS_TILE + wave * RS_WAVE_CHUNK + lane;
#pragma unroll 4
    for (int r = 0; r < RS_ITEMS; r++) {
        uint32_t e = base + r * 64;
        if (e < n) {
            uint64_t v = ds.src == 0 ? (uint64_t)in.pay[e] : (ds.src == 1 ? in.lo[e] : in.hi[e]);
            atomicAdd(&h[(uint32_t)(v >> ds.shift) & ds.mask], 1u);
        }
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

template <bool HAS_HI>
__global__ void __launch_bounds__(RS_THREADS) radix_scatter_kernel(RecArrays in, RecArrays out, uint32_t n, DigitSel ds,
                                                                   const uint32_t* __restrict__ hist_scanned,
                                                                   uint32_t nblocks) {
    __shared__ uint32_t wc[RS_WAVES][256];
    for (int i = threadIdx.x; i < RS_WAVES * 256; i += RS_THREADS) (&wc[0][0])[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t base = blockIdx.x * RS_TILE + wave * RS_WAVE_CHUNK + lane;
    const uint64_t lt = lanemask_lt();
    uint64_t hi[RS_ITEMS], lo[RS_ITEMS];
    uint32_t pay[RS_ITEMS];
    uint32_t rank[RS_ITEMS];            // digit in the top byte, wave-local rank in the low 24 bits
    volatile uint32_t* mywc = &wc[wave][0];
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        uint32_t e = base + r * 64;
        bool valid = e < n;
        if (valid) {
            if (HAS_HI) hi[r] = in.hi[e]; else hi[r] = 0;
            lo[r] = in.lo[e];
            pay[r] = in.pay[e];
        } else { hi[r] = 0; lo[r] = 0; pay[r] = 0; }
        uint32_t d = digit_of(ds, hi[r], lo[r], pay[r]);
        // lanes holding the same digit (multisplit by ballot, 8 digit bits)
        uint64_t same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            uint64_t bal = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? bal : ~bal;
        }
        uint32_t before = __popcll(same & lt);
        uint32_t cnt = __popcll(same);
        uint32_t wbase = 0;
        if (valid) wbase = mywc[d];
        if (valid && before == 0) mywc[d] = wbase + cnt;   // leader bumps the wave's running count
        rank[r] = (d << 24) | (wbase + before);
    }
    __syncthreads();
    {   // per digit: turn per-wave totals into global bases (block base + waves before)
        uint32_t d = threadIdx.x;
        uint32_t run = hist_scanned[(size_t)d * nblocks + blockIdx.x];
#pragma unroll
        for (int w = 0; w < RS_WAVES; w++) {
            uint32_t t = wc[w][d];
            wc[w][d] = run;
            run += t;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        uint32_t e = base + r * 64;
        if (e < n) {
            uint32_t d = rank[r] >> 24;
            uint32_t pos = wc[wave][d] + (rank[r] & 0xffffffu);
            if (HAS_HI) out.hi[pos] = hi[r];
            out.lo[pos] = lo[r];
            out.pay[pos] = pay[r];
        }
    }
}
}  // namespace

int radix_sort_records(dbg_ctx* ctx, uint64_t n64, RecArrays a, RecArrays b, int key_bits, int pay_shift,
                       int pay_bits, bool* result_in_b) {
    *result_in_b = false;
    if (n64 <= 1) return 0;
    if (n64 >= (1ull << 32)) return ctx->fail(110, "radix_sort_records: more than 2^32-1 records in one call");
    const uint32_t n = (uint32_t)n64;
    const bool has_hi = a.hi != nullptr;
    std::vector<DigitSel> passes;
    for (int s = 0; s < pay_bits; s += 8)
        passes.push_back(DigitSel{0, pay_shift + s, (uint32_t)((1u << std::min(8, pay_bits - s)) - 1)});
    int lo_bits = std::min(key_bits, 64);
    for (int s = 0; s < lo_bits; s += 8)
        passes.push_back(DigitSel{1, s, (uint32_t)((1u << std::min(8, lo_bits - s)) - 1)});
    for (int s = 0; s + 64 < key_bits; s += 8)
        passes.push_back(DigitSel{2, s, (uint32_t)((1u << std::min(8, key_bits - 64 - s)) - 1)});

    const uint32_t nblocks = cdiv(n, RS_TILE);
    DBuf<uint32_t> hist, hist_scanned;
    ALLOC_OR_FAIL(ctx, hist, (size_t)256 * nblocks);
    ALLOC_OR_FAIL(ctx, hist_scanned, (size_t)256 * nblocks + 1);
    RecArrays src = a, dst = b;
    bool in_b = false;
    for (const DigitSel& ds : passes) {
        ctx->t_begin("radix_hist", n);
        radix_hist_kernel<<<nblocks, RS_THREADS, 0, ctx->stream>>>(src, n, ds, hist.p, nblocks);
        ctx->t_end();
        LAUNCH_CHECK(ctx, "radix_hist");
        DBG_TRY(scan_exclusive_u32(ctx, hist.p, hist_scanned.p, (uint64_t)256 * nblocks));
        ctx->t_begin("radix_scatter", n);
        if (has_hi) radix_scatter_kernel<true><<<nblocks, RS_THREADS, 0, ctx->stream>>>(src, dst, n, ds, hist_scanned.p, nblocks);
        else        radix_scatter_kernel<false><<<nblocks, RS_THREADS, 0, ctx->stream>>>(src, dst, n, ds, hist_scanned.p, nblocks);
        ctx->t_end();
        LAUNCH_CHECK(ctx, "radix_scatter");
        std::swap(src, dst);
        in_b = !in_b;
    }
    *result_in_b = in_b;
    return 0;
}
