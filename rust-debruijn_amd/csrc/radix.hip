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

// ------------------------------------------------------------------------------------------------
// Hybrid sort for the output table: a few stable LSD passes over the TOP key bits leave the array
// grouped by key prefix; `span_sort_kernel` then finishes the order inside LDS.  Workgroup w owns
// the prefix groups that START inside the record window [w*C, (w+1)*C); it loads them (at most CAP
// records), radix-sorts the full key in LDS (stable, 8 bits per pass, same ballot multisplit as
// above) and writes the sorted keys plus the gathered payload.  HBM sees the records 3-4 times
// instead of once per 8 key bits.
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int SS_THREADS = 512;
constexpr int SS_WAVES = SS_THREADS / 64;
constexpr int SS_ITEMS = 6;
constexpr int SS_CAP = SS_THREADS * SS_ITEMS;       // 3072 records per workgroup
constexpr int SS_WINDOW = 2560;                      // C: a prefix group must be <= CAP - C = 512 records

__device__ __forceinline__ uint32_t key_prefix(uint64_t hi, uint64_t lo, int key_bits, int top_bits) {
    int sh = key_bits - top_bits;                    // drop the low bits
    if (sh >= 64) return (uint32_t)(hi >> (sh - 64));
    if (sh == 0) return (uint32_t)lo;
    return (uint32_t)((lo >> sh) | (hi << (64 - sh)));
}

template <bool HAS_HI, bool IS_SET>
__global__ void __launch_bounds__(SS_THREADS) span_sort_kernel(RecArrays in, uint32_t n, int key_bits, int top_bits,
                                                               const uint32_t* __restrict__ u_pay, const uint32_t* __restrict__ u_msk,
                                                               uint64_t* __restrict__ o_hi, uint64_t* __restrict__ o_lo,
                                                               uint8_t* __restrict__ o_exts, uint16_t* __restrict__ o_count,
                                                               uint32_t* __restrict__ o_setn, uint32_t* __restrict__ o_msk,
                                                               uint32_t* __restrict__ flags) {
    __shared__ uint64_t s_lo[SS_CAP];
    __shared__ uint64_t s_hi[HAS_HI ? SS_CAP : 1];
    __shared__ uint32_t s_idx[SS_CAP];
    __shared__ uint32_t wc[SS_WAVES][256];
    __shared__ uint32_t s_ws[SS_WAVES];
    __shared__ uint32_t s_start, s_end;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t lt = lanemask_lt();
    const uint32_t base = blockIdx.x * SS_WINDOW;
    const uint32_t avail = n - base < (uint32_t)SS_CAP ? n - base : (uint32_t)SS_CAP;
    if (tid == 0) { s_start = 0xffffffffu; s_end = 0xffffffffu; }
    __syncthreads();
    // ---- which prefix groups start in this window? ----
    for (uint32_t i = tid; i < avail; i += SS_THREADS) {
        uint32_t g = base + i;
        uint32_t pc = key_prefix(HAS_HI ? in.hi[g] : 0, in.lo[g], key_bits, top_bits);
        bool boundary = g == 0;
        if (!boundary) boundary = pc != key_prefix(HAS_HI ? in.hi[g - 1] : 0, in.lo[g - 1], key_bits, top_bits);
        if (boundary) {
            if (i < (uint32_t)SS_WINDOW) atomicMin(&s_start, i);
            else atomicMin(&s_end, i);
        }
    }
    __syncthreads();
    const uint32_t start = s_start;
    if (start == 0xffffffffu) return;                           // no group starts here
    uint32_t end = s_end;
    if (end == 0xffffffffu) {
        if (avail == (uint32_t)SS_CAP && base + avail < n) {     // a group longer than CAP - C: caller falls back
            if (tid == 0) atomicOr(flags, 1u);
            return;
        }
        end = avail;
    }
    const uint32_t m = end - start;
    // ---- load (positions: wave-chunked, round-major) ----
    uint64_t hi[SS_ITEMS], lo[SS_ITEMS];
    uint32_t idx[SS_ITEMS];
#pragma unroll
    for (int r = 0; r < SS_ITEMS; r++) {
        uint32_t p = wave * (64 * SS_ITEMS) + r * 64 + lane;
        if (p < m) {
            uint32_t g = base + start + p;
            hi[r] = HAS_HI ? in.hi[g] : 0; lo[r] = in.lo[g]; idx[r] = in.pay[g];
        } else { hi[r] = ~0ull; lo[r] = ~0ull; idx[r] = 0xffffffffu; }   // padding: sorts last (stable)
    }
    // ---- stable LSD radix sort in LDS over the full key ----
    for (int shift = 0; shift < key_bits; shift += 8) {
        for (int i = tid; i < SS_WAVES * 256; i += SS_THREADS) (&wc[0][0])[i] = 0;
        __syncthreads();
        uint32_t rank[SS_ITEMS];
        volatile uint32_t* mywc = &wc[wave][0];
#pragma unroll
        for (int r = 0; r < SS_ITEMS; r++) {
            uint64_t v = shift < 64 ? lo[r] : hi[r];
            uint32_t d = (uint32_t)(v >> (shift & 63)) & 0xffu;
            uint64_t same = ~0ull;
#pragma unroll
            for (int b = 0; b < 8; b++) {
                uint64_t bal = __ballot((d >> b) & 1u);
                same &= ((d >> b) & 1u) ? bal : ~bal;
            }
            uint32_t before = __popcll(same & lt);
            uint32_t cnt = __popcll(same);
            uint32_t wbase = mywc[d];
            if (before == 0) mywc[d] = wbase + cnt;
            rank[r] = (d << 24) | (wbase + before);
        }
        __syncthreads();
        // digit totals -> exclusive scan over the 256 digits -> per-wave bases
        uint32_t tot = 0;
        if (tid < 256) {
#pragma unroll
            for (int w = 0; w < SS_WAVES; w++) tot += wc[w][tid];
        }
        uint32_t incl = tot;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) { uint32_t o = __shfl_up(incl, dd); if (lane >= (uint32_t)dd) incl += o; }
        if (lane == 63) s_ws[wave] = incl;
        __syncthreads();
        if (tid < 256) {
            uint32_t off = 0;
            for (uint32_t w = 0; w < wave; w++) off += s_ws[w];
            uint32_t run = off + incl - tot;                      // exclusive prefix of digit tid
#pragma unroll
            for (int w = 0; w < SS_WAVES; w++) { uint32_t t = wc[w][tid]; wc[w][tid] = run; run += t; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SS_ITEMS; r++) {
            uint32_t d = rank[r] >> 24;
            uint32_t pos = wc[wave][d] + (rank[r] & 0xffffffu);
            s_lo[pos] = lo[r];
            if (HAS_HI) s_hi[pos] = hi[r];
            s_idx[pos] = idx[r];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SS_ITEMS; r++) {
            uint32_t p = wave * (64 * SS_ITEMS) + r * 64 + lane;
            lo[r] = s_lo[p];
            if (HAS_HI) hi[r] = s_hi[p];
            idx[r] = s_idx[p];
        }
        __syncthreads();
    }
    // ---- write sorted keys + gathered payload ----
#pragma unroll
    for (int r = 0; r < SS_ITEMS; r++) {
        uint32_t p = wave * (64 * SS_ITEMS) + r * 64 + lane;
        if (p < m) {
            uint32_t g = base + start + p;
            o_hi[g] = HAS_HI ? hi[r] : 0;
            o_lo[g] = lo[r];
            uint32_t pay = u_pay[idx[r]];
            o_exts[g] = (uint8_t)(pay & 0xffu);
            if (IS_SET) { uint32_t mk = u_msk[idx[r]]; o_msk[g] = mk; o_setn[g] = __popc(mk); }
            else o_count[g] = (uint16_t)(pay >> 8);
        }
    }
}
}  // namespace

// Sorts (key, idx) records ascending by key and gathers the payload; returns *ok = false (nothing
// written) when a key-prefix group is too large for the LDS finisher -- the caller then uses the
// plain LSD sort.  a = input (clobbered), b = scratch.
int sort_table_hybrid(dbg_ctx* ctx, uint64_t n64, RecArrays a, RecArrays b, int key_bits, bool is_set,
                      const uint32_t* u_pay, const uint32_t* u_msk, uint64_t* o_hi, uint64_t* o_lo, uint8_t* o_exts,
                      uint16_t* o_count, uint32_t* o_setn, uint32_t* o_msk, bool* ok, bool* data_in_b) {
    *ok = false;
    *data_in_b = false;
    if (n64 == 0) { *ok = true; return 0; }
    if (n64 >= (1ull << 32) - SS_CAP) return 0;
    const uint32_t n = (uint32_t)n64;
    const bool has_hi = a.hi != nullptr;
    // enough top bits that an average prefix group holds ~32 records
    int top_bits = 0;
    while (top_bits < key_bits && top_bits < 24 && (n64 >> top_bits) > 32) top_bits += 8;
    if (top_bits > key_bits) top_bits = key_bits;
    // digit boundaries sit at multiples of 8 from bit 0 so that no digit straddles the two key words
    const int s0 = top_bits ? ((key_bits - top_bits) / 8) * 8 : key_bits;
    top_bits = key_bits - s0;
    RecArrays src = a, dst = b;
    if (top_bits > 0) {
        const uint32_t nblocks = cdiv(n, RS_TILE);
        DBuf<uint32_t> hist, hist_scanned;
        ALLOC_OR_FAIL(ctx, hist, (size_t)256 * nblocks);
        ALLOC_OR_FAIL(ctx, hist_scanned, (size_t)256 * nblocks + 1);
        for (int s = s0; s < key_bits; s += 8) {
            DigitSel ds = s < 64 ? DigitSel{1, s, (uint32_t)((1u << std::min(8, std::min(64, key_bits) - s)) - 1)}
                                 : DigitSel{2, s - 64, (uint32_t)((1u << std::min(8, key_bits - s)) - 1)};
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
            *data_in_b = !*data_in_b;
        }
    }
    DBuf<uint32_t> flags;
    ALLOC_OR_FAIL(ctx, flags, 1);
    HIP_TRY(ctx, hipMemsetAsync(flags.p, 0, 4, ctx->stream));
    const uint32_t nwg = cdiv(n, SS_WINDOW);
    ctx->t_begin("span_sort", n);
#define GO(HH, SS) span_sort_kernel<HH, SS><<<nwg, SS_THREADS, 0, ctx->stream>>>(src, n, key_bits, top_bits, u_pay, u_msk, \
        o_hi, o_lo, o_exts, o_count, o_setn, o_msk, flags.p)
    if (has_hi) { if (is_set) GO(true, true); else GO(true, false); }
    else        { if (is_set) GO(false, true); else GO(false, false); }
#undef GO
    ctx->t_end();
    LAUNCH_CHECK(ctx, "span_sort");
    uint32_t fl = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&fl, flags.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *ok = fl == 0;
    return 0;
}
