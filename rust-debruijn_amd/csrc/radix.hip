// Stable LSD radix sort of k-mer records (key hi/lo + payload) in HBM, 8 bits per pass.
// This is the generic any-size path (fallback for partitions that do not fit the LDS sorter and
// for the final ordering of the output table).  Per pass: per-block digit histogram -> exclusive
// scan (digit-major) -> scatter with wave-level multisplit ranking (ballot match, wave64).
// It replaces the reference's per-bucket `sort_by_key` (filter.rs:206); stability keeps equal keys
// in input order exactly like Rust's stable merge sort.
#include "dbg_internal.hpp"
#include <algorithm>

namespace {
constexpr int RS_THREADS = 512;
constexpr int RS_WAVES = RS_THREADS / DBG_WAVE;
constexpr int RS_ITEMS = 16;                        // rounds of 64 consecutive elements per wave
constexpr int RS_WAVE_CHUNK = RS_ITEMS * DBG_WAVE;  // 1024
constexpr int RS_TILE = RS_WAVES * RS_WAVE_CHUNK;   // 8192: ~32 elements per digit, 256-byte runs in the output

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
    if (threadIdx.x < 256) h[threadIdx.x] = 0;
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
    if (threadIdx.x < 256) hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// One tile: rank every element inside the tile (wave-level ballot multisplit + per-wave digit counters), bring
// the tile into digit order in LDS one array at a time, and write it out linearly: consecutive lanes then hit
// consecutive addresses of a digit's run (8192/256 = 32 elements on average) instead of 64 scattered slots.
// exclusive scan of each pass's 256 totals (one wave per pass)
__global__ void __launch_bounds__(256) radix_global_scan_kernel(const unsigned int* __restrict__ ghist, unsigned int* __restrict__ gexcl, int npass) {
    const uint32_t p = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((int)p >= npass) return;
    uint32_t run = 0;
    for (int c = 0; c < 4; c++) {
        const uint32_t v = ghist[p * 256 + c * 64 + lane];
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= (uint32_t)d) incl += o; }
        gexcl[p * 256 + c * 64 + lane] = run + incl - v;
        run += __shfl(incl, 63);
    }
}

// LOOKBACK: the tile's global digit offsets come from the published counts of earlier tiles instead of a per-tile histogram pass
// (see radix16_onesweep_kernel below for the scheme; status word = tag (4 bits) | count, tags 2p + 1 / 2p + 2 in pass p).
struct LookBack {
    const unsigned int* gexcl;              // exclusive global digit totals of this pass [256]
    unsigned long long* status;             // [tiles][256]
    unsigned int* tile_counter;
    unsigned int* abort_flag;
    uint32_t pass;
};
struct DigitSel4 { DigitSel d[4]; int n; };
__global__ void __launch_bounds__(512) radix_global_hist_kernel(RecArrays in, uint32_t n, DigitSel4 dsel, unsigned int* __restrict__ ghist) {
    __shared__ uint32_t h[4][256];
    for (int i = threadIdx.x; i < 4 * 256; i += blockDim.x) (&h[0][0])[i] = 0;
    __syncthreads();
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const uint64_t lo = in.lo[e], hi = in.hi ? in.hi[e] : 0ull;
        for (int p = 0; p < dsel.n; p++) atomicAdd(&h[p][digit_of(dsel.d[p], hi, lo, 0u)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < dsel.n * 256; i += blockDim.x) { const uint32_t v = (&h[0][0])[i]; if (v) atomicAdd(&ghist[i], v); }
}

template <bool HAS_HI, bool LOOKBACK = false>
__global__ void __launch_bounds__(RS_THREADS) radix_scatter_kernel(RecArrays in, RecArrays out, uint32_t n, DigitSel ds,
                                                                   const uint32_t* __restrict__ hist_scanned,
                                                                   uint32_t nblocks, LookBack lb = LookBack{}) {
    __shared__ uint64_t stage[RS_TILE];                 // 64 KB; the per-wave counters alias its first 8 KB
    __shared__ uint8_t s_dig[RS_TILE];                  // digit of the element at each tile-sorted position
    __shared__ uint32_t s_gbase[256];                   // global position of tile-sorted position 0 of digit d, minus its tile offset
    __shared__ uint32_t s_ws[4];
    __shared__ uint32_t s_tile;
    uint32_t (*wc)[256] = reinterpret_cast<uint32_t (*)[256]>(stage);
    for (int i = threadIdx.x; i < RS_WAVES * 256; i += RS_THREADS) (&wc[0][0])[i] = 0;
    if (LOOKBACK && threadIdx.x == 0) s_tile = atomicAdd(lb.tile_counter, 1u);       // tiles are numbered in the order they start
    __syncthreads();
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t tile = LOOKBACK ? s_tile : blockIdx.x;
    const uint32_t tile0 = tile * RS_TILE;
    const uint32_t base = tile0 + wave * RS_WAVE_CHUNK + lane;
    const uint32_t tile_n = n - tile0 < (uint32_t)RS_TILE ? n - tile0 : (uint32_t)RS_TILE;
    const uint64_t lt = lanemask_lt();
    uint64_t hi[RS_ITEMS], lo[RS_ITEMS];
    uint32_t pay[RS_ITEMS];
    uint32_t rank[RS_ITEMS];            // digit in the top byte, wave-local rank in the low 24 bits; then the tile position
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
    }
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        bool valid = base + r * 64 < n;
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
    const unsigned long long LB_AGG = (unsigned long long)(2 * lb.pass + 1) << 60, LB_INC = (unsigned long long)(2 * lb.pass + 2) << 60;
    uint32_t tot = 0, run0 = 0;
    {   // digit totals of the tile -> exclusive scan over the 256 digits -> per-wave bases inside the tile
        if (tid < 256) {
#pragma unroll
            for (int w = 0; w < RS_WAVES; w++) tot += wc[w][tid];
            if (LOOKBACK) __hip_atomic_store(&lb.status[(size_t)tile * 256 + tid], LB_AGG | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        uint32_t incl = tot;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) { uint32_t o = __shfl_up(incl, dd); if (lane >= (uint32_t)dd) incl += o; }
        if (tid < 256 && lane == 63) s_ws[wave] = incl;
        __syncthreads();
        if (tid < 256) {
            uint32_t off = 0;
            for (uint32_t w = 0; w < wave; w++) off += s_ws[w];
            uint32_t run = off + incl - tot;                      // tile position of the first element with digit tid
            run0 = run;
            if (!LOOKBACK) s_gbase[tid] = hist_scanned[(size_t)tid * nblocks + blockIdx.x] - run;
#pragma unroll
            for (int w = 0; w < RS_WAVES; w++) { uint32_t t = wc[w][tid]; wc[w][tid] = run; run += t; }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        uint32_t d = rank[r] >> 24;
        rank[r] = wc[wave][d] + (rank[r] & 0xffffffu);            // position inside the digit-ordered tile
    }
    __syncthreads();                                              // counters are dead: the stage buffer may be overwritten
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        if (base + r * 64 < n) {
            stage[rank[r]] = lo[r];
            s_dig[rank[r]] = (uint8_t)digit_of(ds, hi[r], lo[r], pay[r]);
        }
    }
    if (LOOKBACK) {
        int failed = 0;
        if (tid < 256) {
            unsigned long long excl = 0;
            for (int64_t p = (int64_t)tile - 1; p >= 0; p--) {
                unsigned long long v = 0;
                uint32_t spins = 0;
                for (;;) {
                    v = __hip_atomic_load(&lb.status[(size_t)p * 256 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((v >> 60) == (LB_AGG >> 60) || (v >> 60) == (LB_INC >> 60)) break;
                    if (++spins > (1u << 22) || __hip_atomic_load(lb.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { failed = 1; break; }
                }
                if (failed) break;
                excl += v & ((1ull << 60) - 1);
                if ((v >> 60) == (LB_INC >> 60)) break;
            }
            if (!failed) {
                __hip_atomic_store(&lb.status[(size_t)tile * 256 + tid], LB_INC | (excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_gbase[tid] = lb.gexcl[tid] + (uint32_t)excl - run0;
            }
        }
        if (__syncthreads_or(failed)) { if (tid == 0) atomicOr(lb.abort_flag, 1u); return; }
    } else __syncthreads();
    for (uint32_t i = tid; i < tile_n; i += RS_THREADS) out.lo[s_gbase[s_dig[i]] + i] = stage[i];
    if (HAS_HI) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RS_ITEMS; r++) if (base + r * 64 < n) stage[rank[r]] = hi[r];
        __syncthreads();
        for (uint32_t i = tid; i < tile_n; i += RS_THREADS) out.hi[s_gbase[s_dig[i]] + i] = stage[i];
    }
    __syncthreads();
    uint32_t* stage32 = reinterpret_cast<uint32_t*>(stage);
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) if (base + r * 64 < n) stage32[rank[r]] = pay[r];
    __syncthreads();
    for (uint32_t i = tid; i < tile_n; i += RS_THREADS) out.pay[s_gbase[s_dig[i]] + i] = stage32[i];
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
// grouped by key prefix (about 32 records per group); `span_sort_kernel` then finishes the order inside
// LDS.  Workgroup w owns the prefix groups that START inside the record window [w*C, (w+1)*C); it loads
// them (at most CAP records), ranks every record inside its own group by direct key comparison (keys of
// the output table are distinct), permutes through LDS and writes the sorted keys together with the
// decoded payload.  HBM sees the records once per top-bit pass plus once here, instead of once per 8 key bits.
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int SS_THREADS = 512;
// Geometry of the walking finishers: 1536 records staged per workgroup, 256 of them shared with the next window (a group of equal
// prefix longer than that sends the call to the plain LSD sort).  Measured at 2.5e8 reads: 3072 / 512 -> 21.5 ms, 1536 / 512 -> 15.3,
// 1536 / 256 -> 11.3: like the wave-per-group finisher, the pass wants more workgroups per CU (loads in flight), not bigger tiles.
#ifndef DBG_SS_ITEMS
#define DBG_SS_ITEMS 3
#endif
constexpr int SS_ITEMS = DBG_SS_ITEMS;
constexpr int SS_CAP = SS_THREADS * SS_ITEMS;       // records per workgroup
#ifndef DBG_SS_OVERLAP
#define DBG_SS_OVERLAP 256
#endif
constexpr int SS_WINDOW = SS_CAP - DBG_SS_OVERLAP;   // C: a prefix group must be <= CAP - C records
// Geometry of the wave-per-group finishers: a group holds at most 128 records, so the window only needs that much overlap with the
// next one (the walking finisher's 512 would re-read a fifth of the table), and a smaller staging area lets more workgroups
// share a CU -- the kernel is a streaming pass with a sort in the middle, and what it lacks is loads in flight.
#ifndef DBG_SG_ITEMS
#define DBG_SG_ITEMS 3
#endif
constexpr int SG_ITEMS = DBG_SG_ITEMS;
constexpr int SG_CAP = SS_THREADS * SG_ITEMS;
constexpr int SG_WINDOW = SG_CAP - 128;

__device__ __forceinline__ uint32_t key_prefix(uint64_t hi, uint64_t lo, int key_bits, int top_bits) {
    int sh = key_bits - top_bits;                    // drop the low bits
    if (sh >= 128) return 0u;                        // no prefix bits at all (a table of <= 32 records at k = 64): `hi >> 64` is not a shift
    if (sh >= 64) return (uint32_t)(hi >> (sh - 64));
    if (sh == 0) return (uint32_t)lo;
    return (uint32_t)((lo >> sh) | (hi << (64 - sh)));
}

// pay = Exts | count << 8 (CountFilter) or Exts | colour mask << 8 (CountFilterSet)
template <bool HAS_HI, bool IS_SET>
__global__ void __launch_bounds__(SS_THREADS) span_sort_kernel(RecArrays in, uint32_t n, int key_bits, int top_bits,
                                                               uint64_t* __restrict__ o_hi, uint64_t* __restrict__ o_lo,
                                                               uint8_t* __restrict__ o_exts, uint16_t* __restrict__ o_count,
                                                               uint32_t* __restrict__ o_setn, uint32_t* __restrict__ o_msk,
                                                               uint32_t* __restrict__ flags) {
    __shared__ uint64_t s_lo[SS_CAP];
    __shared__ uint64_t s_hi[HAS_HI ? SS_CAP : 1];
    __shared__ uint32_t s_pay[SS_CAP];
    __shared__ uint32_t s_pre[SS_CAP];
    __shared__ uint32_t s_start, s_end;
    const uint32_t tid = threadIdx.x;
    const uint32_t base = blockIdx.x * SS_WINDOW;
    const uint32_t avail = n - base < (uint32_t)SS_CAP ? n - base : (uint32_t)SS_CAP;
    if (tid == 0) { s_start = 0xffffffffu; s_end = 0xffffffffu; }
    __syncthreads();
    // ---- load the window (+ overhang) and find which prefix groups start in it ----
    uint64_t hi[SS_ITEMS], lo[SS_ITEMS];
    uint32_t pay[SS_ITEMS];
#pragma unroll
    for (int r = 0; r < SS_ITEMS; r++) {
        const uint32_t i = tid + r * SS_THREADS;
        if (i < avail) {
            const uint32_t g = base + i;
            hi[r] = HAS_HI ? in.hi[g] : 0; lo[r] = in.lo[g]; pay[r] = in.pay[g];
            const uint32_t pc = key_prefix(hi[r], lo[r], key_bits, top_bits);
            s_pre[i] = pc;
            s_lo[i] = lo[r];
            if (HAS_HI) s_hi[i] = hi[r];
        }
    }
    uint32_t prev0 = 0;                                          // prefix of the record just before the window
    if (base > 0 && tid == 0) prev0 = key_prefix(HAS_HI ? in.hi[base - 1] : 0, in.lo[base - 1], key_bits, top_bits);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SS_ITEMS; r++) {
        const uint32_t i = tid + r * SS_THREADS;
        if (i < avail) {
            bool boundary = i == 0 ? (base == 0 || s_pre[0] != prev0) : s_pre[i] != s_pre[i - 1];
            if (boundary) {
                if (i < (uint32_t)SS_WINDOW) atomicMin(&s_start, i);
                else atomicMin(&s_end, i);
            }
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
    // ---- rank inside the group: position = group start + number of smaller keys in the group ----
    uint32_t pos[SS_ITEMS];
#pragma unroll
    for (int r = 0; r < SS_ITEMS; r++) {
        const uint32_t i = tid + r * SS_THREADS;
        pos[r] = 0xffffffffu;
        if (i >= start && i < end) {
            const uint32_t pc = s_pre[i];
            uint32_t less = 0, gs = i;
            for (uint32_t j = i; j > start && s_pre[j - 1] == pc; j--) {
                const uint64_t jl = s_lo[j - 1];
                const bool lt = HAS_HI ? (s_hi[j - 1] < hi[r] || (s_hi[j - 1] == hi[r] && jl <= lo[r])) : jl <= lo[r];   // equal keys keep input order
                less += lt ? 1u : 0u;
                gs = j - 1;
            }
            for (uint32_t j = i + 1; j < end && s_pre[j] == pc; j++) {
                const uint64_t jl = s_lo[j];
                const bool lt = HAS_HI ? (s_hi[j] < hi[r] || (s_hi[j] == hi[r] && jl < lo[r])) : jl < lo[r];
                less += lt ? 1u : 0u;
            }
            pos[r] = gs + less;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SS_ITEMS; r++) {
        if (pos[r] != 0xffffffffu) {
            s_lo[pos[r]] = lo[r];
            if (HAS_HI) s_hi[pos[r]] = hi[r];
            s_pay[pos[r]] = pay[r];
        }
    }
    __syncthreads();
    // ---- write sorted keys + decoded payload, linearly ----
#pragma unroll
    for (int r = 0; r < SS_ITEMS; r++) {
        const uint32_t i = tid + r * SS_THREADS;
        if (i >= start && i < end) {
            const uint32_t g = base + i;
            o_hi[g] = HAS_HI ? s_hi[i] : 0;
            o_lo[g] = s_lo[i];
            const uint32_t p = s_pay[i];
            o_exts[g] = (uint8_t)(p & 0xffu);
            if (IS_SET) { o_msk[g] = p >> 8; if (o_setn) o_setn[g] = __popc(p >> 8); }
            else o_count[g] = (uint16_t)(p >> 8);
        }
    }
}

// Wave-per-group finisher for the three-array record form (keys longer than 96 bits, or DBG_NO_REC16): the same scheme as
// span_sort16_groups_kernel below -- counting sort of every prefix group on the next `dbits` key bits with per-wave LDS
// counters, then members that share a digit ranked by key from their provisional places; groups of up to 128 records,
// two per lane.
constexpr int SG3_MAXD = 6;
template <bool HAS_HI, bool IS_SET>
__global__ void __launch_bounds__(SS_THREADS) span_sort_groups_kernel(RecArrays in, uint32_t n, int key_bits, int top_bits, int dbits,
                                                                      uint64_t* __restrict__ o_hi, uint64_t* __restrict__ o_lo,
                                                                      uint8_t* __restrict__ o_exts, uint16_t* __restrict__ o_count,
                                                                      uint32_t* __restrict__ o_setn, uint32_t* __restrict__ o_msk,
                                                                      uint32_t* __restrict__ flags) {
    __shared__ uint64_t s_lo[SG_CAP];
    __shared__ uint64_t s_hi[HAS_HI ? SG_CAP : 1];
    __shared__ uint32_t s_pay[SG_CAP];
    __shared__ uint32_t s_pre[SG_CAP + 1];              // prefix of every record; afterwards: start of group g
    __shared__ uint32_t s_cnt[SS_THREADS / 64][1 << SG3_MAXD];
    __shared__ uint32_t s_start, s_end, s_ng;
    __shared__ uint32_t s_wsum[SS_THREADS / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t base = blockIdx.x * SG_WINDOW;
    const uint32_t avail = n - base < (uint32_t)SG_CAP ? n - base : (uint32_t)SG_CAP;
    if (tid == 0) { s_start = 0xffffffffu; s_end = 0xffffffffu; }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SG_ITEMS; r++) {
        const uint32_t i = tid + r * SS_THREADS;
        if (i < avail) {
            const uint32_t g = base + i;
            const uint64_t h = HAS_HI ? in.hi[g] : 0, l = in.lo[g];
            s_lo[i] = l; if (HAS_HI) s_hi[i] = h; s_pay[i] = in.pay[g];
            s_pre[i] = key_prefix(h, l, key_bits, top_bits);
        }
    }
    uint32_t prev0 = 0;
    if (base > 0 && tid == 0) prev0 = key_prefix(HAS_HI ? in.hi[base - 1] : 0, in.lo[base - 1], key_bits, top_bits);
    __syncthreads();
    uint32_t bflags = 0;
#pragma unroll
    for (int r = 0; r < SG_ITEMS; r++) {
        const uint32_t i = tid * SG_ITEMS + r;
        if (i < avail) {
            const uint32_t pc = s_pre[i];
            const bool boundary = i == 0 ? (base == 0 || pc != prev0) : pc != s_pre[i - 1];
            if (boundary) {
                bflags |= 1u << r;
                if (i < (uint32_t)SG_WINDOW) atomicMin(&s_start, i); else atomicMin(&s_end, i);
            }
        }
    }
    __syncthreads();
    const uint32_t start = s_start;
    if (start == 0xffffffffu) return;
    uint32_t end = s_end;
    if (end == 0xffffffffu) {
        if (avail == (uint32_t)SG_CAP && base + avail < n) { if (tid == 0) atomicOr(flags, 1u); return; }
        end = avail;
    }
    {
        uint32_t mine = 0;
#pragma unroll
        for (int r = 0; r < SG_ITEMS; r++) { const uint32_t i = tid * SG_ITEMS + r; if (((bflags >> r) & 1u) && i >= start && i < end) mine++; }
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= (uint32_t)d) incl += o; }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        uint32_t off = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < SS_THREADS / 64; w++) { const uint32_t x = s_wsum[w]; off += (uint32_t)w < wave ? x : 0u; tot += x; }
        uint32_t g = off + incl - mine;
#pragma unroll
        for (int r = 0; r < SG_ITEMS; r++) { const uint32_t i = tid * SG_ITEMS + r; if (((bflags >> r) & 1u) && i >= start && i < end) s_pre[g++] = i; }
        if (tid == 0) { s_ng = tot; s_pre[tot] = end; }
    }
    __syncthreads();
    const uint32_t ng = s_ng;
    const int dshift = key_bits - top_bits - dbits;             // the digit = key bits [dshift, dshift + dbits): inside one key word (caller)
    const uint32_t nd = 1u << dbits, dmask = nd - 1u;
    uint32_t* cnt = s_cnt[wave];
    struct R3 { uint64_t hi, lo; uint32_t pay; };
    auto get = [&](uint32_t i) { return R3{HAS_HI ? s_hi[i] : 0ull, s_lo[i], s_pay[i]}; };
    auto put = [&](uint32_t i, const R3& q) { s_lo[i] = q.lo; if (HAS_HI) s_hi[i] = q.hi; s_pay[i] = q.pay; };
    auto dig = [&](const R3& q) { return (uint32_t)((dshift >= 64 ? q.hi >> (dshift - 64) : q.lo >> dshift)) & dmask; };
    auto lt = [&](uint32_t j, const R3& q) { const uint64_t jl = s_lo[j]; return HAS_HI ? (s_hi[j] < q.hi || (s_hi[j] == q.hi && jl < q.lo)) : jl < q.lo; };
    for (uint32_t gi = wave; gi < ng; gi += SS_THREADS / 64) {
        const uint32_t gs = s_pre[gi], gn = s_pre[gi + 1] - gs;
        if (gn > 128) { if (lane == 0) atomicOr(flags, 1u); continue; }
        if (gn == 1) continue;
        const bool hasA = lane < gn, hasB = 64 + lane < gn;
        R3 qa{0, 0, 0}, qb{0, 0, 0};
        if (hasA) qa = get(gs + lane);
        if (hasB) qb = get(gs + 64 + lane);
        for (uint32_t d = lane; d < nd; d += 64) cnt[d] = 0;
        const uint32_t da = dig(qa), db = dig(qb);
        uint32_t ra = 0, rb = 0;
        if (hasA) ra = atomicAdd(&cnt[da], 1u);
        if (hasB) rb = atomicAdd(&cnt[db], 1u);
        uint32_t carry = 0;
        for (uint32_t d0 = 0; d0 < nd; d0 += 64) {
            const uint32_t c = d0 + lane < nd ? cnt[d0 + lane] : 0u;
            uint32_t incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= (uint32_t)d) incl += o; }
            if (d0 + lane < nd) cnt[d0 + lane] = carry + incl - c;
            carry += __shfl(incl, 63);
        }
        const uint32_t ca = hasA ? cnt[da] : 0u, cb = hasB ? cnt[db] : 0u;
        if (hasA) put(gs + ca + ra, qa);
        if (hasB) put(gs + cb + rb, qb);
        const uint32_t ea = hasA ? (da + 1 < nd ? cnt[da + 1] : gn) : 0u, eb = hasB ? (db + 1 < nd ? cnt[db + 1] : gn) : 0u;
        uint32_t la = 0, lb = 0;
        if (hasA && ea - ca > 1) for (uint32_t j = ca; j < ea; j++) la += lt(gs + j, qa) ? 1u : 0u;
        if (hasB && eb - cb > 1) for (uint32_t j = cb; j < eb; j++) lb += lt(gs + j, qb) ? 1u : 0u;
        if (hasA && ea - ca > 1) put(gs + ca + la, qa);
        if (hasB && eb - cb > 1) put(gs + cb + lb, qb);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SG_ITEMS; r++) {
        const uint32_t i = tid + r * SS_THREADS;
        if (i >= start && i < end) {
            const uint32_t g = base + i;
            o_hi[g] = HAS_HI ? s_hi[i] : 0;
            o_lo[g] = s_lo[i];
            const uint32_t p = s_pay[i];
            o_exts[g] = (uint8_t)(p & 0xffu);
            if (IS_SET) { o_msk[g] = p >> 8; if (o_setn) o_setn[g] = __popc(p >> 8); }
            else o_count[g] = (uint16_t)(p >> 8);
        }
    }
}

template <bool IS_SET>
__global__ void decode_payload_kernel(uint32_t n, const uint32_t* __restrict__ pay, uint8_t* __restrict__ exts,
                                      uint16_t* __restrict__ count, uint32_t* __restrict__ setn, uint32_t* __restrict__ msk) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = pay[i];
    exts[i] = (uint8_t)(p & 0xffu);
    if (IS_SET) { msk[i] = p >> 8; if (setn) setn[i] = __popc(p >> 8); }
    else count[i] = (uint16_t)(p >> 8);
}
}  // namespace

// Sorts (key, payload) records ascending by key and writes the output table columns.  a = input (clobbered),
// b = scratch.  Falls back to the plain LSD sort over every key bit when a key-prefix group is too large for
// the LDS finisher (the records are intact in whichever buffer the top-bit passes ended in).
int sort_table_hybrid(dbg_ctx* ctx, uint64_t n64, RecArrays a, RecArrays b, int key_bits, bool is_set, bool allow_hybrid,
                      uint64_t* o_hi, uint64_t* o_lo, uint8_t* o_exts, uint16_t* o_count, uint32_t* o_setn, uint32_t* o_msk) {
    if (n64 == 0) return 0;
    if (n64 >= (1ull << 32) - SS_CAP) return ctx->fail(110, "sort_table: more than 2^32-1 records in one call");
    const uint32_t n = (uint32_t)n64;
    const bool has_hi = a.hi != nullptr;
    RecArrays src = a, dst = b;
    if (allow_hybrid) {
        // enough top bits that an average prefix group holds ~32 records
        int top_bits = 0;
        while (top_bits < key_bits && top_bits < 32 && (n64 >> top_bits) > 32) top_bits += 8;
        if (top_bits > key_bits) top_bits = key_bits;
        // Passes start exactly at bit key_bits - top_bits and the wave-per-group finisher orders the groups on the next 6 bits,
        // when those bits and the pass digits all lie inside one of the two 64-bit key words (always true for k >= 49, whose
        // top 30 bits sit in the high word).  Otherwise, and with DBG_SORT=bytealigned, digit boundaries sit at multiples of 8
        // from bit 0 -- up to 7 more prefix bits (30 at k = 51: a 4th, 6-bit pass) -- and the walking finisher runs.
        const int dbits = std::min(6, key_bits - top_bits);
        int s0 = key_bits - top_bits;
        const bool one_word = top_bits > 0 && dbits > 0 && (s0 - dbits >= 64 || key_bits <= 64);
        const bool groups = one_word && !(ctx->opt("DBG_SORT") && !strcmp(ctx->opt("DBG_SORT"), "bytealigned")) && (n64 >> top_bits) >= 16;   // see sort_table_hybrid16
        if (!groups) { s0 = top_bits ? (s0 / 8) * 8 : key_bits; top_bits = key_bits - s0; }
        if (top_bits > 0) {
            const uint32_t nblocks = cdiv(n, RS_TILE);
            DBuf<uint32_t> hist, hist_scanned;
            ALLOC_OR_FAIL(ctx, hist, (size_t)256 * nblocks);
            ALLOC_OR_FAIL(ctx, hist_scanned, (size_t)256 * nblocks + 1);
            auto sel_of = [&](int s) {
                return s < 64 ? DigitSel{1, s, (uint32_t)((1u << std::min(8, std::min(64, key_bits) - s)) - 1)}
                              : DigitSel{2, s - 64, (uint32_t)((1u << std::min(8, key_bits - s)) - 1)};
            };
            // look-back passes (default; DBG_ONESWEEP=0: per-tile histogram + scan + scatter), as in sort_table_hybrid16
            const int npass = (key_bits - s0 + 7) / 8;
            bool lookback = !(ctx->opt("DBG_ONESWEEP") && !strcmp(ctx->opt("DBG_ONESWEEP"), "0")) && npass >= 1 && npass <= 4;
            DBuf<unsigned int> ghist, gexcl, lb_ctl;
            DBuf<unsigned long long> lb_status;
            if (lookback) {
                ALLOC_OR_FAIL(ctx, ghist, 4 * 256); ALLOC_OR_FAIL(ctx, gexcl, 4 * 256); ALLOC_OR_FAIL(ctx, lb_ctl, 8);
                ALLOC_OR_FAIL(ctx, lb_status, (size_t)nblocks * 256);
                HIP_TRY(ctx, hipMemsetAsync(ghist.p, 0, 4 * 256 * 4, ctx->stream));
                HIP_TRY(ctx, hipMemsetAsync(lb_status.p, 0, (size_t)nblocks * 256 * 8, ctx->stream));
                HIP_TRY(ctx, hipMemsetAsync(lb_ctl.p, 0, 8 * 4, ctx->stream));
                DigitSel4 d4{};
                d4.n = npass;
                for (int pi = 0; pi < npass; pi++) d4.d[pi] = sel_of(s0 + 8 * pi);
                ctx->t_begin("radix_hist", n);
                radix_global_hist_kernel<<<std::max<uint32_t>(1, std::min<uint32_t>(cdiv(n, 512 * 16), 2048)), 512, 0, ctx->stream>>>(src, n, d4, ghist.p);
                radix_global_scan_kernel<<<1, 256, 0, ctx->stream>>>(ghist.p, gexcl.p, npass);
                ctx->t_end();
                LAUNCH_CHECK(ctx, "radix_global_hist");
            }
            int pi = 0;
            for (int s = s0; s < key_bits; s += 8, pi++) {
                const DigitSel ds = sel_of(s);
                bool done = false;
                if (lookback) {
                    LookBack lb{gexcl.p + pi * 256, lb_status.p, lb_ctl.p + 2 * pi, lb_ctl.p + 2 * pi + 1, (uint32_t)pi};
                    ctx->t_begin("radix_scatter", n);
                    if (has_hi) radix_scatter_kernel<true, true><<<nblocks, RS_THREADS, 0, ctx->stream>>>(src, dst, n, ds, nullptr, nblocks, lb);
                    else        radix_scatter_kernel<false, true><<<nblocks, RS_THREADS, 0, ctx->stream>>>(src, dst, n, ds, nullptr, nblocks, lb);
                    ctx->t_end();
                    LAUNCH_CHECK(ctx, "radix_scatter_lookback");
                    unsigned int ctl[2] = {0, 0};                      // (the pass's input is only intact until the next pass writes over it)
                    HIP_TRY(ctx, hipMemcpyAsync(ctl, lb_ctl.p + 2 * pi, 8, hipMemcpyDeviceToHost, ctx->stream));
                    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                    done = ctl[1] == 0;
                    if (!done) { lookback = false; if (ctx->opt("DBG_DEBUG")) fprintf(stderr, "[sort] look-back pass gave up (shift %d): classic pass instead\n", s); }
                }
                if (!done) {
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
                }
                std::swap(src, dst);
            }
        }
        DBuf<uint32_t> flags;
        ALLOC_OR_FAIL(ctx, flags, 1);
        const uint32_t nwg = cdiv(n, SS_WINDOW);
        for (int form = groups ? 0 : 1; form < 2; form++) {            // 0: wave per group; 1: walking finisher (long groups)
            HIP_TRY(ctx, hipMemsetAsync(flags.p, 0, 4, ctx->stream));
            ctx->t_begin("span_sort", n);
            if (form == 0) {
#define GO(HH, SS) span_sort_groups_kernel<HH, SS><<<cdiv(n, SG_WINDOW), SS_THREADS, 0, ctx->stream>>>(src, n, key_bits, top_bits, dbits, \
                o_hi, o_lo, o_exts, o_count, o_setn, o_msk, flags.p)
                if (has_hi) { if (is_set) GO(true, true); else GO(true, false); }
                else        { if (is_set) GO(false, true); else GO(false, false); }
#undef GO
            } else {
#define GO(HH, SS) span_sort_kernel<HH, SS><<<nwg, SS_THREADS, 0, ctx->stream>>>(src, n, key_bits, top_bits, \
                o_hi, o_lo, o_exts, o_count, o_setn, o_msk, flags.p)
                if (has_hi) { if (is_set) GO(true, true); else GO(true, false); }
                else        { if (is_set) GO(false, true); else GO(false, false); }
#undef GO
            }
            ctx->t_end();
            LAUNCH_CHECK(ctx, "span_sort");
            uint32_t fl = 0;
            HIP_TRY(ctx, hipMemcpyAsync(&fl, flags.p, 4, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (fl == 0) return 0;
            if (ctx->opt("DBG_DEBUG")) fprintf(stderr, "[sort] finisher form %d met a group of equal prefix that is too long for it (n=%u, top_bits=%d)\n", form, n, top_bits);
        }
    }
    // plain LSD sort over every key bit
    bool in_b = false;
    DBG_TRY(radix_sort_records(ctx, n64, src, dst, key_bits, 0, 0, &in_b));
    RecArrays S = in_b ? dst : src;
    if (has_hi) HIP_TRY(ctx, hipMemcpyAsync(o_hi, S.hi, n64 * 8, hipMemcpyDeviceToDevice, ctx->stream));
    else HIP_TRY(ctx, hipMemsetAsync(o_hi, 0, n64 * 8, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(o_lo, S.lo, n64 * 8, hipMemcpyDeviceToDevice, ctx->stream));
    ctx->t_begin("finalize", n64);
    if (is_set) decode_payload_kernel<true><<<cdiv(n, 256), 256, 0, ctx->stream>>>(n, S.pay, o_exts, nullptr, o_setn, o_msk);
    else decode_payload_kernel<false><<<cdiv(n, 256), 256, 0, ctx->stream>>>(n, S.pay, o_exts, o_count, nullptr, nullptr);
    ctx->t_end();
    LAUNCH_CHECK(ctx, "finalize");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// 16-byte record form of the same hybrid sort, for keys of at most 96 bits (k <= 48): one record =
// uint4 {key bits 0..31, 32..63, 64..95, payload}.  One 128-bit load/store per record instead of three
// arrays, one LDS staging pass per tile instead of three, 16 B instead of 20 B of traffic per record and pass.
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int R16_THREADS = 512;
constexpr int R16_WAVES = R16_THREADS / DBG_WAVE;
#ifndef DBG_R16_ITEMS
#define DBG_R16_ITEMS 8
#endif
constexpr int R16_ITEMS = DBG_R16_ITEMS;
constexpr int R16_TILE = R16_THREADS * R16_ITEMS;       // 4096 records = 64 KB of staging: ~16 records (256 B) per digit

__device__ __forceinline__ uint32_t digit16(const uint4& r, int shift) {        // 8 key bits from bit `shift` (< 96) on; a digit may straddle two words
    const uint32_t w = (uint32_t)shift >> 5, s = (uint32_t)shift & 31u;
    const uint32_t lo = w == 0 ? r.x : (w == 1 ? r.y : r.z), hi = w == 0 ? r.y : (w == 1 ? r.z : 0u);
    return (uint32_t)(((((uint64_t)hi << 32) | lo) >> s) & 0xffu);
}

__global__ void __launch_bounds__(R16_THREADS) radix16_hist_kernel(const uint4* __restrict__ in, uint32_t n, int shift, uint32_t mask,
                                                                   uint32_t* __restrict__ hist, uint32_t nblocks) {
    __shared__ uint32_t h[256];
    if (threadIdx.x < 256) h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t base = blockIdx.x * R16_TILE + wave * (R16_ITEMS * 64) + lane;
    const bool one_word = (shift & 31) <= 24;                     // the digit lies inside one 32-bit word: fetch only that word
    const uint32_t* w32 = reinterpret_cast<const uint32_t*>(in) + (shift >> 5);
#pragma unroll
    for (int r = 0; r < R16_ITEMS; r++) {
        const uint32_t e = base + r * 64;
        if (e < n) {
            const uint32_t d = one_word ? (w32[(size_t)e * 4] >> (shift & 31)) & mask : digit16(in[e], shift) & mask;
            atomicAdd(&h[d], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 256) hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

__global__ void __launch_bounds__(R16_THREADS) radix16_scatter_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t n, int shift,
                                                                      uint32_t mask, const uint32_t* __restrict__ hist_scanned, uint32_t nblocks) {
    __shared__ uint4 stage[R16_TILE];                   // 64 KB; the per-wave counters alias its first 8 KB
    __shared__ uint8_t s_dig[R16_TILE];
    __shared__ uint32_t s_gbase[256];
    __shared__ uint32_t s_ws[4];
    uint32_t (*wc)[256] = reinterpret_cast<uint32_t (*)[256]>(stage);
    for (int i = threadIdx.x; i < R16_WAVES * 256; i += R16_THREADS) (&wc[0][0])[i] = 0;
    __syncthreads();
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t tile0 = blockIdx.x * R16_TILE;
    const uint32_t base = tile0 + wave * (R16_ITEMS * 64) + lane;
    const uint32_t tile_n = n - tile0 < (uint32_t)R16_TILE ? n - tile0 : (uint32_t)R16_TILE;
    const uint64_t lt = lanemask_lt();
    uint4 rec[R16_ITEMS];
    uint32_t rank[R16_ITEMS];
    volatile uint32_t* mywc = &wc[wave][0];
#pragma unroll
    for (int r = 0; r < R16_ITEMS; r++) {
        const uint32_t e = base + r * 64;
        rec[r] = e < n ? in[e] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < R16_ITEMS; r++) {
        const bool valid = base + r * 64 < n;
        const uint32_t d = digit16(rec[r], shift) & mask;
        uint64_t same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const uint64_t bal = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? bal : ~bal;
        }
        const uint32_t before = __popcll(same & lt), cnt = __popcll(same);
        uint32_t wbase = 0;
        if (valid) wbase = mywc[d];
        if (valid && before == 0) mywc[d] = wbase + cnt;
        rank[r] = (d << 24) | (wbase + before);
    }
    __syncthreads();
    {
        uint32_t tot = 0;
        if (tid < 256) {
#pragma unroll
            for (int w = 0; w < R16_WAVES; w++) tot += wc[w][tid];
        }
        uint32_t incl = tot;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) { uint32_t o = __shfl_up(incl, dd); if (lane >= (uint32_t)dd) incl += o; }
        if (tid < 256 && lane == 63) s_ws[wave] = incl;
        __syncthreads();
        if (tid < 256) {
            uint32_t off = 0;
            for (uint32_t w = 0; w < wave; w++) off += s_ws[w];
            uint32_t run = off + incl - tot;
            s_gbase[tid] = hist_scanned[(size_t)tid * nblocks + blockIdx.x] - run;
#pragma unroll
            for (int w = 0; w < R16_WAVES; w++) { uint32_t t = wc[w][tid]; wc[w][tid] = run; run += t; }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R16_ITEMS; r++) rank[r] = wc[wave][rank[r] >> 24] + (rank[r] & 0xffffffu);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R16_ITEMS; r++) {
        if (base + r * 64 < n) { stage[rank[r]] = rec[r]; s_dig[rank[r]] = (uint8_t)(digit16(rec[r], shift) & mask); }
    }
    __syncthreads();
    for (uint32_t i = tid; i < tile_n; i += R16_THREADS) out[s_gbase[s_dig[i]] + i] = stage[i];
}

__device__ __forceinline__ uint32_t prefix16(const uint4& r, int key_bits, int top_bits) {
    const uint64_t lo = ((uint64_t)r.y << 32) | r.x, hi = r.z;
    const int sh = key_bits - top_bits;
    if (sh >= 64) return (uint32_t)(hi >> (sh - 64));
    if (sh == 0) return (uint32_t)lo;
    return (uint32_t)((lo >> sh) | (hi << (64 - sh)));
}
__device__ __forceinline__ bool less16(const uint4& a, const uint4& b, bool or_equal) {
    if (a.z != b.z) return a.z < b.z;
    if (a.y != b.y) return a.y < b.y;
    if (a.x != b.x) return a.x < b.x;
    return or_equal;
}

template <bool IS_SET>
__global__ void __launch_bounds__(SS_THREADS) span_sort16_kernel(const uint4* __restrict__ in, uint32_t n, int key_bits, int top_bits,
                                                                 uint64_t* __restrict__ o_hi, uint64_t* __restrict__ o_lo,
                                                                 uint8_t* __restrict__ o_exts, uint16_t* __restrict__ o_count,
                                                                 uint32_t* __restrict__ o_setn, uint32_t* __restrict__ o_msk,
                                                                 uint32_t* __restrict__ flags) {
    __shared__ uint4 s_rec[SS_CAP];                     // 48 KB
    __shared__ uint32_t s_pre[SS_CAP];
    __shared__ uint32_t s_start, s_end;
    const uint32_t tid = threadIdx.x;
    const uint32_t base = blockIdx.x * SS_WINDOW;
    const uint32_t avail = n - base < (uint32_t)SS_CAP ? n - base : (uint32_t)SS_CAP;
    if (tid == 0) { s_start = 0xffffffffu; s_end = 0xffffffffu; }
    __syncthreads();
    uint4 rec[SS_ITEMS];
#pragma unroll
    for (int r = 0; r < SS_ITEMS; r++) {
        const uint32_t i = tid + r * SS_THREADS;
        if (i < avail) {
            rec[r] = in[base + i];
            s_pre[i] = prefix16(rec[r], key_bits, top_bits);
            s_rec[i] = rec[r];
        }
    }
    uint32_t prev0 = 0;
    if (base > 0 && tid == 0) prev0 = prefix16(in[base - 1], key_bits, top_bits);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SS_ITEMS; r++) {
        const uint32_t i = tid + r * SS_THREADS;
        if (i < avail) {
            const bool boundary = i == 0 ? (base == 0 || s_pre[0] != prev0) : s_pre[i] != s_pre[i - 1];
            if (boundary) {
                if (i < (uint32_t)SS_WINDOW) atomicMin(&s_start, i);
                else atomicMin(&s_end, i);
            }
        }
    }
    __syncthreads();
    const uint32_t start = s_start;
    if (start == 0xffffffffu) return;
    uint32_t end = s_end;
    if (end == 0xffffffffu) {
        if (avail == (uint32_t)SS_CAP && base + avail < n) { if (tid == 0) atomicOr(flags, 1u); return; }
        end = avail;
    }
    uint32_t pos[SS_ITEMS];
#pragma unroll
    for (int r = 0; r < SS_ITEMS; r++) {
        const uint32_t i = tid + r * SS_THREADS;
        pos[r] = 0xffffffffu;
        if (i >= start && i < end) {
            const uint32_t pc = s_pre[i];
            uint32_t less = 0, gs = i;
            for (uint32_t j = i; j > start && s_pre[j - 1] == pc; j--) { less += less16(s_rec[j - 1], rec[r], true) ? 1u : 0u; gs = j - 1; }
            for (uint32_t j = i + 1; j < end && s_pre[j] == pc; j++) less += less16(s_rec[j], rec[r], false) ? 1u : 0u;
            pos[r] = gs + less;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SS_ITEMS; r++) if (pos[r] != 0xffffffffu) s_rec[pos[r]] = rec[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SS_ITEMS; r++) {
        const uint32_t i = tid + r * SS_THREADS;
        if (i >= start && i < end) {
            const uint32_t g = base + i;
            const uint4 q = s_rec[i];
            o_hi[g] = q.z;
            o_lo[g] = ((uint64_t)q.y << 32) | q.x;
            o_exts[g] = (uint8_t)(q.w & 0xffu);
            if (IS_SET) { o_msk[g] = q.w >> 8; if (o_setn) o_setn[g] = __popc(q.w >> 8); }
            else o_count[g] = (uint16_t)(q.w >> 8);
        }
    }
}

// The finisher for prefix groups of a few dozen records (three top-bit passes = 24 prefix bits instead of 30: one pass over HBM
// less).  Ranking a record by walking its group (span_sort16_kernel) is a chain of dependent LDS reads as long as the group.
// Here a WAVE takes a group: (1) a counting sort on the next `dbits` key bits -- one LDS counter per digit value, every member
// takes its rank inside its digit with an LDS atomic, the wave scans the counters -- puts the members in (prefix + digit) order;
// (2) members that share a digit (on average every second one has a partner) are ranked among themselves by full key from their
// provisional places.  Groups of up to 128 records are held two per lane; longer ones (heavily repeated prefixes) raise the flag
// and the caller runs the walking finisher.  Canonical keys are denser at small values (min(k-mer, rc)), so groups of twice
// the average size are normal.
constexpr int SG_MAXD = 6;
template <bool IS_SET>
__global__ void __launch_bounds__(SS_THREADS) span_sort16_groups_kernel(const uint4* __restrict__ in, uint32_t n, int key_bits, int top_bits, int dbits,
                                                                        uint64_t* __restrict__ o_hi, uint64_t* __restrict__ o_lo,
                                                                        uint8_t* __restrict__ o_exts, uint16_t* __restrict__ o_count,
                                                                        uint32_t* __restrict__ o_setn, uint32_t* __restrict__ o_msk,
                                                                        uint32_t* __restrict__ flags) {
    __shared__ uint4 s_rec[SG_CAP];                     // 48 KB
    __shared__ uint32_t s_pre[SG_CAP + 1];              // prefix of every record; afterwards: start of group g
    __shared__ uint32_t s_cnt[SS_THREADS / 64][1 << SG_MAXD];   // per wave: members per digit value, then exclusive prefix
    __shared__ uint32_t s_start, s_end, s_ng;
    __shared__ uint32_t s_wsum[SS_THREADS / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t base = blockIdx.x * SG_WINDOW;
    const uint32_t avail = n - base < (uint32_t)SG_CAP ? n - base : (uint32_t)SG_CAP;
    if (tid == 0) { s_start = 0xffffffffu; s_end = 0xffffffffu; }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SG_ITEMS; r++) {
        const uint32_t i = tid + r * SS_THREADS;                   // coalesced load order
        if (i < avail) { const uint4 q = in[base + i]; s_rec[i] = q; s_pre[i] = prefix16(q, key_bits, top_bits); }
    }
    uint32_t prev0 = 0;
    if (base > 0 && tid == 0) prev0 = prefix16(in[base - 1], key_bits, top_bits);
    __syncthreads();
    // thread t looks at records t*ITEMS .. t*ITEMS + ITEMS - 1 (consecutive: the group enumeration is a scan in record order)
    uint32_t bflags = 0;
#pragma unroll
    for (int r = 0; r < SG_ITEMS; r++) {
        const uint32_t i = tid * SG_ITEMS + r;
        if (i < avail) {
            const uint32_t pc = s_pre[i];
            const bool boundary = i == 0 ? (base == 0 || pc != prev0) : pc != s_pre[i - 1];
            if (boundary) {
                bflags |= 1u << r;
                if (i < (uint32_t)SG_WINDOW) atomicMin(&s_start, i); else atomicMin(&s_end, i);
            }
        }
    }
    __syncthreads();
    const uint32_t start = s_start;
    if (start == 0xffffffffu) return;                           // no group starts here
    uint32_t end = s_end;
    if (end == 0xffffffffu) {
        if (avail == (uint32_t)SG_CAP && base + avail < n) { if (tid == 0) atomicOr(flags, 1u); return; }
        end = avail;
    }
    {   // enumerate the groups that start in [start, end): s_pre[g] <- first record of group g, s_pre[ng] <- end
        uint32_t mine = 0;
#pragma unroll
        for (int r = 0; r < SG_ITEMS; r++) { const uint32_t i = tid * SG_ITEMS + r; if (((bflags >> r) & 1u) && i >= start && i < end) mine++; }
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= (uint32_t)d) incl += o; }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();                                           // (every thread is done reading the prefixes in s_pre)
        uint32_t off = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < SS_THREADS / 64; w++) { const uint32_t x = s_wsum[w]; off += (uint32_t)w < wave ? x : 0u; tot += x; }
        uint32_t g = off + incl - mine;
#pragma unroll
        for (int r = 0; r < SG_ITEMS; r++) { const uint32_t i = tid * SG_ITEMS + r; if (((bflags >> r) & 1u) && i >= start && i < end) s_pre[g++] = i; }
        if (tid == 0) { s_ng = tot; s_pre[tot] = end; }
    }
    __syncthreads();
    const uint32_t ng = s_ng;
    const int dshift = key_bits - top_bits - dbits;             // the digit = key bits [dshift, dshift + dbits)
    const uint32_t nd = 1u << dbits, dmask = nd - 1u;
    uint32_t* cnt = s_cnt[wave];
    for (uint32_t gi = wave; gi < ng; gi += SS_THREADS / 64) {
        const uint32_t gs = s_pre[gi], gn = s_pre[gi + 1] - gs;
        if (gn > 128) { if (lane == 0) atomicOr(flags, 1u); continue; }
        if (gn == 1) continue;
        const bool hasA = lane < gn, hasB = 64 + lane < gn;
        uint4 qa = make_uint4(0, 0, 0, 0), qb = make_uint4(0, 0, 0, 0);
        if (hasA) qa = s_rec[gs + lane];
        if (hasB) qb = s_rec[gs + 64 + lane];
        for (uint32_t d = lane; d < nd; d += 64) cnt[d] = 0;
        const uint32_t da = digit16(qa, dshift) & dmask, db = digit16(qb, dshift) & dmask;
        uint32_t ra = 0, rb = 0;                                   // rank inside the digit (any order: step 2 orders equal digits)
        if (hasA) ra = atomicAdd(&cnt[da], 1u);
        if (hasB) rb = atomicAdd(&cnt[db], 1u);
        // exclusive prefix of the counters: lanes take digit values in turn (nd <= 64: one round)
        uint32_t carry = 0;
        for (uint32_t d0 = 0; d0 < nd; d0 += 64) {
            const uint32_t c = d0 + lane < nd ? cnt[d0 + lane] : 0u;
            uint32_t incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= (uint32_t)d) incl += o; }
            if (d0 + lane < nd) cnt[d0 + lane] = carry + incl - c;
            carry += __shfl(incl, 63);
        }
        // provisional places: (prefix + digit) order
        const uint32_t ca = hasA ? cnt[da] : 0u, cb = hasB ? cnt[db] : 0u;                 // first place of my digit
        if (hasA) s_rec[gs + ca + ra] = qa;
        if (hasB) s_rec[gs + cb + rb] = qb;
        // members that share a digit: rank by key among the run [c, c_next)
        const uint32_t ea = hasA ? (da + 1 < nd ? cnt[da + 1] : gn) : 0u, eb = hasB ? (db + 1 < nd ? cnt[db + 1] : gn) : 0u;
        uint32_t la = 0, lb = 0;
        if (hasA && ea - ca > 1) for (uint32_t j = ca; j < ea; j++) la += less16(s_rec[gs + j], qa, false) ? 1u : 0u;
        if (hasB && eb - cb > 1) for (uint32_t j = cb; j < eb; j++) lb += less16(s_rec[gs + j], qb, false) ? 1u : 0u;
        // (the wave's LDS operations are performed in order: every read above precedes the writes below)
        if (hasA && ea - ca > 1) s_rec[gs + ca + la] = qa;
        if (hasB && eb - cb > 1) s_rec[gs + cb + lb] = qb;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SG_ITEMS; r++) {
        const uint32_t i = tid + r * SS_THREADS;
        if (i >= start && i < end) {
            const uint32_t g = base + i;
            const uint4 q = s_rec[i];
            o_hi[g] = q.z;
            o_lo[g] = ((uint64_t)q.y << 32) | q.x;
            o_exts[g] = (uint8_t)(q.w & 0xffu);
            if (IS_SET) { o_msk[g] = q.w >> 8; if (o_setn) o_setn[g] = __popc(q.w >> 8); }
            else o_count[g] = (uint16_t)(q.w >> 8);
        }
    }
}

template <bool IS_SET>
__global__ void decode16_kernel(uint32_t n, const uint4* __restrict__ in, uint64_t* __restrict__ o_hi, uint64_t* __restrict__ o_lo,
                                uint8_t* __restrict__ exts, uint16_t* __restrict__ count, uint32_t* __restrict__ setn, uint32_t* __restrict__ msk) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 q = in[i];
    o_hi[i] = q.z; o_lo[i] = ((uint64_t)q.y << 32) | q.x;
    exts[i] = (uint8_t)(q.w & 0xffu);
    if (IS_SET) { msk[i] = q.w >> 8; if (setn) setn[i] = __popc(q.w >> 8); }
    else count[i] = (uint16_t)(q.w >> 8);
}

// ---- single-pass-per-digit scatter with decoupled look-back (round 3, DBG_ONESWEEP=1) ----------------------------------------
// The classic pass reads the records twice: radix16_hist (per-tile digit counts) -> scan -> radix16_scatter.  The global digit
// totals of ALL prefix passes do not depend on the order of the records, so one pass over the input gives them up front
// (radix16_global_hist_kernel); what a tile still lacks is the number of records with its digit in EARLIER tiles, and that it
// can get from its predecessors while it works (Merrill & Garland's decoupled look-back): a tile publishes its digit counts
// (AGGREGATE), walks back over the published counts of earlier tiles until it meets one that already knows its inclusive
// prefix, and publishes its own (INCLUSIVE).  Tile ids come from an atomic counter in start order, so every predecessor has
// started; waits are bounded (a tile that gives up raises the abort flag and the host repeats the pass the classic way).
constexpr int OS_MAXP = 4;
// status word = tag (4 bits) | count (60 bits); pass p uses the tags 2p + 1 (AGGREGATE) and 2p + 2 (INCLUSIVE), so that the words
// of an earlier pass read as "not published yet" and the array is cleared once per sort, not once per pass
constexpr unsigned long long OS_VAL = (1ull << 60) - 1;
__global__ void __launch_bounds__(512) radix16_global_hist_kernel(const uint4* __restrict__ in, uint32_t n, int s0, int npass, int key_bits,
                                                                   unsigned int* __restrict__ ghist) {
    __shared__ uint32_t h[OS_MAXP][256];
    for (int i = threadIdx.x; i < OS_MAXP * 256; i += blockDim.x) (&h[0][0])[i] = 0;
    __syncthreads();
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const uint4 r = in[e];
        for (int p = 0; p < npass; p++) {
            const int sft = s0 + 8 * p;
            const uint32_t mask = (1u << (key_bits - sft < 8 ? key_bits - sft : 8)) - 1u;
            atomicAdd(&h[p][digit16(r, sft) & mask], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < npass * 256; i += blockDim.x) { const uint32_t v = (&h[0][0])[i]; if (v) atomicAdd(&ghist[i], v); }
}
__global__ void __launch_bounds__(R16_THREADS) radix16_onesweep_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t n, int shift,
                                                                       uint32_t mask, const unsigned int* __restrict__ gexcl,
                                                                       unsigned long long* __restrict__ status, unsigned int* __restrict__ tile_counter,
                                                                       unsigned int* __restrict__ abort_flag, uint32_t pass) {
    const unsigned long long OS_AGG = (unsigned long long)(2 * pass + 1) << 60, OS_INC = (unsigned long long)(2 * pass + 2) << 60;
    __shared__ uint4 stage[R16_TILE];                   // 64 KB; the per-wave counters alias its first 8 KB
    __shared__ uint8_t s_dig[R16_TILE];
    __shared__ uint32_t s_gbase[256];
    __shared__ uint32_t s_ws[4];
    __shared__ uint32_t s_tile;
    uint32_t (*wc)[256] = reinterpret_cast<uint32_t (*)[256]>(stage);
    for (int i = threadIdx.x; i < R16_WAVES * 256; i += R16_THREADS) (&wc[0][0])[i] = 0;
    if (threadIdx.x == 0) s_tile = atomicAdd(tile_counter, 1u);          // tiles are numbered in the order they start
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t tile0 = tile * R16_TILE;
    const uint32_t base = tile0 + wave * (R16_ITEMS * 64) + lane;
    const uint32_t tile_n = n - tile0 < (uint32_t)R16_TILE ? n - tile0 : (uint32_t)R16_TILE;
    const uint64_t lt = lanemask_lt();
    uint4 rec[R16_ITEMS];
    uint32_t rank[R16_ITEMS];
    volatile uint32_t* mywc = &wc[wave][0];
#pragma unroll
    for (int r = 0; r < R16_ITEMS; r++) {
        const uint32_t e = base + r * 64;
        rec[r] = e < n ? in[e] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < R16_ITEMS; r++) {
        const bool valid = base + r * 64 < n;
        const uint32_t d = digit16(rec[r], shift) & mask;
        uint64_t same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const uint64_t bal = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? bal : ~bal;
        }
        const uint32_t before = __popcll(same & lt), cnt = __popcll(same);
        uint32_t wbase = 0;
        if (valid) wbase = mywc[d];
        if (valid && before == 0) mywc[d] = wbase + cnt;
        rank[r] = (d << 24) | (wbase + before);
    }
    __syncthreads();
    uint32_t tot = 0, run0 = 0;
    {
        if (tid < 256) {
#pragma unroll
            for (int w = 0; w < R16_WAVES; w++) tot += wc[w][tid];
            // publish this tile's count of digit `tid` right away: successors can look back past this tile while it works
            __hip_atomic_store(&status[(size_t)tile * 256 + tid], OS_AGG | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the word is the whole message: no other memory to order)
        }
        uint32_t incl = tot;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) { uint32_t o = __shfl_up(incl, dd); if (lane >= (uint32_t)dd) incl += o; }
        if (tid < 256 && lane == 63) s_ws[wave] = incl;
        __syncthreads();
        if (tid < 256) {
            uint32_t off = 0;
            for (uint32_t w = 0; w < wave; w++) off += s_ws[w];
            uint32_t run = off + incl - tot;
            run0 = run;
#pragma unroll
            for (int w = 0; w < R16_WAVES; w++) { uint32_t t = wc[w][tid]; wc[w][tid] = run; run += t; }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R16_ITEMS; r++) rank[r] = wc[wave][rank[r] >> 24] + (rank[r] & 0xffffffu);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R16_ITEMS; r++) {
        if (base + r * 64 < n) { stage[rank[r]] = rec[r]; s_dig[rank[r]] = (uint8_t)(digit16(rec[r], shift) & mask); }
    }
    // ---- look back: records with my digit in earlier tiles ----
    int failed = 0;
    if (tid < 256) {
        unsigned long long excl = 0;
        for (int64_t p = (int64_t)tile - 1; p >= 0; p--) {
            unsigned long long v = 0;
            uint32_t spins = 0;
            for (;;) {
                v = __hip_atomic_load(&status[(size_t)p * 256 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((v >> 60) == (OS_AGG >> 60) || (v >> 60) == (OS_INC >> 60)) break;
                if (++spins > (1u << 22) || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { failed = 1; break; }
            }
            if (failed) break;
            excl += v & OS_VAL;
            if ((v >> 60) == (OS_INC >> 60)) break;
        }
        if (!failed) {
            __hip_atomic_store(&status[(size_t)tile * 256 + tid], OS_INC | (excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_gbase[tid] = gexcl[tid] + (uint32_t)excl - run0;
        }
    }
    if (__syncthreads_or(failed)) { if (tid == 0) atomicOr(abort_flag, 1u); return; }
    for (uint32_t i = tid; i < tile_n; i += R16_THREADS) out[s_gbase[s_dig[i]] + i] = stage[i];
}

static int radix16_pass(dbg_ctx* ctx, const uint4* src, uint4* dst, uint32_t n, int shift, int bits, uint32_t* hist, uint32_t* hist_scanned, uint32_t nblocks) {
    const uint32_t mask = (1u << bits) - 1;
    ctx->t_begin("radix_hist", n);
    radix16_hist_kernel<<<nblocks, R16_THREADS, 0, ctx->stream>>>(src, n, shift, mask, hist, nblocks);
    ctx->t_end();
    LAUNCH_CHECK(ctx, "radix16_hist");
    DBG_TRY(scan_exclusive_u32(ctx, hist, hist_scanned, (uint64_t)256 * nblocks));
    ctx->t_begin("radix_scatter", n);
    radix16_scatter_kernel<<<nblocks, R16_THREADS, 0, ctx->stream>>>(src, dst, n, shift, mask, hist_scanned, nblocks);
    ctx->t_end();
    LAUNCH_CHECK(ctx, "radix16_scatter");
    return 0;
}
}  // namespace

// a = input (clobbered), b = scratch; key_bits <= 96
int sort_table_hybrid16(dbg_ctx* ctx, uint64_t n64, uint4* a, uint4* b, int key_bits, bool is_set, bool allow_hybrid,
                        uint64_t* o_hi, uint64_t* o_lo, uint8_t* o_exts, uint16_t* o_count, uint32_t* o_setn, uint32_t* o_msk) {
    if (n64 == 0) return 0;
    if (n64 >= (1ull << 32) - SS_CAP) return ctx->fail(110, "sort_table: more than 2^32-1 records in one call");
    const uint32_t n = (uint32_t)n64;
    const uint32_t nblocks = cdiv(n, R16_TILE);
    DBuf<uint32_t> hist, hist_scanned;
    ALLOC_OR_FAIL(ctx, hist, (size_t)256 * nblocks);
    ALLOC_OR_FAIL(ctx, hist_scanned, (size_t)256 * nblocks + 1);
    uint4 *src = a, *dst = b;
    int sorted_from = key_bits;                              // bits [sorted_from, key_bits) are in order
    if (allow_hybrid) {
        // Enough top bits that an average prefix group holds ~32 records, in 8-bit passes that start exactly at bit
        // key_bits - top_bits (a digit may straddle the 32-bit words of a record); the wave-per-group finisher orders such
        // groups with a counting sort on the next 6 key bits.  DBG_SORT=bytealigned: the round-1 form (digits on byte boundaries,
        // up to 7 more prefix bits -- a fourth pass at k = 47 -- and the walking finisher), which is also what runs when a
        // group is too long for a wave.
        int top_bits = 0;
        while (top_bits < key_bits && top_bits < 32 && (n64 >> top_bits) > 32) top_bits += 8;
        if (top_bits > key_bits) top_bits = key_bits;
        // A wave per group pays ~100 instructions per group: right for groups of dozens of records, wrong for small ones (n well
        // below 32 x 2^top_bits) -- those keep the round-1 form, whose extra prefix bits leave almost nothing to the finisher.
        const bool bytealigned = ctx->opt("DBG_SORT") && !strcmp(ctx->opt("DBG_SORT"), "bytealigned");
        const bool small_groups = (n64 >> top_bits) < 16;
        if (small_groups && (n64 >> top_bits) >= 2 && top_bits + 8 <= std::min(key_bits, 32)) top_bits += 8;   // the walking finisher wants (nearly) nothing left
        int s0 = key_bits - top_bits;
        if (bytealigned) { s0 = top_bits ? (s0 / 8) * 8 : key_bits; top_bits = key_bits - s0; }
        const int npass = (key_bits - s0 + 7) / 8;
        // look-back passes (default; DBG_ONESWEEP=0: the classic histogram + scatter passes): -2.1 ms per step at C2
        const bool onesweep = !(ctx->opt("DBG_ONESWEEP") && !strcmp(ctx->opt("DBG_ONESWEEP"), "0")) && npass >= 1 && npass <= OS_MAXP;
        DBuf<unsigned int> ghist, gexcl, os_ctl;
        DBuf<unsigned long long> os_status;
        if (onesweep) {
            ALLOC_OR_FAIL(ctx, ghist, OS_MAXP * 256); ALLOC_OR_FAIL(ctx, gexcl, OS_MAXP * 256); ALLOC_OR_FAIL(ctx, os_ctl, 2 * OS_MAXP);
            ALLOC_OR_FAIL(ctx, os_status, (size_t)nblocks * 256);
            HIP_TRY(ctx, hipMemsetAsync(ghist.p, 0, OS_MAXP * 256 * 4, ctx->stream));
            HIP_TRY(ctx, hipMemsetAsync(os_status.p, 0, (size_t)nblocks * 256 * 8, ctx->stream));
            HIP_TRY(ctx, hipMemsetAsync(os_ctl.p, 0, 2 * OS_MAXP * 4, ctx->stream));
            ctx->t_begin("radix_hist", n);
            radix16_global_hist_kernel<<<std::max<uint32_t>(1, std::min<uint32_t>(cdiv(n, 512 * 16), 2048)), 512, 0, ctx->stream>>>(src, n, s0, npass, key_bits, ghist.p);
            radix_global_scan_kernel<<<1, 256, 0, ctx->stream>>>(ghist.p, gexcl.p, npass);
            ctx->t_end();
            LAUNCH_CHECK(ctx, "radix16_global_hist");
        }
        int pi = 0;
        bool onesweep_broken = false;                        // after a pass gave up the status words are no longer trusted
        for (int s = s0; s < key_bits; s += 8, pi++) {
            bool done = false;
            if (onesweep && !onesweep_broken) {
                ctx->t_begin("radix_scatter", n);
                radix16_onesweep_kernel<<<nblocks, R16_THREADS, 0, ctx->stream>>>(src, dst, n, s, (1u << std::min(8, key_bits - s)) - 1u, gexcl.p + pi * 256,
                                                                                 os_status.p, os_ctl.p + 2 * pi, os_ctl.p + 2 * pi + 1, (uint32_t)pi);
                ctx->t_end();
                LAUNCH_CHECK(ctx, "radix16_onesweep");
                // the pass's input is only intact until the next pass writes over it: a pass that gave up is repeated now
                unsigned int ctl[2] = {0, 0};
                HIP_TRY(ctx, hipMemcpyAsync(ctl, os_ctl.p + 2 * pi, 8, hipMemcpyDeviceToHost, ctx->stream));
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                done = ctl[1] == 0;
                if (!done) onesweep_broken = true;
                if (!done && ctx->opt("DBG_DEBUG")) fprintf(stderr, "[sort] look-back pass gave up (shift %d): classic pass instead\n", s);
            }
            if (!done) DBG_TRY(radix16_pass(ctx, src, dst, n, s, std::min(8, key_bits - s), hist.p, hist_scanned.p, nblocks));
            std::swap(src, dst);
        }
        sorted_from = s0;
        DBuf<uint32_t> flags;
        ALLOC_OR_FAIL(ctx, flags, 1);
        const uint32_t nwg = cdiv(n, SS_WINDOW);
        const int dbits = std::min(6, key_bits - top_bits);
        for (int form = (bytealigned || dbits <= 0 || small_groups) ? 1 : 0; form < 2; form++) {       // 0: wave per group; 1: walking finisher
            HIP_TRY(ctx, hipMemsetAsync(flags.p, 0, 4, ctx->stream));
            ctx->t_begin("span_sort", n);
            if (form == 0) {
                if (is_set) span_sort16_groups_kernel<true><<<cdiv(n, SG_WINDOW), SS_THREADS, 0, ctx->stream>>>(src, n, key_bits, top_bits, dbits, o_hi, o_lo, o_exts, o_count, o_setn, o_msk, flags.p);
                else span_sort16_groups_kernel<false><<<cdiv(n, SG_WINDOW), SS_THREADS, 0, ctx->stream>>>(src, n, key_bits, top_bits, dbits, o_hi, o_lo, o_exts, o_count, o_setn, o_msk, flags.p);
            } else {
                if (is_set) span_sort16_kernel<true><<<nwg, SS_THREADS, 0, ctx->stream>>>(src, n, key_bits, top_bits, o_hi, o_lo, o_exts, o_count, o_setn, o_msk, flags.p);
                else span_sort16_kernel<false><<<nwg, SS_THREADS, 0, ctx->stream>>>(src, n, key_bits, top_bits, o_hi, o_lo, o_exts, o_count, o_setn, o_msk, flags.p);
            }
            ctx->t_end();
            LAUNCH_CHECK(ctx, "span_sort16");
            uint32_t fl = 0;
            HIP_TRY(ctx, hipMemcpyAsync(&fl, flags.p, 4, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (fl == 0) return 0;
            if (ctx->opt("DBG_DEBUG")) fprintf(stderr, "[sort] finisher form %d met a group of equal prefix that is too long for it (n=%u, top_bits=%d)\n", form, n, top_bits);
        }
    }
    // plain LSD sort over every key bit (stable passes compose, so the top-bit passes above need not be undone: a full
    // LSD from bit 0 re-sorts everything)
    (void)sorted_from;
    for (int s = 0; s < key_bits; s += 8) {
        DBG_TRY(radix16_pass(ctx, src, dst, n, s, std::min(8, key_bits - s), hist.p, hist_scanned.p, nblocks));
        std::swap(src, dst);
    }
    ctx->t_begin("finalize", n64);
    if (is_set) decode16_kernel<true><<<cdiv(n, 256), 256, 0, ctx->stream>>>(n, src, o_hi, o_lo, o_exts, nullptr, o_setn, o_msk);
    else decode16_kernel<false><<<cdiv(n, 256), 256, 0, ctx->stream>>>(n, src, o_hi, o_lo, o_exts, o_count, nullptr, nullptr);
    ctx->t_end();
    LAUNCH_CHECK(ctx, "decode16");
    return 0;
}
