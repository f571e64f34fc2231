// Dense path of filter_kmers (src/filter.rs:139-231) for short k-mers, 4 <= k <= 15.
//
// The super-k-mer fast path needs k >= 16 (its internal minimizers are 15 bases), and the generic sort-based path moves one
// 12-byte record per k-mer instance through several radix passes (measured 3 Gkmer/s).  For k <= 15 the whole key space --
// 4^k <= 2^30 canonical values -- fits in HBM as a directly addressed table: one 64-bit word per possible k-mer holding the count
// (40 bits) and the Exts byte, followed for CountFilterSet by a 64-bit label mask in the same 16-byte entry (17 GB at k = 15), so
// that a k-mer instance touches ONE cache line.  One wavefront walks a read 64 k-mers at a time
// (iter_kmer_exts + min_rc_flip + Exts::rc, lib.rs:812-841, :224-231, :729-748, exactly as the generic extractor) and does
// count += 1 / Exts |= e / mask |= 1 << label (CountFilter / CountFilterSet::summarize, filter.rs:53-62, :85-100) with global
// atomics: the returning add of the count brings the entry's Exts back, and an OR is only sent for Exts (or label) bits that are
// still missing, which after the first few reads is almost never.  The table index IS the key, so one compaction pass over the table emits the valid entries in ascending key order
// (filter.rs:205-230) -- no sort.  Very short k-mers (k <= 7: at most 16384 entries) are counted in per-workgroup LDS tables
// first, because a few thousand hot addresses would serialise the device's atomic units.
#include "dbg_internal.hpp"
#include <algorithm>
#include <cstring>
#include <vector>

int seq_max_label(dbg_ctx* c, const SeqDev& s, uint32_t* out);      // fastpath.hip

namespace {
constexpr int DENSE_LDS_K = 7;                  // k <= 7: 4^k x u32 fits a workgroup's LDS
constexpr int DENSE_RANGE_K = 8;                // k = 8: two key-range passes with 32768 counters in LDS each: 12 -> 67 Gkmer/s.  (k = 9 would take
                                                // eight passes: measured 157 ms against 154 for the global atomics -- no gain, not enabled)

__device__ __forceinline__ uint32_t load_label(const void* data, uint32_t width, uint64_t i) {
    if (width == 1) return ((const uint8_t*)data)[i];
    if (width == 2) return ((const uint16_t*)data)[i];
    return ((const uint32_t*)data)[i];
}

// LDS_COUNT: 0 = every count is a global atomic; 1 = the whole key space in this workgroup's LDS (k <= 7); 2 (round 4, k = 8) = the
// key space cut into ranges of DENSE_RANGE keys, blockIdx.y names the range: every range pass walks all reads and counts the k-mers
// of its range in 128 KB of LDS (one 1024-thread workgroup per CU) -- re-extracting a k-mer is ~45 vector instructions, an
// uncontended LDS atomic a few cycles, against a device atomic on one of 3*10^4..1.3*10^5 hot addresses.
constexpr uint32_t DENSE_RANGE_BITS = 15, DENSE_RANGE = 1u << DENSE_RANGE_BITS;
template <bool STRANDED, bool IS_SET, int LDS_COUNT>
__global__ void __launch_bounds__(LDS_COUNT == 2 ? 1024 : 256) dense_count_kernel(SeqDev s, int k, unsigned long long* __restrict__ tab) {
    constexpr uint32_t ES = IS_SET ? 2 : 1;                          // 64-bit words per entry: {count | Exts << 40} [, label mask]
    extern __shared__ uint32_t s_cnt[];                              // LDS_COUNT: counters of this workgroup (4^k, or one range)
    const uint32_t nkeys = LDS_COUNT == 2 ? DENSE_RANGE : 1u << (2 * k);
    const uint32_t range = LDS_COUNT == 2 ? blockIdx.y : 0u;
    if (LDS_COUNT) {
        for (uint32_t i = threadIdx.x; i < nkeys; i += blockDim.x) s_cnt[i] = 0;
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t si = wave; si < s.n; si += n_waves) {
        const uint32_t len = s.length[si];
        if (len < (uint32_t)k) continue;                                       // lib.rs:813
        const uint64_t st = s.start[si];
        const uint32_t nk = len - (uint32_t)k + 1;
        const uint32_t sexts = s.exts ? s.exts[si] : 0u;
        unsigned long long lbit = 0;
        if (IS_SET) lbit = 1ull << (s.data ? load_label(s.data, s.data_width, si) & 63u : 0u);
        for (uint32_t j = lane; j < nk; j += 64) {
            // the k-mer (at most 30 bits) and its right neighbour lie in two consecutive words; both are fetched unconditionally (index
            // clamped to the buffer); the left neighbour is in the first of them, or the last base of the word before when the k-mer starts a word
            const uint64_t o = st + j, wi = o >> 5;
            const uint32_t bo = (uint32_t)(o & 31);
            const int sft = (int)bo * 2;
            // (all three loads are issued unconditionally: a word load inside a divergent branch is what tools/micro/kmer_fetch.hip
            //  showed returning wrong words for whole wavefronts -- round 3; the selection is done on the loaded values)
            const uint64_t w0 = s.words[wi], w1 = s.words[wi + 1 < s.n_words ? wi + 1 : wi], wm = s.words[wi ? wi - 1 : 0];
            const uint64_t top = sft ? (w0 << sft) | (w1 >> (64 - sft)) : w0;
            K128 km{0ull, top >> (64 - 2 * k)};
            bool flip = false;
            if (!STRANDED) {
                const K128 rc = kmer_rc(km, k);
                if (!k128_lt(km, rc)) { km = rc; flip = true; }                  // ties flip (lib.rs:226-230)
            }
            const uint32_t key = (uint32_t)km.lo;
            if (LDS_COUNT == 2 && (key >> DENSE_RANGE_BITS) != range) continue;          // another range pass counts it
            // lib.rs:820-832: interior exts from the neighbouring bases, boundary exts from seq_exts
            auto base_at = [&](uint32_t q) -> uint32_t { return (uint32_t)((q < 32 ? w0 >> (62 - 2 * q) : w1 >> (62 - 2 * (q - 32))) & 3ull); };
            const uint32_t left = j == 0 ? (sexts & 0x0fu) : (1u << (bo ? base_at(bo - 1) : (uint32_t)(wm & 3ull)));
            const uint32_t right = (j + (uint32_t)k == len) ? (sexts & 0xf0u) : (16u << base_at(bo + (uint32_t)k));
            uint32_t ex = left | right;
            if (flip) ex = exts_rc(ex);
            unsigned long long* e = tab + (uint64_t)key * ES;
            unsigned long long old;
            if (LDS_COUNT) { atomicAdd(&s_cnt[LDS_COUNT == 2 ? key & (DENSE_RANGE - 1) : key], 1u); old = e[0]; }
            else old = atomicAdd(&e[0], 1ull);
            // (a stale value can only lack bits that are set by now: the OR is then sent needlessly, never skipped wrongly)
            if (((uint32_t)(old >> 40) & ex) != ex) atomicOr(&e[0], (unsigned long long)ex << 40);
            if (IS_SET && (e[1] & lbit) == 0) atomicOr(&e[1], lbit);
        }
    }
    if (LDS_COUNT) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nkeys; i += blockDim.x) {
            const uint32_t v = s_cnt[i];
            if (v) atomicAdd(&tab[(uint64_t)(LDS_COUNT == 2 ? range * DENSE_RANGE + i : i) * ES], (unsigned long long)v);
        }
    }
}

// valid / occupied entries per tile of 1024 table slots
constexpr uint32_t DT = 1024;
constexpr unsigned long long DCNT = (1ull << 40) - 1;            // count field of an entry
__device__ __forceinline__ bool dense_valid(bool is_set, unsigned long long c, uint64_t min_obs) {
    return c && (is_set ? c >= min_obs : (c > 65535ull ? 65535ull : c) >= min_obs);     // CountFilter compares its saturated u16 count (filter.rs:57-61)
}
template <bool IS_SET>
__global__ void __launch_bounds__(256) dense_tile_counts_kernel(const unsigned long long* __restrict__ tab,
                                                                uint32_t nkeys, uint64_t min_obs, uint32_t* __restrict__ n_valid,
                                                                uint32_t* __restrict__ n_all, uint32_t* __restrict__ n_lab) {
    constexpr uint32_t ES = IS_SET ? 2 : 1;
    __shared__ uint32_t s_v[4], s_a[4], s_l[4];
    uint32_t v = 0, a = 0, l = 0;
    for (uint32_t t = 0; t < DT / 256; t++) {
        const uint32_t i = blockIdx.x * DT + t * 256 + threadIdx.x;
        if (i < nkeys) {
            const unsigned long long c = tab[(uint64_t)i * ES] & DCNT;
            const bool ok = dense_valid(IS_SET, c, min_obs);
            a += c ? 1u : 0u; v += ok ? 1u : 0u;
            if (IS_SET && ok) l += (uint32_t)__popcll(tab[(uint64_t)i * ES + 1]);
        }
    }
    for (int d = 32; d; d >>= 1) { v += __shfl_xor(v, d); a += __shfl_xor(a, d); l += __shfl_xor(l, d); }
    if ((threadIdx.x & 63) == 0) { s_v[threadIdx.x >> 6] = v; s_a[threadIdx.x >> 6] = a; s_l[threadIdx.x >> 6] = l; }
    __syncthreads();
    if (threadIdx.x == 0) {
        n_valid[blockIdx.x] = s_v[0] + s_v[1] + s_v[2] + s_v[3];
        n_all[blockIdx.x] = s_a[0] + s_a[1] + s_a[2] + s_a[3];
        n_lab[blockIdx.x] = s_l[0] + s_l[1] + s_l[2] + s_l[3];
    }
}

// one wavefront per tile walks its 1024 slots in order, 64 at a time: ranks by ballot, output ascending by key
template <bool IS_SET>
__global__ void __launch_bounds__(64) dense_emit_kernel(const unsigned long long* __restrict__ tab, uint32_t nkeys, uint64_t min_obs,
                                                        const uint64_t* __restrict__ off_valid, const uint64_t* __restrict__ off_all,
                                                        const uint64_t* __restrict__ off_lab, uint32_t n_tiles,
                                                        uint64_t* __restrict__ key_hi, uint64_t* __restrict__ key_lo, uint8_t* __restrict__ exts,
                                                        uint16_t* __restrict__ count, uint64_t* __restrict__ set_off, uint32_t* __restrict__ set_val,
                                                        uint64_t* __restrict__ all_hi, uint64_t* __restrict__ all_lo) {
    constexpr uint32_t ES = IS_SET ? 2 : 1;
    const uint32_t lane = threadIdx.x;
    const uint64_t lt = lanemask_lt();
    uint64_t ov = off_valid[blockIdx.x], oa = off_all[blockIdx.x], ol = IS_SET ? off_lab[blockIdx.x] : 0;
    for (uint32_t t = 0; t < DT / 64; t++) {
        const uint32_t i = blockIdx.x * DT + t * 64 + lane;
        unsigned long long w = 0;
        if (i < nkeys) w = tab[(uint64_t)i * ES];
        const unsigned long long c = w & DCNT;
        const bool occ = c != 0;
        const bool ok = dense_valid(IS_SET, c, min_obs);
        const uint64_t bo = __ballot(occ), bv = __ballot(ok);
        if (all_lo && occ) { const uint64_t q = oa + (uint32_t)__popcll(bo & lt); all_lo[q] = i; if (all_hi) all_hi[q] = 0; }
        unsigned long long m = 0;
        uint32_t nl = 0;
        if (IS_SET && ok) { m = tab[(uint64_t)i * ES + 1]; nl = (uint32_t)__popcll(m); }
        uint32_t incl = nl;
        if (IS_SET) {
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= (uint32_t)d) incl += o; }
        }
        if (ok) {
            const uint64_t q = ov + (uint32_t)__popcll(bv & lt);
            key_lo[q] = i;
            if (key_hi) key_hi[q] = 0;
            exts[q] = (uint8_t)(w >> 40);
            if (IS_SET) {
                uint64_t o = ol + incl - nl;
                set_off[q] = o;
                while (m) { set_val[o++] = (uint32_t)__ffsll((long long)m) - 1u; m &= m - 1; }   // ascending = sort(); dedup() (filter.rs:97-98)
            } else count[q] = (uint16_t)(c > 65535ull ? 65535ull : c);                                  // saturating count (filter.rs:57)
        }
        ov += (uint32_t)__popcll(bv); oa += (uint32_t)__popcll(bo);
        if (IS_SET) ol += (uint32_t)__shfl((int)incl, 63);
    }
    if (IS_SET && blockIdx.x == n_tiles - 1 && lane == 0) set_off[ov] = ol;
}
}  // namespace

// returns 0 and sets *used when the dense path produced the table (4 <= k <= 15, labels < 64)
int filter_kmers_dense(dbg_ctx* c, const SeqDev& s, const dbg_filter_params* prm, uint64_t n_kmers, dbg_kmer_table* out, bool* used) {
    *used = false;
    const int k = (int)prm->k;
    if (k < 4 || k > 15) return 0;
    // The table is 4^k entries whatever the input: cleared and compacted in full (8.6 GB at k = 15, 17 GB with label masks).  A
    // call with few k-mers for its k would pay that for nothing (and push pooled memory out): it takes the generic path unless
    // DBG_PATH=dense insists.  The count field holds 40 bits: an input of 2^40 instances or more could carry into the Exts.
    const char* force = c->opt("DBG_PATH");
    const bool forced = force && !strcmp(force, "dense");
    if (n_kmers >= (1ull << 40)) return 0;
    if (!forced && n_kmers < ((1ull << (2 * k)) >> 6)) return 0;
    const bool is_set = prm->summarizer == DBG_COUNT_FILTER_SET, stranded = prm->stranded != 0, report_all = prm->report_all_kmers != 0;
    if (is_set && s.data) {
        uint32_t mx = 0;
        DBG_TRY(seq_max_label(c, s, &mx));
        if (mx >= 64) return 0;                                      // the label mask holds 64 labels: larger alphabets take the generic path
    }
    const uint32_t nkeys = 1u << (2 * k);
    const uint32_t n_tiles = cdiv(nkeys, DT);
    DBuf<uint32_t> t_valid, t_all, t_lab;
    DBuf<unsigned long long> tab;
    DBuf<uint64_t> o_valid, o_all, o_lab;
    const size_t tab_words = (size_t)nkeys * (is_set ? 2 : 1);
    if (!tab.alloc(c, tab_words)) return 0;                          // no room for the table (17 GB at k = 15 with label masks): the generic path takes the call
    ALLOC_OR_FAIL(c, t_valid, n_tiles); ALLOC_OR_FAIL(c, t_all, n_tiles); ALLOC_OR_FAIL(c, t_lab, n_tiles);
    ALLOC_OR_FAIL(c, o_valid, (size_t)n_tiles + 1); ALLOC_OR_FAIL(c, o_all, (size_t)n_tiles + 1); ALLOC_OR_FAIL(c, o_lab, (size_t)n_tiles + 1);
    HIP_TRY(c, hipMemsetAsync(tab.p, 0, tab_words * 8, c->stream));
    if (s.n && n_kmers) {
        const uint32_t blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((s.n + 3) / 4, 256ull * 16));
        const bool lds = k <= DENSE_LDS_K;
        // k = 8: key-range passes with the counts in LDS (DBG_DENSE_RANGES=0: the global-atomic form, for A/B measurements)
        const bool ranges = !lds && k <= DENSE_RANGE_K && !(c->opt("DBG_DENSE_RANGES") && !strcmp(c->opt("DBG_DENSE_RANGES"), "0"));
        const size_t shm = lds ? (size_t)nkeys * 4 : (ranges ? (size_t)DENSE_RANGE * 4 : 0);
        const dim3 grid = ranges ? dim3((uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((s.n + 15) / 16, 256ull)), nkeys >> DENSE_RANGE_BITS) : dim3(blocks);
        const uint32_t threads = ranges ? 1024u : 256u;
        c->t_begin("dense_count", n_kmers);
#define DL(ST, SET, LD) do { if (shm) HIP_TRY(c, hipFuncSetAttribute((const void*)dense_count_kernel<ST, SET, LD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)); \
        dense_count_kernel<ST, SET, LD><<<grid, threads, shm, c->stream>>>(s, k, tab.p); } while (0)
#define DGO(ST, SET) do { if (lds) DL(ST, SET, 1); else if (ranges) DL(ST, SET, 2); else DL(ST, SET, 0); } while (0)
        if (stranded) { if (is_set) DGO(true, true); else DGO(true, false); }
        else { if (is_set) DGO(false, true); else DGO(false, false); }
#undef DGO
#undef DL
        c->t_end();
        LAUNCH_CHECK(c, "dense_count");
    }
    c->t_begin("dense_compact", nkeys);
    if (is_set) dense_tile_counts_kernel<true><<<n_tiles, 256, 0, c->stream>>>(tab.p, nkeys, prm->min_kmer_obs, t_valid.p, t_all.p, t_lab.p);
    else dense_tile_counts_kernel<false><<<n_tiles, 256, 0, c->stream>>>(tab.p, nkeys, prm->min_kmer_obs, t_valid.p, t_all.p, t_lab.p);
    LAUNCH_CHECK(c, "dense_tile_counts");
    DBG_TRY(scan_exclusive_u32_u64(c, t_valid.p, o_valid.p, n_tiles));
    DBG_TRY(scan_exclusive_u32_u64(c, t_all.p, o_all.p, n_tiles));
    DBG_TRY(scan_exclusive_u32_u64(c, t_lab.p, o_lab.p, n_tiles));
    uint64_t n_valid = 0, n_all = 0, n_lab = 0;
    HIP_TRY(c, hipMemcpyAsync(&n_valid, o_valid.p + n_tiles, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(&n_all, o_all.p + n_tiles, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(&n_lab, o_lab.p + n_tiles, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    DBuf<uint64_t> k_hi, k_lo, a_hi, a_lo, set_off;
    DBuf<uint8_t> exts;
    DBuf<uint16_t> count;
    DBuf<uint32_t> set_val;
    const size_t nv = std::max<uint64_t>(n_valid, 1), na = std::max<uint64_t>(n_all, 1);
    ALLOC_OR_FAIL(c, k_hi, nv); ALLOC_OR_FAIL(c, k_lo, nv); ALLOC_OR_FAIL(c, exts, nv);
    if (is_set) { ALLOC_OR_FAIL(c, set_off, nv + 1); ALLOC_OR_FAIL(c, set_val, std::max<uint64_t>(n_lab, 1)); HIP_TRY(c, hipMemsetAsync(set_off.p, 0, 8, c->stream)); }
    else ALLOC_OR_FAIL(c, count, nv);
    if (report_all) { ALLOC_OR_FAIL(c, a_hi, na); ALLOC_OR_FAIL(c, a_lo, na); }
    if (is_set) dense_emit_kernel<true><<<n_tiles, 64, 0, c->stream>>>(tab.p, nkeys, prm->min_kmer_obs, o_valid.p, o_all.p, o_lab.p, n_tiles,
                                                                      k_hi.p, k_lo.p, exts.p, count.p, set_off.p, set_val.p, a_hi.p, a_lo.p);
    else dense_emit_kernel<false><<<n_tiles, 64, 0, c->stream>>>(tab.p, nkeys, prm->min_kmer_obs, o_valid.p, o_all.p, o_lab.p, n_tiles,
                                                                 k_hi.p, k_lo.p, exts.p, count.p, set_off.p, set_val.p, a_hi.p, a_lo.p);
    c->t_end();
    LAUNCH_CHECK(c, "dense_emit");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    memset(out, 0, sizeof(*out));
    out->n = n_valid;
    out->key_hi = k_hi.take(); out->key_lo = k_lo.take(); out->exts = exts.take(); out->count = count.take();
    out->set_off = set_off.take(); out->set_val = set_val.take(); out->n_set_val = is_set ? n_lab : 0;
    if (report_all) { out->n_all = n_all; out->all_hi = a_hi.take(); out->all_lo = a_lo.take(); }
    out->n_kmer_instances = n_kmers; out->n_passes = 1; out->on_device = 1;
    *used = true;
    return 0;
}
