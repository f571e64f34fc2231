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
constexpr int DENSE_RANGE_K = 8;                // k = 8 with DBG_DENSE_PART=0: two key-range passes with 32768 counters in LDS each: 12 -> 67 Gkmer/s (k = 9 would take
                                                // eight passes: measured 157 ms against 154 for the global atomics -- no gain).  By default k >= 8 partitions its instances (below).

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
__global__ void __launch_bounds__(LDS_COUNT == 2 ? 1024 : 256) dense_count_kernel(SeqDev s, uint64_t r0, uint64_t r1, int k, unsigned long long* __restrict__ tab) {
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
    for (uint64_t si = r0 + wave; si < r1; si += n_waves) {
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

// ------------------------------------------------------------------------------------------------
// 8 <= k <= 15 (round 4): PARTITION, then count in LDS.  A directly addressed table needs one device atomic per k-mer instance, and random
// device-scope atomics complete at 1.7-2.4*10^10 per second whatever is done to them (tools/micro/atomic_noret.hip): 13-23 Gkmer/s.
// Instead the instances -- {key bits | Exts [| label]} words of 4 (CountFilter) or 8 bytes (CountFilterSet) -- are first brought into the
// order of their key's top bits: final partition f = key >> 15 holds the instances of 32768 consecutive keys, which one workgroup then
// counts in 128 KB of LDS and adds to the table's entries f * 32768 ... (its own: no atomics).  The partitioning is a counting sort in
// at most two levels of <= 256 parts: (A) one pass over the reads counts the instances of every final partition (LDS histogram per
// workgroup, flushed once) -> exact offsets; (B) a second pass extracts the instances and scatters them by the top <= 8 key bits:
// a wavefront collects 2048 instances in LDS, ranks them within their part with LDS atomics, reserves every part's stretch of the
// tile with ONE global atomic and writes the instances there; (C, k >= 12) the same again inside every level-1 part on the next <= 7
// bits; (D) count.  Exts and labels are ORed into the table entry only where bits are missing, as in the kernels above (the entries of a
// partition are 256-512 KB: cache-resident while its workgroup runs).  Reads are taken in batches that bound the two instance buffers.
// ------------------------------------------------------------------------------------------------
#ifndef DBG_PART_TILE1
#define DBG_PART_TILE1 512
#endif
#ifndef DBG_PART_TILE2
#define DBG_PART_TILE2 4096
#endif
constexpr uint32_t PART_TILE1 = DBG_PART_TILE1, PART_TILE2 = DBG_PART_TILE2;     // instances a wavefront collects before it writes them out (level 1 / level 2)
template <bool IS_SET> struct DenseInst { typedef uint32_t type; };
template <> struct DenseInst<true> { typedef unsigned long long type; };
// instance word: CountFilter  u32 = key bits (< 2^24) | Exts << 24;  CountFilterSet  u64 = key bits | Exts << 32 | label << 40
template <bool IS_SET> __device__ __forceinline__ typename DenseInst<IS_SET>::type dense_inst_make(uint32_t keybits, uint32_t ex, uint32_t label) {
    if (IS_SET) return (typename DenseInst<IS_SET>::type)((unsigned long long)keybits | ((unsigned long long)ex << 32) | ((unsigned long long)label << 40));
    return (typename DenseInst<IS_SET>::type)(keybits | (ex << 24));
}
__device__ __forceinline__ uint32_t dense_inst_key(uint32_t w) { return w & 0xffffffu; }
__device__ __forceinline__ uint32_t dense_inst_key(unsigned long long w) { return (uint32_t)w; }
__device__ __forceinline__ uint32_t dense_inst_ex(uint32_t w) { return w >> 24; }
__device__ __forceinline__ uint32_t dense_inst_ex(unsigned long long w) { return (uint32_t)(w >> 32) & 0xffu; }
// CountFilterSet with ONE partition level (k <= 11): only 15 key bits are left, so the label rides in the 4-byte word too: key | label << 16 | Exts << 24
__device__ __forceinline__ uint32_t dense_inst_label(uint32_t w) { return (w >> 16) & 63u; }
__device__ __forceinline__ uint32_t dense_inst_label(unsigned long long w) { return (uint32_t)(w >> 40) & 63u; }

// k-mer j of a read (iter_kmer_exts + min_rc_flip + Exts::rc, as in dense_count_kernel): canonical key and, when WITH_EX, its Exts
template <bool STRANDED, bool WITH_EX>
__device__ __forceinline__ void dense_kmer_at(const SeqDev& s, int k, uint64_t st, uint32_t len, uint32_t sexts, uint32_t j, uint32_t* key_out, uint32_t* ex_out) {
    const uint64_t o = st + j, wi = o >> 5;
    const uint32_t bo = (uint32_t)(o & 31);
    const int sft = (int)bo * 2;
    // (unconditional loads, indices clamped: see dense_count_kernel)
    const uint64_t w0 = s.words[wi], w1 = s.words[wi + 1 < s.n_words ? wi + 1 : wi], wm = s.words[wi ? wi - 1 : 0];
    const uint64_t top = sft ? (w0 << sft) | (w1 >> (64 - sft)) : w0;
    K128 km{0ull, top >> (64 - 2 * k)};
    bool flip = false;
    if (!STRANDED) {
        const K128 rc = kmer_rc(km, k);
        if (!k128_lt(km, rc)) { km = rc; flip = true; }
    }
    *key_out = (uint32_t)km.lo;
    if (WITH_EX) {
        auto base_at = [&](uint32_t q) -> uint32_t { return (uint32_t)((q < 32 ? w0 >> (62 - 2 * q) : w1 >> (62 - 2 * (q - 32))) & 3ull); };
        const uint32_t left = j == 0 ? (sexts & 0x0fu) : (1u << (bo ? base_at(bo - 1) : (uint32_t)(wm & 3ull)));
        const uint32_t right = (j + (uint32_t)k == len) ? (sexts & 0xf0u) : (16u << base_at(bo + (uint32_t)k));
        uint32_t ex = left | right;
        if (flip) ex = exts_rc(ex);
        *ex_out = ex;
    }
}

// (A) one pass over the reads [r0, r1) counts, for every wavefront w of the grid and every level-1 part p, the instances that wavefront w of
// pass (B) -- which takes the same reads: read r0 + w, r0 + w + n_waves, ... -- will write into part p: wave_cnt[p * n_waves + w].  Scanned in
// that order they are every wavefront's exact offsets: pass (B) needs no shared cursor.  With a second level, the instances of every final
// partition are counted too (one LDS histogram per workgroup, flushed once).
template <bool STRANDED>
__global__ void __launch_bounds__(1024) dense_part_hist_kernel(SeqDev s, uint64_t r0, uint64_t r1, int k, uint32_t shift1, uint32_t p1n, uint32_t n_final /* 0: one level */,
                                                               uint32_t* __restrict__ wave_cnt, unsigned long long* __restrict__ fine_cnt) {
    extern __shared__ uint32_t s_h[];                                // 16 waves x p1n, then n_final
    uint32_t* const s_fine = s_h + 16 * p1n;
    for (uint32_t i = threadIdx.x; i < 16 * p1n + n_final; i += blockDim.x) s_h[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    uint32_t* const mine = s_h + (threadIdx.x >> 6) * p1n;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    uint64_t si = r0 + wave;
    uint32_t len_n = 0;
    uint64_t st_n = 0;
    if (si < r1) { len_n = s.length[si]; st_n = s.start[si]; }
    for (; si < r1; si += n_waves) {
        const uint32_t len = len_n;
        const uint64_t st = st_n;
        { const uint64_t nx = si + n_waves < r1 ? si + n_waves : si; len_n = s.length[nx]; st_n = s.start[nx]; }   // the next read's, requested early
        if (len < (uint32_t)k) continue;
        const uint32_t nk = len - (uint32_t)k + 1;
        for (uint32_t j0 = lane; j0 < nk; j0 += 192) {                 // (three windows of 64 k-mers: their word loads are in flight together)
            uint32_t key[3], ex;
#pragma unroll
            for (int u = 0; u < 3; u++) dense_kmer_at<STRANDED, false>(s, k, st, len, 0u, j0 + 64u * u < nk ? j0 + 64u * u : 0u, &key[u], &ex);
#pragma unroll
            for (int u = 0; u < 3; u++) {
                if (j0 + 64u * u >= nk) continue;
                atomicAdd(&mine[key[u] >> shift1], 1u);
                if (n_final) atomicAdd(&s_fine[key[u] >> DENSE_RANGE_BITS], 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t p = lane; p < p1n; p += 64) wave_cnt[(uint64_t)p * n_waves + wave] = mine[p];
    for (uint32_t i = threadIdx.x; i < n_final; i += blockDim.x) { const uint32_t v = s_fine[i]; if (v) atomicAdd(&fine_cnt[i], (unsigned long long)v); }
}
// one level: final partition p starts where its first wavefront's stretch starts
__global__ void dense_part_fineoff_kernel(const uint64_t* __restrict__ wave_off, uint64_t n_waves, uint32_t p1n, uint64_t* __restrict__ fine_off) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p <= p1n) fine_off[p] = wave_off[(uint64_t)p * n_waves];
}

// a wavefront's tile: instances with their part; the tile's instances per part are counted as they arrive.  Offsets are 32-bit: a batch
// holds fewer than 2^32 instances (the host sees to it).  LDS per wavefront decides how many wavefronts a CU holds, and that decides the
// speed of these latency-bound kernels: 7 KB with 512 four-byte instances and 128 parts.
template <class INST, uint32_t TILE, uint32_t NP>
struct PartTile {
    INST raw[TILE], sorted[TILE];
    uint8_t part[TILE], spart[TILE];
    uint32_t hist[NP], loc[NP];
    uint32_t base[NP];                                               // PRIVATE: where this wavefront's next instance of the part goes
    uint32_t delta[NP];                                              // sorted position q of the tile -> out[delta[part] + q]   (mod 2^32)
};
// The tile is brought into part order in LDS (counting sort: the parts' counts are known, an instance's place within its part is drawn
// from the part's LDS counter), every part's stretch of the output is set aside -- from the wavefront's own running offsets when they
// were computed beforehand (PRIVATE: level 1, where <= 256 cursors shared by every wavefront would be 256 hot addresses: same-address
// device atomics complete at ~10^8/s), else with one global atomic per part -- and the sorted tile leaves with neighbouring lanes
// writing neighbouring words (writing each instance straight to its place cost 17 of 30 ms at k = 11: 64 separate 4-byte stores per
// instruction).
template <bool PRIVATE, class INST, uint32_t TILE, uint32_t NP>
__device__ __forceinline__ void part_tile_flush(PartTile<INST, TILE, NP>& t, uint32_t fill, uint32_t n_parts, unsigned long long* __restrict__ cursor, INST* __restrict__ out) {
    const uint32_t lane = threadIdx.x;
    constexpr int PER = NP / 64;
    __syncthreads();                                                 // (one wavefront per workgroup: orders the LDS traffic)
    // exclusive offsets of the parts inside the tile: lane l owns parts PER * l .. PER * l + PER - 1
    {
        uint32_t h[PER], sum = 0;
#pragma unroll
        for (int q = 0; q < PER; q++) { h[q] = PER * lane + q < n_parts ? t.hist[PER * lane + q] : 0u; sum += h[q]; }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= (uint32_t)d) incl += o; }
        uint32_t run = incl - sum;
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const uint32_t p = PER * lane + q;
            if (p < n_parts) {
                t.loc[p] = run;
                uint32_t at = 0;
                if (PRIVATE) { at = t.base[p]; t.base[p] = at + h[q]; }
                else if (h[q]) at = (uint32_t)atomicAdd(&cursor[p], (unsigned long long)h[q]);
                t.delta[p] = at - run;
                t.hist[p] = 0;
            }
            run += h[q];
        }
    }
    __syncthreads();
    for (uint32_t i = lane; i < fill; i += 64) {
        const uint32_t p = t.part[i];
        const uint32_t q = t.loc[p] + atomicAdd(&t.hist[p], 1u);
        t.sorted[q] = t.raw[i];
        t.spart[q] = (uint8_t)p;
    }
    __syncthreads();
    for (uint32_t q = lane; q < fill; q += 64) out[(uint32_t)(t.delta[t.spart[q]] + q)] = t.sorted[q];
    for (uint32_t p = lane; p < n_parts; p += 64) t.hist[p] = 0;
    __syncthreads();
}

// (B) level 1: wavefront blockIdx.x extracts the instances of its reads and scatters them by key >> shift1 into its own stretches
// (wave_off, from pass A); the instance keeps the key bits below
template <bool STRANDED, bool IS_SET, uint32_t NP, bool COMPACT = false>
__global__ void __launch_bounds__(64) dense_part_scatter1_kernel(SeqDev s, uint64_t r0, uint64_t r1, int k, uint32_t shift1, uint32_t n_parts,
                                                                 const uint64_t* __restrict__ wave_off, typename DenseInst<IS_SET && !COMPACT>::type* __restrict__ out) {
    typedef typename DenseInst<IS_SET && !COMPACT>::type INST;
    __shared__ PartTile<INST, PART_TILE1, NP> t;
    const uint32_t lane = threadIdx.x;
    const uint64_t lt = lanemask_lt();
    for (uint32_t p = lane; p < NP; p += 64) { t.hist[p] = 0; t.base[p] = p < n_parts ? (uint32_t)wave_off[(uint64_t)p * gridDim.x + blockIdx.x] : 0u; }
    __syncthreads();
    uint32_t fill = 0;                                               // wave-uniform
    const uint32_t keep = (1u << shift1) - 1u;
    // a read's length / start / Exts / label are a chain of dependent loads in front of its words: the next read's are requested while
    // this one is processed (unconditionally, index clamped -- see dense_count_kernel)
    auto meta_of = [&](uint64_t si, uint32_t& len, uint64_t& st, uint32_t& sexts, uint32_t& label) {
        len = s.length[si]; st = s.start[si];
        sexts = s.exts ? s.exts[si] : 0u;
        label = (IS_SET && s.data) ? load_label(s.data, s.data_width, si) & 63u : 0u;
    };
    uint64_t si = r0 + blockIdx.x;
    uint32_t len_n = 0, sexts_n = 0, label_n = 0;
    uint64_t st_n = 0;
    if (si < r1) meta_of(si, len_n, st_n, sexts_n, label_n);
    for (; si < r1; si += gridDim.x) {
        const uint32_t len = len_n, sexts = sexts_n, label = label_n;
        const uint64_t st = st_n;
        meta_of(si + gridDim.x < r1 ? si + gridDim.x : si, len_n, st_n, sexts_n, label_n);
        if (len < (uint32_t)k) continue;
        const uint32_t nk = len - (uint32_t)k + 1;
        // three windows of 64 k-mers per step (a 150-base read in one): their word loads are all in flight before the first LDS access
        for (uint32_t j0 = 0; j0 < nk; j0 += 192) {
            if (fill + 192 > PART_TILE1) { part_tile_flush<true>(t, fill, n_parts, nullptr, out); fill = 0; }
            uint32_t key[3], ex[3];
            bool on[3];
#pragma unroll
            for (int u = 0; u < 3; u++) {
                const uint32_t j = j0 + 64u * u + lane;
                on[u] = j < nk;
                dense_kmer_at<STRANDED, true>(s, k, st, len, sexts, on[u] ? j : 0u, &key[u], &ex[u]);
            }
#pragma unroll
            for (int u = 0; u < 3; u++) {
                const uint64_t bal = __ballot(on[u]);
                if (on[u]) {
                    const uint32_t pos = fill + (uint32_t)__popcll(bal & lt), p = key[u] >> shift1;
                    t.raw[pos] = COMPACT ? (INST)((key[u] & keep) | (label << 16) | (ex[u] << 24)) : (INST)dense_inst_make<IS_SET && !COMPACT>(key[u] & keep, ex[u], label);
                    t.part[pos] = (uint8_t)p;
                    atomicAdd(&t.hist[p], 1u);
                }
                fill += (uint32_t)__popcll(bal);
            }
        }
    }
    if (fill) part_tile_flush<true>(t, fill, n_parts, nullptr, out);
}

// ---- level 1, second form (DBG_DENSE_RAW, the default): extract once into a raw buffer, then scatter uniform tiles ----
// The kernels above walk reads inside the scatter: one wavefront per workgroup (reads differ in length), 512-instance tiles, three dependent
// phases per tile -- latency-bound.  Here the instances are first EXTRACTED in read order (a read's instances are consecutive: offsets from a
// scan of the reads' k-mer counts; every store coalesced) into {key, Exts [| label << 8]} arrays, one workgroup per block of 512 consecutive
// reads, which also counts the block's instances per level-1 part; scanned in (part, block) order these are every block's exact offsets.  The
// scatter then is the level-2 kernel's twin: a 256-thread workgroup per block takes its stretch of the raw arrays in tiles of 4096, 16 loads
// in flight per thread, counting sort in LDS, write-out behind the block's own running offsets.
constexpr uint32_t DENSE_BLK = 512;              // reads per block
template <bool IS_SET> struct DenseAux { typedef uint8_t type; };
template <> struct DenseAux<true> { typedef uint16_t type; };
__global__ void dense_nk_kernel(SeqDev s, uint64_t r0, uint64_t r1, int k, uint32_t* __restrict__ nk) {
    const uint64_t i = r0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < r1) { const uint32_t len = s.length[i]; nk[i - r0] = len >= (uint32_t)k ? len - (uint32_t)k + 1u : 0u; }
}
template <bool STRANDED, bool IS_SET>
__global__ void __launch_bounds__(256) dense_extract_kernel(SeqDev s, uint64_t r0, uint64_t r1, int k, uint32_t shift1, uint32_t p1n, const uint64_t* __restrict__ inst_off,
                                                            uint32_t n_blocks, uint32_t* __restrict__ raw_key, typename DenseAux<IS_SET>::type* __restrict__ raw_aux,
                                                            uint32_t* __restrict__ blk_cnt) {
    typedef typename DenseAux<IS_SET>::type AUX;
    __shared__ uint32_t s_h[4][128];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (uint32_t i = tid; i < 4 * 128; i += 256) (&s_h[0][0])[i] = 0;
    __syncthreads();
    const uint64_t rb = r0 + (uint64_t)blockIdx.x * DENSE_BLK, re = rb + DENSE_BLK < r1 ? rb + DENSE_BLK : r1;
    // a read's length / start / Exts / label / offset are a chain of dependent loads in front of its words: the next read's are requested
    // while this one is processed (unconditionally, index clamped -- see dense_count_kernel)
    auto meta_of = [&](uint64_t si, uint32_t& len, uint64_t& st, uint32_t& sexts, uint32_t& label, uint64_t& off) {
        len = s.length[si]; st = s.start[si]; off = inst_off[si - r0];
        sexts = s.exts ? s.exts[si] : 0u;
        label = (IS_SET && s.data) ? load_label(s.data, s.data_width, si) & 63u : 0u;
    };
    uint64_t si = rb + wv;
    uint32_t len_n = 0, sexts_n = 0, label_n = 0;
    uint64_t st_n = 0, off_n = 0;
    if (si < re) meta_of(si, len_n, st_n, sexts_n, label_n, off_n);
    for (; si < re; si += 4) {
        const uint32_t len = len_n, sexts = sexts_n, label = label_n;
        const uint64_t st = st_n, off = off_n;
        meta_of(si + 4 < re ? si + 4 : si, len_n, st_n, sexts_n, label_n, off_n);
        if (len < (uint32_t)k) continue;
        const uint32_t nk = len - (uint32_t)k + 1;
        for (uint32_t j0 = lane; j0 < nk; j0 += 192) {                 // three windows of 64 k-mers: their word loads are in flight together
            uint32_t key[3], ex[3];
#pragma unroll
            for (int u = 0; u < 3; u++) dense_kmer_at<STRANDED, true>(s, k, st, len, sexts, j0 + 64u * u < nk ? j0 + 64u * u : 0u, &key[u], &ex[u]);
#pragma unroll
            for (int u = 0; u < 3; u++) {
                const uint32_t j = j0 + 64u * u;
                if (j >= nk) continue;
                raw_key[off + j] = key[u];
                raw_aux[off + j] = (AUX)(ex[u] | (IS_SET ? label << 8 : 0u));
                if (p1n > 4) atomicAdd(&s_h[wv][key[u] >> shift1], 1u);
            }
            if (p1n <= 4) {                                              // two or four parts (k = 8): 64 lanes on 2-4 LDS addresses serialise -- count by ballot
#pragma unroll
                for (int u = 0; u < 3; u++) {
                    const bool on = j0 + 64u * u < nk;
                    for (uint32_t p = 0; p < p1n; p++) {
                        const uint32_t c = (uint32_t)__popcll(__ballot(on && (key[u] >> shift1) == p));
                        if (lane == 0 && c) s_h[wv][p] += c;
                    }
                }
            }
        }
    }
    __syncthreads();
    if (tid < p1n) blk_cnt[(uint64_t)tid * n_blocks + blockIdx.x] = s_h[0][tid] + s_h[1][tid] + s_h[2][tid] + s_h[3][tid];
}
template <bool IS_SET, bool COMPACT>
__global__ void __launch_bounds__(256) dense_part_scatter1t_kernel(const uint32_t* __restrict__ raw_key, const typename DenseAux<IS_SET>::type* __restrict__ raw_aux,
                                                                   const uint64_t* __restrict__ inst_off, uint64_t nr, uint32_t shift1, uint32_t n_parts,
                                                                   const uint64_t* __restrict__ blk_off, uint32_t n_blocks,
                                                                   typename DenseInst<IS_SET && !COMPACT>::type* __restrict__ out) {
    typedef typename DenseInst<IS_SET && !COMPACT>::type INST;
    constexpr uint32_t TILE = PART_TILE2 / (sizeof(INST) == 8 ? 2 : 1), PER = TILE / 256;
    __shared__ INST raw[TILE], sorted[TILE];
    __shared__ uint8_t part[TILE], spart[TILE];
    __shared__ uint32_t hist[128], loc[128], delta[128], base[128];
    const uint32_t tid = threadIdx.x, b = blockIdx.x;
    if (tid < 128) { hist[tid] = 0; base[tid] = tid < n_parts ? (uint32_t)blk_off[(uint64_t)tid * n_blocks + b] : 0u; }
    const uint64_t rb = (uint64_t)b * DENSE_BLK, re = rb + DENSE_BLK < nr ? rb + DENSE_BLK : nr;
    const uint64_t beg = inst_off[rb], end = inst_off[re];
    const uint32_t keep = (1u << shift1) - 1u;
    __syncthreads();
    for (uint64_t t0 = beg; t0 < end; t0 += TILE) {
        const uint32_t fill = (uint32_t)(end - t0 < TILE ? end - t0 : TILE);
        uint32_t kk[PER], aa[PER];
#pragma unroll
        for (uint32_t u = 0; u < PER; u++) { const uint32_t i = tid + 256u * u; const uint64_t g = t0 + (i < fill ? i : fill - 1); kk[u] = raw_key[g]; aa[u] = raw_aux[g]; }
#pragma unroll
        for (uint32_t u = 0; u < PER; u++) {
            const uint32_t i = tid + 256u * u;
            const uint32_t p = kk[u] >> shift1, ex = aa[u] & 0xffu, label = aa[u] >> 8;
            if (i < fill) {
                raw[i] = COMPACT ? (INST)((kk[u] & keep) | (label << 16) | (ex << 24)) : (INST)dense_inst_make<IS_SET && !COMPACT>(kk[u] & keep, ex, label);
                part[i] = (uint8_t)p;
                if (n_parts > 4) atomicAdd(&hist[p], 1u);
            }
            if (n_parts <= 4) {                                          // (k = 8: see dense_extract_kernel)
                for (uint32_t q = 0; q < n_parts; q++) {
                    const uint32_t c = (uint32_t)__popcll(__ballot(i < fill && p == q));
                    if ((tid & 63u) == 0 && c) atomicAdd(&hist[q], c);
                }
            }
        }
        __syncthreads();
        if (tid < 64) {                                              // exclusive offsets of the parts inside the tile; their stretches behind the block's running offsets
            const uint32_t h0 = hist[2 * tid], h1 = hist[2 * tid + 1];
            uint32_t incl = h0 + h1;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (tid >= (uint32_t)d) incl += o; }
            const uint32_t q0 = incl - h0 - h1, q1 = q0 + h0;
            loc[2 * tid] = q0; loc[2 * tid + 1] = q1;
            const uint32_t b0 = base[2 * tid], b1 = base[2 * tid + 1];
            delta[2 * tid] = b0 - q0; delta[2 * tid + 1] = b1 - q1;
            base[2 * tid] = b0 + h0; base[2 * tid + 1] = b1 + h1;
            hist[2 * tid] = 0; hist[2 * tid + 1] = 0;
        }
        __syncthreads();
        if (n_parts > 4) {
#pragma unroll
            for (uint32_t u = 0; u < PER; u++) {
                const uint32_t i = tid + 256u * u;
                if (i < fill) {
                    const uint32_t p = part[i];
                    const uint32_t q = loc[p] + atomicAdd(&hist[p], 1u);
                    sorted[q] = raw[i];
                    spart[q] = (uint8_t)p;
                }
            }
        } else {
            // few parts: a wavefront's places come from ballots -- its members of part pp over all its rows are counted first, ONE returning atomic
            // per part sets their stretch aside, and the ballots, taken again, give every member its place in it
            const uint64_t lt = lanemask_lt();
            uint32_t pv[PER];
#pragma unroll
            for (uint32_t u = 0; u < PER; u++) { const uint32_t i = tid + 256u * u; pv[u] = i < fill ? part[i] : 0xffu; }
            for (uint32_t pp = 0; pp < n_parts; pp++) {
                uint32_t tot = 0;
#pragma unroll
                for (uint32_t u = 0; u < PER; u++) tot += (uint32_t)__popcll(__ballot(pv[u] == pp));
                uint32_t b = 0;
                if ((tid & 63u) == 0 && tot) b = atomicAdd(&hist[pp], tot);
                b = __shfl(b, 0) + loc[pp];
#pragma unroll
                for (uint32_t u = 0; u < PER; u++) {
                    const uint64_t m = __ballot(pv[u] == pp);
                    if (pv[u] == pp) {
                        const uint32_t i = tid + 256u * u, q = b + (uint32_t)__popcll(m & lt);
                        sorted[q] = raw[i];
                        spart[q] = (uint8_t)pp;
                    }
                    b += (uint32_t)__popcll(m);
                }
            }
        }
        __syncthreads();
        for (uint32_t q = tid; q < fill; q += 256) out[(uint32_t)(delta[spart[q]] + q)] = sorted[q];
        if (tid < 128) hist[tid] = 0;
        __syncthreads();
    }
}
// two levels: instances per final partition, counted over the level-1 output (part p1 = blockIdx.y lies at [part_off[p1 * stride], part_off[(p1 + 1) * stride]))
template <class INST>
__global__ void __launch_bounds__(256) dense_fine_hist_kernel(const INST* __restrict__ in, const uint64_t* __restrict__ part_off, uint64_t stride, uint32_t l2,
                                                              unsigned long long* __restrict__ fine_cnt) {
    __shared__ uint32_t hist[256];
    const uint32_t tid = threadIdx.x, p1 = blockIdx.y, n_sub = 1u << l2;
    hist[tid] = 0;
    __syncthreads();
    const uint64_t beg = part_off[(uint64_t)p1 * stride], end = part_off[(uint64_t)(p1 + 1) * stride];
    for (uint64_t i0 = beg + (uint64_t)blockIdx.x * 2048 + tid; i0 < end; i0 += (uint64_t)gridDim.x * 2048) {
        INST w[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const uint64_t i = i0 + 256ull * u; w[u] = in[i < end ? i : end - 1]; }
#pragma unroll
        for (int u = 0; u < 8; u++) if (i0 + 256ull * u < end) atomicAdd(&hist[(dense_inst_key(w[u]) >> DENSE_RANGE_BITS) & (n_sub - 1u)], 1u);
    }
    __syncthreads();
    if (tid < n_sub && hist[tid]) atomicAdd(&fine_cnt[((uint64_t)p1 << l2) | tid], (unsigned long long)hist[tid]);
}

// (C) level 2: the instances of level-1 part blockIdx.y, scattered by their next key bits (those above the 15 that name the LDS counter).
// The work is uniform here (no reads to walk), so a 256-thread workgroup shares one tile: 4096 four-byte (2048 eight-byte) instances are
// loaded with 16 (8) independent loads per thread, brought into sub-part order in LDS, every sub-part's stretch reserved with one global
// atomic per tile, and written out with neighbouring threads on neighbouring words (runs of >= 32 instances at 128 sub-parts).
// (CountFilterSet: 8-byte instances in, 4-byte ones out -- after this level only 15 key bits are left and the label fits: key | label << 16 | Exts << 24)
template <bool IS_SET>
__global__ void __launch_bounds__(256) dense_part_scatter2_kernel(const typename DenseInst<IS_SET>::type* __restrict__ in, const uint64_t* __restrict__ fine_off,
                                                                  uint32_t l2, unsigned long long* __restrict__ cursor2, uint32_t* __restrict__ out) {
    typedef typename DenseInst<IS_SET>::type INST;
    constexpr uint32_t TILE = PART_TILE2 / (IS_SET ? 2 : 1), PER = TILE / 256;
    __shared__ uint32_t raw[TILE], sorted[TILE];
    __shared__ uint8_t part[TILE], spart[TILE];
    __shared__ uint32_t hist[256], loc[256], delta[256];
    const uint32_t tid = threadIdx.x, p1 = blockIdx.y, n_sub = 1u << l2;          // n_sub <= 256: at most 8 bits are left for the second level
    hist[tid] = 0;
    const uint64_t beg = fine_off[(uint64_t)p1 << l2], end = fine_off[(uint64_t)(p1 + 1) << l2];
    unsigned long long* const cur = cursor2 + ((uint64_t)p1 << l2);
    __syncthreads();
    for (uint64_t t0 = beg + (uint64_t)blockIdx.x * TILE; t0 < end; t0 += (uint64_t)gridDim.x * TILE) {
        const uint32_t fill = (uint32_t)(end - t0 < TILE ? end - t0 : TILE);
        INST w[PER];
#pragma unroll
        for (uint32_t u = 0; u < PER; u++) { const uint32_t i = tid + 256u * u; w[u] = in[t0 + (i < fill ? i : fill - 1)]; }
#pragma unroll
        for (uint32_t u = 0; u < PER; u++) {
            const uint32_t i = tid + 256u * u;
            if (i < fill) {
                const uint32_t sub = (dense_inst_key(w[u]) >> DENSE_RANGE_BITS) & (n_sub - 1u);
                raw[i] = IS_SET ? (dense_inst_key(w[u]) & (DENSE_RANGE - 1)) | (dense_inst_label(w[u]) << 16) | (dense_inst_ex(w[u]) << 24) : (uint32_t)w[u];
                part[i] = (uint8_t)sub;
                atomicAdd(&hist[sub], 1u);
            }
        }
        __syncthreads();
        if (tid < 64) {                                              // exclusive offsets of the sub-parts inside the tile; their stretches of the output
            uint32_t h[4], sum = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) { h[q] = hist[4 * tid + q]; sum += h[q]; }
            uint32_t incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (tid >= (uint32_t)d) incl += o; }
            uint32_t run = incl - sum;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t p = 4 * tid + q;
                loc[p] = run;
                delta[p] = (h[q] ? (uint32_t)atomicAdd(&cur[p], (unsigned long long)h[q]) : 0u) - run;
                hist[p] = 0;
                run += h[q];
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t u = 0; u < PER; u++) {
            const uint32_t i = tid + 256u * u;
            if (i < fill) {
                const uint32_t p = part[i];
                const uint32_t q = loc[p] + atomicAdd(&hist[p], 1u);
                sorted[q] = raw[i];
                spart[q] = (uint8_t)p;
            }
        }
        __syncthreads();
        for (uint32_t q = tid; q < fill; q += 256) out[(uint32_t)(delta[spart[q]] + q)] = sorted[q];
        hist[tid] = 0;
        __syncthreads();
    }
}

// (D) count: the instances of final partition blockIdx.x (split blockIdx.y of gridDim.y) in LDS -- count (24 bits) and Exts (8 bits) of a key
// in one word, so that an instance touches no global memory but its own word -- then into the partition's own table entries (the labels of
// CountFilterSet: dense_part_labels_kernel).  A workgroup's share is taken in stretches of < 2^24 instances: the count cannot carry.
template <bool IS_SET, class INST>
__global__ void __launch_bounds__(1024) dense_part_count_kernel(const INST* __restrict__ in, const uint64_t* __restrict__ fine_off,
                                                                unsigned long long* __restrict__ tab) {
    constexpr uint32_t ES = IS_SET ? 2 : 1;
    extern __shared__ uint32_t s_cnt[];                              // DENSE_RANGE words: count | Exts << 24
    const uint32_t f = blockIdx.x;
    const uint64_t beg0 = fine_off[f], n = fine_off[f + 1] - beg0;
    const uint64_t beg = beg0 + n * blockIdx.y / gridDim.y, end = beg0 + n * (blockIdx.y + 1) / gridDim.y;
    unsigned long long* const mytab = tab + ((uint64_t)f << DENSE_RANGE_BITS) * ES;
    for (uint64_t c0 = beg; c0 < end; c0 += (1ull << 24) - 1) {
        const uint64_t c1 = end - c0 < (1ull << 24) - 1 ? end : c0 + (1ull << 24) - 1;
        for (uint32_t i = threadIdx.x; i < DENSE_RANGE; i += blockDim.x) s_cnt[i] = 0;
        __syncthreads();
        // (four loads in flight per thread: the kernel waits for instance words, not for the LDS)
        for (uint64_t i0 = c0 + threadIdx.x; i0 < c1; i0 += 4ull * blockDim.x) {
            INST w[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const uint64_t i = i0 + (uint64_t)u * blockDim.x; w[u] = in[i < c1 ? i : c1 - 1]; }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (i0 + (uint64_t)u * blockDim.x >= c1) continue;
                const uint32_t low = dense_inst_key(w[u]) & (DENSE_RANGE - 1), ex = dense_inst_ex(w[u]);
                const uint32_t old = atomicAdd(&s_cnt[low], 1u);
                // (a stale value can only lack bits that are set by now: the OR is then sent needlessly, never skipped wrongly)
                if (((old >> 24) & ex) != ex) atomicOr(&s_cnt[low], ex << 24);
            }
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < DENSE_RANGE; i += blockDim.x) {
            const uint32_t v = s_cnt[i];
            if (!v) continue;
            unsigned long long* e = &mytab[(uint64_t)i * ES];
            const unsigned long long add = (unsigned long long)(v & 0xffffffu), exb = (unsigned long long)(v >> 24) << 40;
            if (gridDim.y > 1) { atomicAdd(e, add); if ((__hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & exb) != exb) atomicOr(e, exb); }
            else {
                // the partition is this workgroup's alone: a load and a store that go to the memory side
                const unsigned long long old = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(e, (old + add) | exb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
    }
}

// CountFilterSet: the label masks of final partition blockIdx.x, 32 labels (group `grp`) at a time in LDS -- one more pass over the partition's
// instances per group of 32 labels in use, instead of a random read of the table entry per instance (34 of 44 ms at k = 11)
template <class INST>
__global__ void __launch_bounds__(1024) dense_part_labels_kernel(const INST* __restrict__ in, const uint64_t* __restrict__ fine_off,
                                                                 unsigned long long* __restrict__ tab, uint32_t grp) {
    extern __shared__ uint32_t s_m[];                                // DENSE_RANGE masks of 32 labels
    const uint32_t f = blockIdx.x;
    const uint64_t beg0 = fine_off[f], n = fine_off[f + 1] - beg0;
    const uint64_t beg = beg0 + n * blockIdx.y / gridDim.y, end = beg0 + n * (blockIdx.y + 1) / gridDim.y;
    if (beg == end) return;
    for (uint32_t i = threadIdx.x; i < DENSE_RANGE; i += blockDim.x) s_m[i] = 0;
    __syncthreads();
    for (uint64_t i0 = beg + threadIdx.x; i0 < end; i0 += 4ull * blockDim.x) {
        INST w[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint64_t i = i0 + (uint64_t)u * blockDim.x; w[u] = in[i < end ? i : end - 1]; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t low = dense_inst_key(w[u]) & (DENSE_RANGE - 1), label = dense_inst_label(w[u]);
            if (i0 + (uint64_t)u * blockDim.x >= end || (label >> 5) != grp) continue;
            const uint32_t bit = 1u << (label & 31u);
            if ((s_m[low] & bit) == 0u) atomicOr(&s_m[low], bit);
        }
    }
    __syncthreads();
    unsigned long long* const mytab = tab + ((uint64_t)f << DENSE_RANGE_BITS) * 2;
    for (uint32_t i = threadIdx.x; i < DENSE_RANGE; i += blockDim.x) {
        const uint32_t m = s_m[i];
        if (m) atomicOr(&mytab[(uint64_t)i * 2 + 1], (unsigned long long)m << (32u * grp));
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
    uint32_t mx = 0;
    if (is_set && s.data) {
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
    // reads [r0_, r1_) counted with one device atomic per instance (k <= 8: in LDS first)
    auto count_atomic = [&](uint64_t r0_, uint64_t r1_) -> int {
        const uint64_t nr = r1_ - r0_;
        const uint32_t blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((nr + 3) / 4, 256ull * 16));
        const bool lds = k <= DENSE_LDS_K;
        // k = 8: key-range passes with the counts in LDS (DBG_DENSE_RANGES=0: the global-atomic form, for A/B measurements)
        const bool ranges = !lds && k <= DENSE_RANGE_K && !(c->opt("DBG_DENSE_RANGES") && !strcmp(c->opt("DBG_DENSE_RANGES"), "0"));
        const size_t shm = lds ? (size_t)nkeys * 4 : (ranges ? (size_t)DENSE_RANGE * 4 : 0);
        const dim3 grid = ranges ? dim3((uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((nr + 15) / 16, 256ull)), nkeys >> DENSE_RANGE_BITS) : dim3(blocks);
        const uint32_t threads = ranges ? 1024u : 256u;
        c->t_begin("dense_count", n_kmers);
#define DL(ST, SET, LD) do { if (shm) HIP_TRY(c, hipFuncSetAttribute((const void*)dense_count_kernel<ST, SET, LD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)); \
        dense_count_kernel<ST, SET, LD><<<grid, threads, shm, c->stream>>>(s, r0_, r1_, k, tab.p); } while (0)
#define DGO(ST, SET) do { if (lds) DL(ST, SET, 1); else if (ranges) DL(ST, SET, 2); else DL(ST, SET, 0); } while (0)
        if (stranded) { if (is_set) DGO(true, true); else DGO(true, false); }
        else { if (is_set) DGO(false, true); else DGO(false, false); }
#undef DGO
#undef DL
        c->t_end();
        LAUNCH_CHECK(c, "dense_count");
        return 0;
    };
    // 8 <= k <= 15: partition, then count in LDS (DBG_DENSE_PART=0: the atomic form -- k = 8: the key-range passes -- for A/B measurements).  Batches of reads bound the
    // two instance buffers; a batch whose buffers cannot be had is counted with atomics into the same table.
    const bool partitioned = k >= DENSE_RANGE_K && !(c->opt("DBG_DENSE_PART") && !strcmp(c->opt("DBG_DENSE_PART"), "0"));
    if (s.n && n_kmers && !partitioned) DBG_TRY(count_atomic(0, s.n));
    if (s.n && n_kmers && partitioned) {
        const uint32_t F = 2u * (uint32_t)k - DENSE_RANGE_BITS, n_final = 1u << F;      // final partitions: key >> 15
        // one level up to 128 partitions (k <= 11); above that the bits are shared between the two levels -- few parts make the level-1 tiles'
        // runs long (its wavefronts collect 512 instances), a handful of sub-parts would make the level-2 counters hot; level 2 takes up to 8 bits
        uint32_t L1 = F <= 7 ? F : std::max(F - 8u, std::min((F + 1) / 2, 6u));     // k = 12 .. 15: 5 + 4, 6 + 5, 6 + 7, 7 + 8 bits (measured: DBG_DENSE_L1)
        if (c->opt("DBG_DENSE_L1")) L1 = std::min(std::max((uint32_t)atoi(c->opt("DBG_DENSE_L1")), F > 8 ? F - 8 : 1u), std::min(F, 7u));   // (measurements)
        const uint32_t L2 = F - L1, P1 = 1u << L1, shift1 = 2u * (uint32_t)k - L1;
        const uint64_t inst_cap = c->opt("DBG_DENSE_BATCH") ? (uint64_t)atoll(c->opt("DBG_DENSE_BATCH")) : 4000000000ull;
        const uint64_t reads_per_batch = std::max<uint64_t>(1, (uint64_t)((double)s.n * std::min(1.0, (double)inst_cap / (double)n_kmers)));
        const bool raw_form = !(c->opt("DBG_DENSE_RAW") && !strcmp(c->opt("DBG_DENSE_RAW"), "0"));
        DBuf<unsigned long long> fine_cnt, cursor;
        DBuf<uint64_t> fine_off, wave_off;
        DBuf<uint32_t> wave_cnt;
        ALLOC_OR_FAIL(c, fine_cnt, n_final); ALLOC_OR_FAIL(c, cursor, n_final); ALLOC_OR_FAIL(c, fine_off, (size_t)n_final + 1);
        for (uint64_t r0_ = 0; r0_ < s.n; r0_ += reads_per_batch) {
            const uint64_t r1_ = std::min<uint64_t>(s.n, r0_ + reads_per_batch), nr = r1_ - r0_;
            const bool compact = is_set && !L2;                          // one level: the label fits the 4-byte instance
            const size_t isz = is_set && !compact ? 8 : 4;
            const uint8_t* final_buf = nullptr;
            DBuf<uint8_t> buf1, buf2;
            bool counted = false;                                        // (this batch went to the atomic form)
            if (raw_form) {
                // ---- level 1, second form: instance offsets of the reads, extraction into the raw arrays + per-block counts, tile scatter ----
                const uint32_t n_blocks = (uint32_t)cdiv(nr, DENSE_BLK);
                DBuf<uint32_t> nkv, blk_cnt, raw_key;
                DBuf<uint64_t> inst_off, blk_off;
                DBuf<uint8_t> raw_aux;
                ALLOC_OR_FAIL(c, nkv, nr); ALLOC_OR_FAIL(c, inst_off, nr + 1);
                ALLOC_OR_FAIL(c, blk_cnt, (size_t)P1 * n_blocks); ALLOC_OR_FAIL(c, blk_off, (size_t)P1 * n_blocks + 1);
                c->t_begin("dense_part_hist", n_kmers);
                dense_nk_kernel<<<cdiv(nr, 256), 256, 0, c->stream>>>(s, r0_, r1_, k, nkv.p);
                LAUNCH_CHECK(c, "dense_nk");
                DBG_TRY(scan_exclusive_u32_u64(c, nkv.p, inst_off.p, nr));
                uint64_t n_inst = 0;
                HIP_TRY(c, hipMemcpyAsync(&n_inst, inst_off.p + nr, 8, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                if (!n_inst) { c->t_end(); continue; }
                if (n_inst >= (1ull << 32) || !raw_key.alloc(c, n_inst) || !raw_aux.alloc(c, n_inst * (is_set ? 2 : 1)) || !buf1.alloc(c, n_inst * isz) ||
                    (L2 && !buf2.alloc(c, n_inst * 4))) {
                    c->t_end();
                    raw_key.release(); raw_aux.release(); buf1.release(); buf2.release();
                    DBG_TRY(count_atomic(r0_, r1_));
                    continue;
                }
#define EX(ST, SET) dense_extract_kernel<ST, SET><<<n_blocks, 256, 0, c->stream>>>(s, r0_, r1_, k, shift1, P1, inst_off.p, n_blocks, raw_key.p, (DenseAux<SET>::type*)raw_aux.p, blk_cnt.p)
                if (stranded) { if (is_set) EX(true, true); else EX(true, false); }
                else { if (is_set) EX(false, true); else EX(false, false); }
#undef EX
                c->t_end();
                LAUNCH_CHECK(c, "dense_extract");
                DBG_TRY(scan_exclusive_u32_u64(c, blk_cnt.p, blk_off.p, (uint64_t)P1 * n_blocks));
                c->t_begin("dense_part_scatter", n_kmers);
                if (!is_set) dense_part_scatter1t_kernel<false, false><<<n_blocks, 256, 0, c->stream>>>(raw_key.p, raw_aux.p, inst_off.p, nr, shift1, P1, blk_off.p, n_blocks, (uint32_t*)buf1.p);
                else if (compact) dense_part_scatter1t_kernel<true, true><<<n_blocks, 256, 0, c->stream>>>(raw_key.p, (const uint16_t*)raw_aux.p, inst_off.p, nr, shift1, P1, blk_off.p, n_blocks, (uint32_t*)buf1.p);
                else dense_part_scatter1t_kernel<true, false><<<n_blocks, 256, 0, c->stream>>>(raw_key.p, (const uint16_t*)raw_aux.p, inst_off.p, nr, shift1, P1, blk_off.p, n_blocks, (unsigned long long*)buf1.p);
                LAUNCH_CHECK(c, "dense_part_scatter1t");
                if (L2) {
                    HIP_TRY(c, hipMemsetAsync(fine_cnt.p, 0, (size_t)n_final * 8, c->stream));
                    const dim3 hgrid(std::max(8u, 2048u / P1), P1);
                    if (is_set) dense_fine_hist_kernel<unsigned long long><<<hgrid, 256, 0, c->stream>>>((const unsigned long long*)buf1.p, blk_off.p, n_blocks, L2, fine_cnt.p);
                    else dense_fine_hist_kernel<uint32_t><<<hgrid, 256, 0, c->stream>>>((const uint32_t*)buf1.p, blk_off.p, n_blocks, L2, fine_cnt.p);
                    LAUNCH_CHECK(c, "dense_fine_hist");
                    DBG_TRY(scan_exclusive_u64(c, (const uint64_t*)fine_cnt.p, fine_off.p, n_final));
                } else dense_part_fineoff_kernel<<<cdiv(P1 + 1, 256), 256, 0, c->stream>>>(blk_off.p, n_blocks, P1, fine_off.p);
                HIP_TRY(c, hipStreamSynchronize(c->stream));           // (the raw arrays go back to the pool before level 2 asks for nothing more: they leave scope here)
            } else {
            // the grid of pass A (16 wavefronts per workgroup) and of pass B (one per workgroup) hold the same number of wavefronts
            const uint32_t hist_blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((nr + 15) / 16, 512ull));
            const uint64_t n_waves = (uint64_t)hist_blocks * 16;
            ALLOC_OR_FAIL(c, wave_cnt, (size_t)P1 * n_waves); ALLOC_OR_FAIL(c, wave_off, (size_t)P1 * n_waves + 1);
            if (L2) HIP_TRY(c, hipMemsetAsync(fine_cnt.p, 0, (size_t)n_final * 8, c->stream));
            c->t_begin("dense_part_hist", n_kmers);
            {
                const size_t shm = ((size_t)16 * P1 + (L2 ? n_final : 0)) * 4;
#define HL(ST) do { HIP_TRY(c, hipFuncSetAttribute((const void*)dense_part_hist_kernel<ST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)); \
                dense_part_hist_kernel<ST><<<hist_blocks, 1024, shm, c->stream>>>(s, r0_, r1_, k, shift1, P1, L2 ? n_final : 0u, wave_cnt.p, fine_cnt.p); } while (0)
                if (stranded) HL(true); else HL(false);
#undef HL
            }
            c->t_end();
            LAUNCH_CHECK(c, "dense_part_hist");
            DBG_TRY(scan_exclusive_u32_u64(c, wave_cnt.p, wave_off.p, (uint64_t)P1 * n_waves));
            if (L2) DBG_TRY(scan_exclusive_u64(c, (const uint64_t*)fine_cnt.p, fine_off.p, n_final));
            else dense_part_fineoff_kernel<<<cdiv(P1 + 1, 256), 256, 0, c->stream>>>(wave_off.p, n_waves, P1, fine_off.p);
            uint64_t n_inst = 0;
            HIP_TRY(c, hipMemcpyAsync(&n_inst, wave_off.p + (uint64_t)P1 * n_waves, 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (!n_inst) continue;
            // no room, or more instances than the kernels' 32-bit offsets hold (a batch of unusually long reads): this batch takes the atomic form
            if (n_inst >= (1ull << 32) || !buf1.alloc(c, n_inst * isz) || (L2 && !buf2.alloc(c, n_inst * 4))) {
                buf1.release(); buf2.release();
                DBG_TRY(count_atomic(r0_, r1_));
                continue;
            }
            c->t_begin("dense_part_scatter", n_kmers);
            {
#define SL(ST, SET) do { if (SET && compact) dense_part_scatter1_kernel<ST, SET, 128, true><<<(uint32_t)n_waves, 64, 0, c->stream>>>(s, r0_, r1_, k, shift1, P1, wave_off.p, (uint32_t*)buf1.p); \
                else dense_part_scatter1_kernel<ST, SET, 128><<<(uint32_t)n_waves, 64, 0, c->stream>>>(s, r0_, r1_, k, shift1, P1, wave_off.p, (DenseInst<SET>::type*)buf1.p); } while (0)
                if (stranded) { if (is_set) SL(true, true); else SL(true, false); }
                else { if (is_set) SL(false, true); else SL(false, false); }
#undef SL
            }
            LAUNCH_CHECK(c, "dense_part_scatter1");
            }   // (first form of level 1)
            (void)counted;
            final_buf = buf1.p;
            if (L2) {
                HIP_TRY(c, hipMemcpyAsync(cursor.p, fine_off.p, (size_t)n_final * 8, hipMemcpyDeviceToDevice, c->stream));
                const dim3 grid(std::max(8u, 4096u / P1), P1);                // (~4096 workgroups whatever the fan-out of level 1)
                if (is_set) dense_part_scatter2_kernel<true><<<grid, 256, 0, c->stream>>>((const unsigned long long*)buf1.p, fine_off.p, L2, cursor.p, (uint32_t*)buf2.p);
                else dense_part_scatter2_kernel<false><<<grid, 256, 0, c->stream>>>((const uint32_t*)buf1.p, fine_off.p, L2, cursor.p, (uint32_t*)buf2.p);
                LAUNCH_CHECK(c, "dense_part_scatter2");
                final_buf = buf2.p;
            }
            c->t_end();
            c->t_begin("dense_part_count", n_kmers);
            {
                const uint32_t splits = std::max(1u, 512u / n_final);
                const dim3 grid(n_final, splits);
                const size_t shm = (size_t)DENSE_RANGE * 4;
#define CNT(SET, T) do { HIP_TRY(c, hipFuncSetAttribute((const void*)dense_part_count_kernel<SET, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)); \
                    dense_part_count_kernel<SET, T><<<grid, 1024, shm, c->stream>>>((const T*)final_buf, fine_off.p, tab.p); } while (0)
#define LAB(T) do { HIP_TRY(c, hipFuncSetAttribute((const void*)dense_part_labels_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)); \
                    for (uint32_t grp = 0; grp <= mx / 32; grp++) dense_part_labels_kernel<T><<<grid, 1024, shm, c->stream>>>((const T*)final_buf, fine_off.p, tab.p, grp); } while (0)
                if (!is_set) CNT(false, uint32_t);
                else { CNT(true, uint32_t); LAB(uint32_t); }             // (the final buffer's instances are 4-byte words for CountFilterSet too)
#undef LAB
#undef CNT
            }
            c->t_end();
            LAUNCH_CHECK(c, "dense_part_count");
            HIP_TRY(c, hipStreamSynchronize(c->stream));               // the buffers go back to the pool
        }
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
