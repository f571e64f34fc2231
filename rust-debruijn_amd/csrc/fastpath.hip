// Fast counting path:  reads --(minimizer scan)--> super-k-mer records binned by hashed minimizer
//                      --(one workgroup per bin, LDS hash table)--> (k-mer, Exts, count[, colour set])
//                      --(order-restoring sort)--> the ascending table filter_kmers hands to
//                      BoomHashMap2::new (src/filter.rs:227-230).
//
// Why this shape on MI355X: every instance of a canonical k-mer contains the same p-mers, so its
// window minimizer -- hence its bin -- is a function of the k-mer alone (the MSP argument,
// src/msp.rs:279-324).  Keeping k-mers packed as super-k-mers until they are
// inside a CU means HBM sees ~2 B per k-mer instance instead of a 17-20 B record per radix pass, and
// the count/OR reduction (CountFilter / CountFilterSet::summarize, filter.rs:53-62, :85-100)
// happens in the 160 KB LDS of one CU with 32-bit LDS atomics.  Bins of any size are streamed in chunks;
// if a bin holds more distinct k-mers than the table, the workgroup re-streams it in P hash-selected
// passes (the same idea as the reference's bucket_ranges passes, filter.rs:156-168).
//
// The internal partition uses its own minimizer scheme (p up to 15, canonical p-mers ordered by a bijective
// hash -- a `permutation` in the reference's terms, msp.rs:55-59 -- pieces cut where the window's minimum VALUE
// changes, so that the choice is the same on both strands); it is invisible in the result.
// Records go straight from the scan into per-bin slabs.  Because the cut points depend on the sequence alone, reads that
// cover the same stretch of the genome produce identical records: the counting workgroup merges them (weight + colour
// union) before expanding k-mers.  See DESIGN.md section 3.1 for the measurements behind each structural choice.
#include "dbg_internal.hpp"
#include "dbg_msp_device.hpp"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

// ------------------------------------------------------------------------------------------------
// internal minimizer order + bin hash
// ------------------------------------------------------------------------------------------------
struct LabelInv { uint32_t v[64]; uint32_t on; };       // colour index -> D1 label (sparse label alphabets)
struct FastCfg {
    int k, p;
    int stranded;
    uint32_t nbins;
    const uint8_t* lmap;        // D1 label -> dense colour index (sparse label alphabets), or null: the label is the index
    int strand_norm;            // store every piece as the smaller of (piece, reverse complement): non-stranded counting with odd k
};

// p-mer from the two words that hold it (w1 is ignored when the p-mer ends inside w0): no branches
__device__ __forceinline__ uint32_t pmer_from_words(uint64_t w0, uint64_t w1, uint32_t s /* bit offset 0..62 */, int p) {
    const uint64_t v = (w0 << s) | ((w1 >> 1) >> (63 - s));
    return (uint32_t)(v >> (64 - 2 * p));
}
// The scan kernel addresses a read through a wave-uniform word pointer `wr` (word of the read's first
// base) and 32-bit base offsets `rel` from that word's first base: address arithmetic stays 32-bit and
// the base pointer lives in scalar registers.  Word indices are clamped to last_rel (unconditional loads).
__device__ __forceinline__ uint32_t rel_pmer(const uint64_t* __restrict__ wr, uint32_t rel, uint32_t last_rel, int p) {
    const uint32_t wi = rel >> 5, s = (rel & 31u) * 2u;
    const uint64_t w0 = wr[wi < last_rel ? wi : last_rel], w1 = wr[wi + 1 < last_rel ? wi + 1 : last_rel];
    return pmer_from_words(w0, w1, s, p);
}
__device__ __forceinline__ uint32_t rel_base(const uint64_t* __restrict__ wr, uint32_t rel) {
    return (uint32_t)(wr[rel >> 5] >> (62 - 2 * (rel & 31u))) & 3u;
}
// nb (1..32) bases starting at rel, left-aligned in a u64 (rest zero)
__device__ __forceinline__ uint64_t rel_word(const uint64_t* __restrict__ wr, uint32_t rel, uint32_t last_rel, uint32_t nb) {
    const uint32_t wi = rel >> 5, s = (rel & 31u) * 2u;
    const uint64_t w0 = wr[wi < last_rel ? wi : last_rel], w1 = wr[wi + 1 < last_rel ? wi + 1 : last_rel];
    const uint64_t v = (w0 << s) | ((w1 >> 1) >> (63 - s));
    return v & (~0ull << (64 - 2 * nb));
}
// reverse complement of a right-aligned p-mer held in 32 bits
__device__ __forceinline__ uint32_t pmer_rc32(uint32_t pm, int p) {
    uint32_t r = __brev(pm);
    r = ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
    return (~r) >> (32 - 2 * p);
}

// bijection on 2p-bit integers (odd multiply and xorshift are both invertible mod 2^(2p))
__device__ __forceinline__ uint32_t mix_pmer(uint32_t x, int p) {
    const uint32_t mask = p >= 16 ? 0xffffffffu : ((1u << (2 * p)) - 1);
#ifdef DBG_MIX_CHEAP
    x = (x + (x << 3)) & mask; x ^= x >> p;
    x = (x + (x << 5)) & mask; x ^= x >> 7;
    x = (x + (x << 2)) & mask; x ^= x >> p;
    return x;
#endif
    x = (x * 0x9E3779B1u) & mask;
    x ^= x >> p;
    x = (x * 0x85EBCA6Bu) & mask;
    x ^= x >> p;
    return x;
}
struct FastScore {
    int p, stranded;
    __device__ __forceinline__ uint32_t operator()(uint32_t pm) const {
        if (!stranded) { uint32_t r = pmer_rc32(pm, p); pm = pm < r ? pm : r; }
        return mix_pmer(pm, p);
    }
};
// bin of a super-k-mer from its minimizer's ordering hash (a bijection of the canonical p-mer).  Minimizer hashes
// crowd towards small values, so the bin index comes from a second, unrelated mix of that value.
__device__ __forceinline__ uint32_t bin_of_hash(const FastCfg& c, uint32_t mh) {
    uint32_t h = mh * 0xC2B2AE35u + 0x27D4EB2Fu;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return (uint32_t)(((uint64_t)h * c.nbins) >> 32);
}

// ------------------------------------------------------------------------------------------------
// Super-k-mer scan, one wavefront per read.  Lanes own p-mer positions (3 x 64 positions = one tile of
// 128 window starts + the k-p tail); the sliding-window argmin is a log-step min in LDS over packed
// (hash << 32 | position) words, so equal hashes resolve to the leftmost position; a window starts a
// new super-k-mer when its argmin position differs from its predecessor's.  Records are written in
// read order to a temporary buffer (wave-level allocation, one atomic per tile) together with their
// bin id; a second, purely bandwidth-bound kernel moves them into bin order.
// record = NBW words: bases left-aligned, MSB first; the low META_BITS bits of the last word (always free: NBW is
// chosen that way) hold len (7) | exts << 7 (8) | D1 << 15 (6: the fast path handles up to 64 colours).  24 bytes at k = 47.
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_TILE_W = 128;                // window starts per tile
constexpr int SCAN_ARR = 192 + 64;              // positions per tile + padding for the shifted reads
constexpr uint32_t SCAN_CHUNK = 1024;           // records a wave reserves per global atomic (one hot address otherwise)
constexpr uint32_t BIN_INVALID = 0xffffffffu;   // unused slot of a reserved chunk
constexpr int META_BITS = 21;
constexpr int WEIGHT_BITS = 4;                  // sharded flow: (copies - 1) of a record merged on the sending rank, directly above the meta bits
constexpr uint32_t NCLS = 1;                    // length classes per bin: records of similar k-mer count sit together
                                                // so that the 64 records a wave processes finish at about the same time

// Pending pieces (start window, end window, minimizer hash, lane of the read in the current batch of 64) wait in a per-wave
// LDS ring and are turned into records 64 at a time, on full wavefronts (record building is the longest code of the scan).
// DIRECT: a record goes straight to slot atomicAdd(cursor[bin]) of its bin's fixed-capacity slab (slab_cap records per
// bin); only the records of bins that outgrow their slab take the read-order temporary buffer and the scatter kernel.
constexpr uint32_t PLC = 256;                   // ring slots per wave (wave-per-read kernel: 4 words per piece)
constexpr uint32_t PLC_PACKED = 128;            // lane-per-read kernel: 2 words per piece {start | end << 11 | lane << 22, hash}
// LAB (label lists, fast_labellists.hpp): the record has one more word, which carries the read's full D1 label -- the 6-bit colour in
// the meta bits is all there is otherwise.  (First form: a side array at the record's index; the second scattered store cost the
// scan 8 of 40 ms at the C2 shape, and the exchange of the sharded flow would have had to carry two streams.)
template <int NBW, bool DIRECT, bool PACKED = false, bool LAB = false>
struct PieceEmitter {
    static constexpr int RW = NBW + (LAB ? 1 : 0);
    static constexpr uint32_t CAP = PACKED ? PLC_PACKED : PLC;
    const SeqDev& s;
    const FastCfg& c;
    uint32_t* __restrict__ hist;
    uint64_t* __restrict__ tmp_recs;
    uint32_t* __restrict__ tmp_bin;
    unsigned long long* __restrict__ tmp_cursor;
    uint64_t tmp_cap;
    uint32_t* __restrict__ flags;
    uint64_t* __restrict__ slab;
    uint32_t slab_cap;
    uint32_t* __restrict__ cursor;
    uint32_t* PL;                               // this wave's ring: 4 x PLC words (PACKED: 2 x PLC_PACKED)
    uint32_t lane;
    uint64_t lt;
    uint64_t last_word;
    uint32_t pl_head = 0, pl_n = 0;             // wave-uniform
    uint64_t chunk_base = 0;                    // wave-uniform sub-allocator over reserved chunks of the temporary buffer
    uint32_t chunk_used = SCAN_CHUNK;

    // lanes with `on` append one piece each (ring order = lane order); the counters stay in scalar registers
    __device__ __forceinline__ void push(bool on, uint32_t ps, uint32_t pe, uint32_t pa, uint32_t pr) {
        const uint64_t em = __ballot(on);
        if (on) {
            const uint32_t q = (pl_head + pl_n + (uint32_t)__popcll(em & lt)) & (CAP - 1);
            if (PACKED) { PL[q] = ps | (pe << 11) | (pr << 22); PL[CAP + q] = pa; }
            else { PL[q] = ps; PL[CAP + q] = pe; PL[2 * CAP + q] = pa; PL[3 * CAP + q] = pr; }
        }
        pl_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pl_n + (uint32_t)__popcll(em)));
    }

    // v_*: metadata of the batch's reads, one read per lane (a piece's read is fetched back from the owning lane)
    __device__ __forceinline__ void flush(uint32_t cnt, uint32_t v_m, uint64_t v_st, uint32_t v_ex, uint32_t v_d1) {
#ifdef DBG_ABL_NO_FLUSH
        pl_head = (pl_head + cnt) & (CAP - 1); pl_n -= cnt; return;
#endif
        const int k = c.k;
        const uint64_t* __restrict__ w = s.words;
        const uint32_t q = (pl_head + lane) & (CAP - 1);
        const bool act = lane < cnt;
        uint32_t ps, pe, pa, pr;
        if (PACKED) {
            const uint32_t x = PL[q];
            ps = x & 2047u; pe = (x >> 11) & 2047u; pr = act ? x >> 22 : 0u; pa = PL[CAP + q];
        } else {
            ps = PL[q]; pe = PL[CAP + q]; pa = PL[2 * CAP + q]; pr = act ? PL[3 * CAP + q] : 0u;
        }
        const uint32_t m = __shfl(v_m, pr);
        const uint64_t st = __shfl(v_st, pr);
        const uint32_t sexts = __shfl(v_ex, pr), d1 = __shfl(v_d1, pr);
        pl_head = (uint32_t)__builtin_amdgcn_readfirstlane((int)((pl_head + cnt) & (CAP - 1)));
        pl_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pl_n - cnt));
        // The slot in the bin's slab comes from a global atomic with return (the longest latency of the flush): it is requested
        // first and the record is built while it is in flight.
        static_assert(NCLS == 1, "one bin per minimizer class");
        uint32_t b = 0, r = 0;
        if (act) {
            b = bin_of_hash(c, pa);
#ifdef DBG_ABL_NO_ATOMIC
            if (DIRECT) r = (pa * 2654435761u + lane * 40503u) % slab_cap;
#else
            if (DIRECT) r = atomicAdd(&cursor[b], 1u);
#endif
        }
        uint64_t rv[RW];
#pragma unroll
        for (int qq = 0; qq < RW; qq++) rv[qq] = 0;
        if (act) {
            const uint32_t nwin = m - (uint32_t)k + 1;
            const uint32_t len = (pe == nwin ? m : pe + (uint32_t)k - 1) - ps;
            // bases [ps - 1, ps + len] of the read (the piece and its two neighbours) lie in at most NBW + 1 words (2 * 32 * NBW >=
            // 2k - p + 20 meta bits, and the piece may start anywhere in a word): they are fetched once, everything else is shifts
            const uint64_t o = st + ps, wi0 = o >> 5;
            const uint32_t sh = (uint32_t)(o & 31) * 2u;
            uint64_t x[NBW + 1];
#pragma unroll
            for (int qq = 0; qq <= NBW; qq++) x[qq] = w[wi0 + qq < last_word ? wi0 + qq : last_word];
            uint32_t le = sexts & 0xfu, re = sexts >> 4;
            if (ps > 0) le = 1u << (sh ? (uint32_t)(x[0] >> (64u - sh)) & 3u : packed_get(w, o - 1));
#pragma unroll
            for (int qq = 0; qq < NBW; qq++) {
                const uint64_t v = (x[qq] << sh) | ((x[qq + 1] >> 1) >> (63u - sh));      // bases [32 qq, 32 qq + 32) of the piece
                const uint32_t b0 = (uint32_t)qq * 32;
                const uint32_t nb = b0 < len ? (len - b0 < 32 ? len - b0 : 32) : 0;
                rv[qq] = nb ? v & (~0ull << (64 - 2 * nb)) : 0ull;
                if (ps + len < m && len >= b0 && len < b0 + 32) re = 1u << ((uint32_t)(v >> (62u - 2u * (len - b0))) & 3u);   // base right of the piece
            }
            // Non-stranded counting canonicalises every k-mer anyway, so a piece may be stored on either strand: storing the smaller of
            // (piece, reverse complement) makes the copies that reads of the two strands produce IDENTICAL records, which the counting
            // workgroup (and the sender-side merge of the sharded flow) then merge.  Odd k only: a k-mer that is its own reverse
            // complement (even k) takes its Exts in the orientation the READ presents it (ties flip, lib.rs:226-230), which a
            // stored strand would change.
            if (c.strand_norm) {
                uint64_t t[NBW], rc[NBW];
#pragma unroll
                for (int qq = 0; qq < NBW; qq++) t[qq] = ~rev2_64(rv[NBW - 1 - qq]);       // base i of the piece -> slot 32 NBW - 1 - i
                const uint32_t S = 2u * (32u * NBW - len), ws = S >> 6, bs = S & 63u;      // the rc sits at the END of the slots: shift it up
                // word shift in two conditional steps (ws <= 3: len >= k >= 16), then the bit shift across neighbours: no register indexing
#pragma unroll
                for (int qq = 0; qq < NBW; qq++) t[qq] = (ws & 1u) ? (qq + 1 < NBW ? t[qq + 1] : 0ull) : t[qq];
#pragma unroll
                for (int qq = 0; qq < NBW; qq++) t[qq] = (ws & 2u) ? (qq + 2 < NBW ? t[qq + 2] : 0ull) : t[qq];
                bool less = false, decided = false;
#pragma unroll
                for (int qq = 0; qq < NBW; qq++) {
                    const uint64_t nx = qq + 1 < NBW ? t[qq + 1] : 0ull;
                    rc[qq] = (t[qq] << bs) | ((nx >> 1) >> (63u - bs));
                    if (!decided && rc[qq] != rv[qq]) { decided = true; less = rc[qq] < rv[qq]; }
                }
                if (less) {
#pragma unroll
                    for (int qq = 0; qq < NBW; qq++) rv[qq] = rc[qq];
                    const uint32_t e = exts_rc((re << 4) | le);
                    le = e & 0xfu; re = e >> 4;
                }
            }
            rv[NBW - 1] |= (uint64_t)len | ((uint64_t)((re << 4) | le) << 7) | ((uint64_t)(d1 & 63u) << 15);
            if (LAB) rv[RW - 1] = (uint64_t)d1;
        }
        auto store_rec = [&](uint64_t* o) {
            if (RW % 2 == 0) {
#pragma unroll
                for (int qq = 0; qq < RW / 2; qq++) ((ulonglong2*)o)[qq] = make_ulonglong2(rv[2 * qq], rv[2 * qq + 1]);
            } else {
#pragma unroll
                for (int qq = 0; qq < RW; qq++) o[qq] = rv[qq];
            }
        };
        bool to_tmp = act;                        // lanes whose record goes to the read-order buffer
        if (DIRECT) {
            to_tmp = act && r >= slab_cap;
#ifndef DBG_ABL_NO_STORE
            if (act && !to_tmp) store_rec(slab + ((uint64_t)b * slab_cap + r) * RW);
#else
            if (act && !to_tmp && rv[0] == 0x123456789ull) store_rec(slab + ((uint64_t)b * slab_cap + r) * RW);
#endif
        }
        const uint64_t tm = __ballot(to_tmp);
        if (!tm) return;
        const uint32_t nt = (uint32_t)__popcll(tm);
        if (chunk_used + nt > SCAN_CHUNK) {
            unsigned long long nb = 0;
            if (lane == 0) nb = atomicAdd(tmp_cursor, (unsigned long long)SCAN_CHUNK);
            chunk_base = __shfl(nb, 0);
            chunk_used = 0;
        }
        const uint64_t idx = chunk_base + chunk_used + (uint32_t)__popcll(tm & lt);
        chunk_used += nt;
        if (chunk_base + SCAN_CHUNK > tmp_cap) { if (lane == 0) atomicOr(&flags[0], 1u); return; }
        if (to_tmp) {
            atomicAdd(&hist[b], 1u);
            store_rec(tmp_recs + idx * RW);
            tmp_bin[idx] = b;
        }
    }
};

// Wave-per-read scan (reads longer than long_min bases: contigs, long reads).
template <int NBW, bool DIRECT, bool LAB = false>
__global__ void __launch_bounds__(256) sk_scan_kernel(SeqDev s, FastCfg c, uint32_t* __restrict__ hist,
                                                      uint64_t* __restrict__ tmp_recs, uint32_t* __restrict__ tmp_bin,
                                                      unsigned long long* __restrict__ tmp_cursor, uint64_t tmp_cap,
                                                      uint32_t* __restrict__ flags, uint64_t* __restrict__ slab,
                                                      uint32_t slab_cap, uint32_t* __restrict__ cursor, uint32_t long_min) {
    __shared__ uint32_t s_arr[4][SCAN_ARR];      // ordering hash of the canonical p-mer at each position of the tile
    __shared__ uint32_t s_pl[4][4 * PLC];        // per-wave ring of pending pieces
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* A = s_arr[wave];
    uint32_t* PL = s_pl[wave];
    A[192 + lane] = ~0u;
    const int k = c.k, p = c.p;
    const uint32_t W = (uint32_t)(k - p + 1);
    const uint64_t* __restrict__ w = s.words;
    const uint64_t last_word = s.n_words ? s.n_words - 1 : 0;
    const FastScore sc{c.p, c.stranded};
    const uint64_t gwave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t lt = lanemask_lt();
    PieceEmitter<NBW, DIRECT, false, LAB> E{s, c, hist, tmp_recs, tmp_bin, tmp_cursor, tmp_cap, flags, slab, slab_cap, cursor, PL, lane, lt, last_word};

    // 64 reads per wave iteration: their metadata arrives in three coalesced loads and is broadcast
    // lane by lane, so the per-read critical path holds a single HBM round trip (the packed words).
    for (uint64_t rb = gwave * 64; rb < s.n; rb += nwaves * 64) {
      const uint64_t my = rb + lane;
      uint32_t v_m = 0, v_ex = 0, v_d1 = 0;
      uint64_t v_st = 0;
      if (my < s.n) {
          v_m = s.length[my]; v_st = s.start[my];
          if (s.exts) v_ex = s.exts[my];
          if (s.data) v_d1 = s.data_width == 1 ? ((const uint8_t*)s.data)[my] : (s.data_width == 2 ? ((const uint16_t*)s.data)[my] : ((const uint32_t*)s.data)[my]);
          if (c.lmap) v_d1 = c.lmap[v_d1];
      }
      const uint32_t nb_reads = (uint32_t)(s.n - rb < 64 ? s.n - rb : 64);
      for (uint32_t rj = 0; rj < nb_reads; rj++) {
        const uint32_t m = __shfl(v_m, rj);
        if (m < (uint32_t)k || m <= long_min) continue;     // shorter reads belong to the lane-per-read kernel
        const uint64_t st = __shfl(v_st, rj);
        const uint32_t nwin = m - (uint32_t)k + 1, npos = m - (uint32_t)p + 1;
        const uint64_t w_first = st >> 5;                                      // word holding the read's first base
        const uint64_t* __restrict__ wr = w + w_first;
        const uint32_t sb = (uint32_t)(st & 31);                               // its position inside that word
        const uint32_t last_rel = (uint32_t)(last_word - w_first < 0x7fffffffull ? last_word - w_first : 0x7fffffffull);
        uint32_t open_start = 0;            // start window of the piece still open (wave-uniform)
        uint32_t open_vstart = 0;           // start window of the run of equal minimizer values still open
        uint32_t carry_arg = 0xffffffffu;   // minimizer hash of the last window of the previous tile

        for (uint32_t t0 = 0; t0 < nwin; t0 += SCAN_TILE_W) {
            // ---- hashed p-mers of this tile: all six word loads are issued before any is consumed
            //      (indices clamped to the buffer, results of out-of-range lanes are discarded) ----
            uint32_t ov[3];                         // running window minimum, packed
            {
                uint64_t l0[3], l1[3];
                uint32_t sh[3];
                // the clamps only matter for the last sequences of the buffer: one scalar test per tile picks the plain loads
                if (((sb + t0 + 191u) >> 5) + 1u <= last_rel) {
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) {
                        const uint32_t rel = sb + t0 + ch * 64 + lane;
                        const uint32_t wi = rel >> 5;
                        sh[ch] = (rel & 31u) * 2u;
                        l0[ch] = wr[wi];
                        l1[ch] = wr[wi + 1];
                    }
                } else {
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) {
                        const uint32_t rel = sb + t0 + ch * 64 + lane;
                        const uint32_t wi = rel >> 5;
                        sh[ch] = (rel & 31u) * 2u;
                        l0[ch] = wr[wi < last_rel ? wi : last_rel];
                        l1[ch] = wr[wi + 1 < last_rel ? wi + 1 : last_rel];
                    }
                }
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    const uint32_t pos = t0 + ch * 64 + lane;
                    const uint32_t pm = pmer_from_words(l0[ch], l1[ch], sh[ch], p);
                    // ordering key = bijective hash of the canonical p-mer (< 2^30): equal keys are equal p-mers, so the
                    // window minimum VALUE -- all that is kept -- is the same on both strands and in every read
                    ov[ch] = pos < npos ? sc(pm) : 0xffffffffu;
                    A[ch * 64 + lane] = ov[ch];
                }
            }
            // ---- log-step sliding minimum: after the loop ov covers [i, i + span) ----
            uint32_t span = 1;
            for (; 2 * span <= W; span <<= 1) {
                const uint32_t x0 = A[lane + span], x1 = A[64 + lane + span], x2 = A[128 + lane + span];
                ov[0] = x0 < ov[0] ? x0 : ov[0];
                ov[1] = x1 < ov[1] ? x1 : ov[1];
                ov[2] = x2 < ov[2] ? x2 : ov[2];
                A[lane] = ov[0];
                A[64 + lane] = ov[1];
                A[128 + lane] = ov[2];
            }
            const uint32_t rem = W - span;
            uint32_t arg[2];
#pragma unroll
            for (int ch = 0; ch < 2; ch++) {
                const uint32_t y = A[ch * 64 + lane + rem];
                arg[ch] = y < ov[ch] ? y : ov[ch];                               // hash of the window's minimizer
            }
            // ---- piece boundaries ----
            uint32_t up0 = __shfl_up(arg[0], 1), up1 = __shfl_up(arg[1], 1);
            uint32_t last0 = __shfl(arg[0], 63);
            uint32_t prev[2] = {lane ? up0 : carry_arg, lane ? up1 : last0};
            // A piece = maximal run of windows with the same minimizer value, cut every W windows (a record holds
            // at most W k-mers; only repeats -- the same p-mer again within a window length -- run longer).
            bool isb[2];
            uint64_t mask[2];
#pragma unroll
            for (int ch = 0; ch < 2; ch++) {
                uint32_t i = t0 + ch * 64 + lane;
                isb[ch] = i < nwin && (i == 0 || arg[ch] != prev[ch]);
                mask[ch] = __ballot(isb[ch]);
            }
#pragma unroll
            for (int ch = 0; ch < 2; ch++) {
                const uint32_t i = t0 + ch * 64 + lane;
                const uint64_t le = mask[ch] & (lt | (1ull << lane));               // value boundaries at or before this lane
                uint32_t vs;                                                       // start of the run this window is in
                if (le) vs = t0 + ch * 64 + (63 - __clzll(le));
                else if (ch == 1 && mask[0]) vs = t0 + (63 - __clzll(mask[0]));
                else vs = open_vstart;
                const uint32_t d = i - vs;
                if (i < nwin && d >= W && d % W == 0) isb[ch] = true;
            }
            {   // the next tile needs the start of the run that is open at the end of this one
                uint32_t nv = open_vstart;
                if (mask[1]) nv = t0 + 64 + (63 - __clzll(mask[1]));
                else if (mask[0]) nv = t0 + (63 - __clzll(mask[0]));
                open_vstart = nv;
            }
            mask[0] = __ballot(isb[0]);
            mask[1] = __ballot(isb[1]);
            const uint32_t tile_last = (nwin - t0 < (uint32_t)SCAN_TILE_W ? nwin - t0 : (uint32_t)SCAN_TILE_W) - 1;
            const uint32_t last_arg = tile_last < 64 ? __shfl(arg[0], tile_last) : __shfl(arg[1], tile_last - 64);
            const bool read_ends = t0 + SCAN_TILE_W >= nwin;
            uint32_t new_open = open_start;
            if (mask[1]) new_open = t0 + 64 + (63 - __clzll(mask[1]));
            else if (mask[0]) new_open = t0 + (63 - __clzll(mask[0]));
            // ---- collect the pieces this tile closes: the boundary at window i closes the piece that
            //      started at the previous boundary; lane 0 also closes the last piece when the read ends
            //      here.  They are compacted into a wave-private list so that the (long) record-building
            //      code below runs once per tile with one piece per lane ----
            uint32_t npieces = 0;
#pragma unroll
            for (int slot = 0; slot < 2; slot++) {
                const uint32_t i = t0 + slot * 64 + lane;
                const bool emit = isb[slot] && i > 0;
                const uint64_t em = __ballot(emit);
                if (emit) {
                    uint32_t ps;
                    const uint64_t below = mask[slot] & lt;
                    if (below) ps = t0 + slot * 64 + (63 - __clzll(below));
                    else if (slot == 1 && mask[0]) ps = t0 + (63 - __clzll(mask[0]));
                    else ps = open_start;
                    const uint32_t q = (E.pl_head + E.pl_n + npieces + __popcll(em & lt)) & (PLC - 1);
                    PL[q] = ps; PL[PLC + q] = i; PL[2 * PLC + q] = prev[slot]; PL[3 * PLC + q] = rj;
                }
                npieces += __popcll(em);
            }
            if (read_ends) {
                const uint32_t q = (E.pl_head + E.pl_n + npieces) & (PLC - 1);
                if (lane == 0) { PL[q] = new_open; PL[PLC + q] = nwin; PL[2 * PLC + q] = last_arg; PL[3 * PLC + q] = rj; }
                npieces++;
            }
            E.pl_n += npieces;
            while (E.pl_n >= 64) E.flush(64u, v_m, v_st, v_ex, v_d1);
            open_start = new_open;
            carry_arg = last_arg;
        }
      }
      if (E.pl_n) E.flush(E.pl_n, v_m, v_st, v_ex, v_d1);   // the lanes' read metadata changes with the next batch
    }
}

// Lane-per-read scan (the default for reads of up to SCAN_LANE_MAX bases): a wavefront takes 64 reads, one per lane, and
// every lane walks its own read base by base -- rolling forward and reverse-complement p-mer, ordering hash, sliding-window
// minimum, piece boundaries -- so that all 64 lanes do useful work at every step (the wave-per-read formulation hashes 136
// p-mer positions in 3 x 64 lane slots and runs a 5-step log-step minimum over them: about 3x the instructions per read).
// Control flow stays wave-uniform: the window minimum is the van Herk / Gil-Werman scheme with blocks of W elements held
// IN PLACE in (W + 1) x 64 LDS words per wave (slot j of every lane in one 256-byte row: conflict-free).  While block B is
// scanned, slot j still holds the suffix minimum S[j] of block B-1 until element j of block B overwrites it:
//   window ending at element e = B*W + j  =  min(S_{B-1}[j+1], prefix minimum of block B up to j)      (S[W] = +inf);
// at the end of a block one backward pass turns the raw values into suffix minima.  4 LDS operations per element, no
// data-dependent branches (a literal "rescan when the minimizer expires" loop would run on almost every step, because
// among 64 lanes some minimizer nearly always expires).  Steps run in groups of four between the wave-uniform events
// (half-word of bases used up, block complete): the four suffix minima are requested before any of the group's stores, so
// one LDS round trip serves four steps.
constexpr uint32_t SCAN_LANE_MAX = 1024;        // longer reads go to the wave-per-read kernel (a wave lasts as long as its longest read)
__host__ __device__ constexpr uint32_t scan_lane_lds_words(uint32_t W) { return (W + 1) * 64 + 2 * PLC_PACKED; }
template <int NBW, bool DIRECT, bool LAB = false>
__global__ void __launch_bounds__(64) sk_scan_lane_kernel(SeqDev s, FastCfg c, uint32_t* __restrict__ hist,
                                                           uint64_t* __restrict__ tmp_recs, uint32_t* __restrict__ tmp_bin,
                                                           unsigned long long* __restrict__ tmp_cursor, uint64_t tmp_cap,
                                                           uint32_t* __restrict__ flags, uint64_t* __restrict__ slab,
                                                           uint32_t slab_cap, uint32_t* __restrict__ cursor) {
    extern __shared__ uint32_t s_dyn[];
    // one wave per workgroup: the waves share nothing, and 9.5 KB granules pack a CU's LDS (16 waves) better than 38 KB ones (12)
    const uint32_t lane = threadIdx.x;
    const int k = c.k, p = c.p;
    const uint32_t W = (uint32_t)(k - p + 1);
    uint32_t* SV = s_dyn + lane;                                    // slot j of this lane: SV[j * 64]
    uint32_t* PL = s_dyn + (W + 1) * 64;
    SV[W * 64] = 0xffffffffu;                                       // S[W]: the empty suffix
    const uint64_t* __restrict__ w = s.words;
    const uint64_t last_word = s.n_words ? s.n_words - 1 : 0;
    const uint64_t gwave = blockIdx.x, nwaves = gridDim.x;
    const uint64_t lt = lanemask_lt();
    PieceEmitter<NBW, DIRECT, true, LAB> E{s, c, hist, tmp_recs, tmp_bin, tmp_cursor, tmp_cap, flags, slab, slab_cap, cursor, PL, lane, lt, last_word};
    const uint32_t pmask = (1u << (2 * p)) - 1u, top = 2u * (uint32_t)(p - 1);      // p <= 15
    const bool stranded = c.stranded != 0;

    for (uint64_t rb = gwave * 64; rb < s.n; rb += nwaves * 64) {
        const uint64_t my = rb + lane;
        uint32_t v_m = 0, v_ex = 0, v_d1 = 0;
        uint64_t v_st = 0;
        if (my < s.n) {
            v_m = s.length[my]; v_st = s.start[my];
            if (s.exts) v_ex = s.exts[my];
            if (s.data) v_d1 = s.data_width == 1 ? ((const uint8_t*)s.data)[my] : (s.data_width == 2 ? ((const uint16_t*)s.data)[my] : ((const uint32_t*)s.data)[my]);
            if (c.lmap) v_d1 = c.lmap[v_d1];
        }
        uint32_t m = v_m;
        if (m > SCAN_LANE_MAX) { atomicOr(&flags[1], 1u); m = 0; }              // left to the wave-per-read kernel
        if (m < (uint32_t)k) m = 0;
        const uint32_t nwin = m ? m - (uint32_t)k + 1 : 0u;
        uint32_t mmax = m;
#pragma unroll
        for (int d = 32; d; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)mmax, d); mmax = o > mmax ? o : mmax; }
        mmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)mmax);
        if (mmax == 0) continue;
        const uint64_t w_first = v_st >> 5;
        const uint64_t* __restrict__ wr = w + w_first;
        const uint32_t sb2 = (uint32_t)(v_st & 31) * 2u;
        const uint32_t last_rel = (uint32_t)(last_word - w_first < 0x7fffffffull ? last_word - w_first : 0x7fffffffull);
        uint64_t prev = wr[0];
        uint32_t widx = 1;
        uint32_t fw = 0, rc = 0, P = 0xffffffffu, cur_val = 0, ps = 0, cur = 0, lo_half = 0;
        uint32_t j = 0;                                                           // slot of the next element (wave-uniform)

        // one base: roll both strands' p-mers
        auto roll = [&]() {
            const uint32_t b = cur >> 30;
            cur <<= 2;
            fw = ((fw << 2) | b) & pmask;
            rc = (rc >> 2) | ((3u - b) << top);
        };
        // element of this lane: ordering hash of the canonical p-mer.  Past the read's end (shorter reads of the batch) the
        // values are garbage, which is harmless: a window sw < nwin only covers elements sw .. sw + W - 1 <= npos - 1, and both
        // halves of its minimum (suffix of the previous block, prefix of the current one) lie inside the window.
        auto elem = [&]() -> uint32_t { return mix_pmer(stranded ? fw : (fw < rc ? fw : rc), p); };
        // window sw (wave-uniform index) has minimum `out`: piece boundaries
        auto window = [&](uint32_t sw, uint32_t out) {
            const bool isb = sw < nwin && (sw == 0 || out != cur_val || sw - ps == W);
            E.push(isb && sw > 0, ps, sw, cur_val, lane);                         // the boundary closes the piece that was open
            ps = isb ? sw : ps;
            cur_val = isb ? out : cur_val;
        };

        for (uint32_t t = 0; t < mmax;) {
            if ((t & 31u) == 0) {                                                 // the next 32 bases of every lane's read, left-aligned
                const uint64_t nw = wr[widx < last_rel ? widx : last_rel];
                widx++;
                const uint64_t chunk = (prev << sb2) | ((nw >> 1) >> (63u - sb2));
                prev = nw;
                cur = (uint32_t)(chunk >> 32);
                lo_half = (uint32_t)chunk;
            } else if ((t & 15u) == 0) cur = lo_half;
            uint32_t n = 16u - (t & 15u);                                         // steps until the next wave-uniform event
            n = mmax - t < n ? mmax - t : n;
            if (t + 1 < (uint32_t)p) {                                            // the first p-mer is not complete yet
                n = (uint32_t)p - 1 - t < n ? (uint32_t)p - 1 - t : n;
                for (uint32_t u = 0; u < n; u++) roll();
                t += n;
                continue;
            }
            n = W - j < n ? W - j : n;
            uint32_t e = t + 1 - (uint32_t)p;                                     // element = p-mer position of the base rolled next
            if (e + 1 < W) {                                                      // the first window is not complete yet: fill block 0
                n = W - 1 - e < n ? W - 1 - e : n;
                for (uint32_t u = 0; u < n; u++) {
                    roll();
                    const uint32_t h = elem();
                    P = h < P ? h : P;
                    SV[(j + u) * 64] = h;
                }
            } else {
                uint32_t u = 0;
                while (u < n) {
                    if (n - u >= 4) {
                        uint32_t* q = SV + (j + u) * 64;
                        const uint32_t s1 = q[64], s2 = q[128], s3 = q[192], s4 = q[256];
                        uint32_t h, out;
                        // (the ring holds 63 + 64 pieces: it is emptied after every window, which lets 16 waves instead of 14 share a CU's LDS)
                        roll(); h = elem(); P = h < P ? h : P; out = s1 < P ? s1 : P; q[0] = h;   window(e + u + 1 - W, out);
                        while (E.pl_n >= 64) E.flush(64u, v_m, v_st, v_ex, v_d1);
                        roll(); h = elem(); P = h < P ? h : P; out = s2 < P ? s2 : P; q[64] = h;  window(e + u + 2 - W, out);
                        while (E.pl_n >= 64) E.flush(64u, v_m, v_st, v_ex, v_d1);
                        roll(); h = elem(); P = h < P ? h : P; out = s3 < P ? s3 : P; q[128] = h; window(e + u + 3 - W, out);
                        while (E.pl_n >= 64) E.flush(64u, v_m, v_st, v_ex, v_d1);
                        roll(); h = elem(); P = h < P ? h : P; out = s4 < P ? s4 : P; q[192] = h; window(e + u + 4 - W, out);
                        u += 4;
                    } else {
                        uint32_t* q = SV + (j + u) * 64;
                        const uint32_t s1 = q[64];
                        roll();
                        const uint32_t h = elem();
                        P = h < P ? h : P;
                        const uint32_t out = s1 < P ? s1 : P;
                        q[0] = h;
                        window(e + u + 1 - W, out);
                        u++;
                    }
                    while (E.pl_n >= 64) E.flush(64u, v_m, v_st, v_ex, v_d1);
                }
            }
            t += n;
            j += n;
            if (j == W) {                                                         // block complete: raw values -> suffix minima, in place
                uint32_t run = 0xffffffffu;
                uint32_t q = W;
                for (; q >= 5; q -= 4) {                                          // slots q-1 .. q-4 (slot 0 is never read back)
                    uint32_t* r = SV + (q - 4) * 64;
                    uint32_t x3 = r[192], x2 = r[128], x1 = r[64], x0 = r[0];
                    x3 = x3 < run ? x3 : run; x2 = x2 < x3 ? x2 : x3; x1 = x1 < x2 ? x1 : x2; x0 = x0 < x1 ? x0 : x1;
                    r[192] = x3; r[128] = x2; r[64] = x1; r[0] = x0;
                    run = x0;
                }
                for (; q >= 2; q--) {
                    const uint32_t x = SV[(q - 1) * 64];
                    run = x < run ? x : run;
                    SV[(q - 1) * 64] = run;
                }
                j = 0;
                P = 0xffffffffu;
            }
        }
        E.push(nwin > 0, ps, nwin, cur_val, lane);                                // every read's last piece
        while (E.pl_n >= 64) E.flush(64u, v_m, v_st, v_ex, v_d1);
        if (E.pl_n) E.flush(E.pl_n, v_m, v_st, v_ex, v_d1);                       // the lanes' read metadata changes with the next batch
    }
}

template <int RW>
__global__ void __launch_bounds__(256) sk_scatter_kernel(const uint64_t* __restrict__ tmp_recs, const uint32_t* __restrict__ tmp_bin,
                                                         uint64_t n_recs, const uint64_t* __restrict__ bin_off,
                                                         uint32_t* __restrict__ cursor, uint64_t* __restrict__ recs) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_recs) return;
    uint32_t b = tmp_bin[i];
    if (b == BIN_INVALID) return;
    uint64_t r = bin_off[b] + atomicAdd(&cursor[b], 1u);
    const uint64_t* src = tmp_recs + i * RW;
    uint64_t* dst = recs + r * RW;
    if (RW % 2 == 0) {                                   // 32-byte records: two 16-byte moves
#pragma unroll
        for (int q = 0; q < RW / 2; q++) ((ulonglong2*)dst)[q] = ((const ulonglong2*)src)[q];
    } else {
#pragma unroll
        for (int q = 0; q < RW; q++) dst[q] = src[q];
    }
}

// ------------------------------------------------------------------------------------------------
// per-bin LDS hash-table counter
// ------------------------------------------------------------------------------------------------
struct FastOut {
    uint64_t* hi;       // null when k <= 32
    uint64_t* lo;
    uint32_t* pay;      // CountFilter: exts | min(count, 65535) << 8;  CountFilterSet: exts | colour mask << 8 (labels < 24)
    uint4* rec16;       // k <= 48: the same as one 16-byte record {key 0..31, 32..63, 64..95, pay} (hi/lo/pay unused)
    // colour sets of more than 24 colours (WIDE): the payload is the record's own output position, Exts and the 64-bit colour
    // mask wait in these side arrays at that position and are gathered after the sort
    uint4* w_rec;       // {colours 0..31, colours 32..63, Exts, 0}
    // report_all_kmers (filter.rs:208-212): every distinct k-mer, valid or not; null = not requested
    uint64_t* all_hi;
    uint64_t* all_lo;
    unsigned long long* all_cursor;
    uint64_t all_cap;
};

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ __forceinline__ uint32_t fmix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
// 64-bit hash of a key: low bits -> bucket, bits 16.. -> pass selection, high word -> tag.
// Two 64-bit multiplies (= 2 x three quarter-rate 32-bit multiplies on this hardware).  Round 1 found a third mixing round free
// because the kernel waited on LDS; since round 2 it is bound by instruction issue, and dropping the third round is worth 1.8 %
// with the same probe statistics.  DBG_HASH_1MUL (measurement): the high word folded in by rotations, one multiply.
__device__ __forceinline__ uint64_t hash_key(uint64_t hi, uint64_t lo) {
#if defined(DBG_HASH_FOLD)
    // Round 6 (measurement): the key's four 32-bit words folded by rotations into one word x (3-4 full-rate instructions), ONE
    // 32 x 32 -> 64 multiply (v_mad_u64_u32, quarter rate), its halves xor-ed for the bucket / pass bits; the tag is x itself -- it
    // only has to tell the keys of a bucket apart, and the 128-bit key compare guards every match.  12 issue slots instead of 30.
    const uint32_t x = (uint32_t)lo ^ rotl32((uint32_t)(lo >> 32), 11) ^ rotl32((uint32_t)hi, 22) ^ rotl32((uint32_t)(hi >> 32), 5);
    const uint64_t p = (uint64_t)x * 0x9E3779B1ull;
    const uint32_t m = (uint32_t)p ^ (uint32_t)(p >> 32);
    return ((uint64_t)(x ^ (x >> 15)) << 32) | m;
#elif defined(DBG_HASH_1MUL)
    uint64_t h = lo ^ ((hi << 21) | (hi >> 43)) ^ (hi >> 7);
    h ^= h >> 29; h *= 0xD6E8FEB86659FD93ull;
    h ^= h >> 32;
    return h;
#else
    uint64_t h = lo ^ (hi * 0x9E3779B97F4A7C15ull);
    h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull;
    h ^= h >> 32;
    return h;
#endif
}

// Inclusive prefix sum over the 64 lanes of a wave with DPP row shifts and broadcasts: seven VALU instructions, no LDS
// traffic (the shuffle-based form issues six ds_bpermute through the LDS crossbar, which the probing waves of the other
// workgroup on the CU are saturating).
__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v) {
    // row_shr:1..3 combine within groups of four, row_shr:4/8 within a row of 16, row_bcast:15/31 across rows
    const uint32_t a = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    const uint32_t b = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    const uint32_t c = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, false);
    v = v + a + b + c;
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xe, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xc, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

// Workgroup barrier that waits for this wave's LDS operations only.  __syncthreads() also waits for outstanding global loads
// (s_waitcnt vmcnt(0)), which would expose the latency of the record prefetch at the first barrier after it is issued.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NT>
__device__ __forceinline__ uint32_t block_inclusive_scan(uint32_t v, uint32_t* s_wsum, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = wave_inclusive_scan_u32(v);
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; w++) {
        uint32_t x = s_wsum[w];
        if (w < wave) base += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return base + incl;
}

// the same over four 16-bit counters packed in a u64 (no field may reach 65536 in the block total): the two 32-bit halves hold
// two fields each and never carry into one another, so they are scanned as two independent words
template <int NT>
__device__ __forceinline__ uint64_t block_inclusive_scan64(uint64_t v, uint64_t* s_wsum, uint64_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t incl = (uint64_t)wave_inclusive_scan_u32((uint32_t)v) | ((uint64_t)wave_inclusive_scan_u32((uint32_t)(v >> 32)) << 32);
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    uint64_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; w++) {
        uint64_t x = s_wsum[w];
        if (w < wave) base += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return base + incl;
}

// One workgroup (NT threads = NT/64 waves) owns one bin and one LDS hash table of T entries.
// The workgroup streams the bin in batches of NT records, cut into chunks of <= 4 k-mers that are dealt to
// all lanes (see "stream the bin" below):
//   * insert = bucketised linear probing on 32-bit tags (4 per bucket).  A free slot is claimed with
//     CAS(tag, 0, tag|BUSY); the claimer writes the 64/128-bit key and then stores the final tag
//     (LDS operations of one wave are performed in order, so the key is visible before the tag flips).
//     A lane that meets its own tag with BUSY set re-reads it; this never deadlocks: a claimer in
//     another wave progresses independently, and a claimer in the same wave has executed its key/tag
//     stores before the loop's next iteration starts;
//   * count += 1, exts |= e, colour mask |= 1 << d with LDS atomics on the slot.
// A pass that runs out of slots (or is more than 7/8 full with batches still to come) is re-split by hash
// (work stack below).
// Measured on MI355X (profiles/): per-lane rounds 1.08 per k-mer (5.9 % lost CAS races, 1.4 % next bucket,
// 1 % BUSY re-reads); the kernel is VALU-bound (~65 % of SIMD cycles) once the probe is kept to one
// dependent LDS access per branch.
constexpr uint32_t TAG_BUSY = 0x80000000u;
#ifndef DBG_SKIP_EMIT
#define DBG_SKIP_EMIT 0
#endif

#ifdef DBG_COUNT_STATS
#define STAT(i, v) atomicAdd(&s_stat[i], (uint32_t)(v))
#else
#define STAT(i, v) do {} while (0)
#endif
#ifdef DBG_PHASE_TIMES
__device__ unsigned long long g_phase_cycles[8];
#define PH(i) do { if (tid == 0) { unsigned long long now_ = wall_clock64(); atomicAdd(&g_phase_cycles[i], now_ - t_prev_); t_prev_ = now_; } } while (0)
#else
#define PH(i) do {} while (0)
#endif
// DBG_ABL_COUNT = n (measurement builds, tools/abl_count.sh): the counting kernel stops after phase n of a bin -- 1 segment bounds,
// 2 table cleared, 3 records staged, 4 chunk map built, 5 chunks set up (no k-mer is rolled or inserted)
#ifndef DBG_ABL_COUNT
#define DBG_ABL_COUNT 0
#endif
#ifdef DBG_ASM_MARKS     // phase markers in the assembly listing
#define MARK(t) asm volatile("; ##MARK " t ::: "memory")
#else
#define MARK(t) do {} while (0)
#endif
template <int KW, int NBW, bool IS_SET, int NT, int T, bool WIDE = false, bool WEIGHTED = false>
__device__ __forceinline__ void bin_count_body(const uint32_t bin_idx, const uint64_t* __restrict__ recs, const uint64_t* __restrict__ recs_alt, uint32_t alt_from,
                                                       const uint64_t* __restrict__ seg_beg, const uint64_t* __restrict__ seg_end,
                                                       uint32_t n_src, uint64_t seg_stride,
                                                       int k, int stranded, uint64_t min_obs, FastOut out, uint64_t out_cap,
                                                       unsigned long long* __restrict__ out_cursor, uint32_t* __restrict__ gflags) {
    // WEIGHTED (sharded flow): a record may stand for several copies, merged on the sending rank (slab_merge_kernel)
    constexpr uint32_t wmask = WEIGHTED ? (1u << WEIGHT_BITS) - 1u : 0u;
    constexpr int RW = NBW;
    constexpr int NWV = NT / 64;
    __shared__ __attribute__((aligned(16))) uint32_t s_tag[T];
    __shared__ __attribute__((aligned(16))) uint64_t s_key[KW * T];   // KW == 2: {lo, hi} pairs, one ds_read_b128 per entry
    __shared__ uint32_t s_cnt[T];
    __shared__ uint32_t s_aux[T];               // Exts | colour mask << 8 (CountFilterSet labels < 24); WIDE: colours 0..31
    static_assert(!WIDE || IS_SET, "WIDE is a colour-set layout");
    __shared__ uint32_t s_aux2[WIDE ? T : 1];   // WIDE: colours 32..63
    __shared__ uint32_t s_ex4[WIDE ? T / 4 : 1];   // WIDE: Exts, one byte per entry
    typedef typename std::conditional<WIDE, unsigned long long, uint32_t>::type cmask_t;
    constexpr uint32_t CAPC = (NBW == 4 || WIDE ? 4 : 6) * NT;      // chunk-map capacity per batch
    __shared__ uint64_t s_slab[RW * NT];        // staged batch of records, word-major
    __shared__ __attribute__((aligned(4))) uint16_t s_cmap[CAPC];   // chunk -> record slot | chunk index << 10
    constexpr uint32_t DD = NT >= 512 ? 1024 : 512;                 // slots of the duplicate filter (>= 2 per staged record)
    static_assert(CAPC * 2 >= DD * 4 && DD >= 2 * NT && NT <= 1023, "the duplicate filter borrows s_cmap");
    uint32_t* const s_dd = reinterpret_cast<uint32_t*>(s_cmap);      // duplicate filter of the batch (dead before s_cmap is filled)
    __shared__ uint32_t s_w[NT / 2];            // per staged record: how many identical records of the batch it stands for (u16 halves)
    __shared__ cmask_t s_cmk[IS_SET ? NT : 1];  // ... and the union of their colours (CountFilterSet)
    __shared__ uint32_t s_m, s_cproc, s_nextq, s_nst, s_bad;
#ifdef DBG_COUNT_STATS
    __shared__ uint32_t s_stat[16];
    if (threadIdx.x < 16) s_stat[threadIdx.x] = 0;
#endif
    __shared__ uint32_t s_wsum[NWV];
    __shared__ uint64_t s_wsum64[NWV];
    __shared__ uint32_t s_flag[2];              // [0] table overflow, [1] claimed entries
    __shared__ unsigned long long s_base, s_base_all;
    __shared__ uint32_t s_st[24];               // work stack of passes: P | r << 16 (P <= 4096: at most 13 levels, one pending sibling each)

    const uint32_t tid = threadIdx.x, lane = tid & 63;
#ifdef DBG_PHASE_TIMES
    unsigned long long t_prev_ = wall_clock64();
#endif
    // The bin's records arrive as n_src segments (one per source rank after the all-to-all; the bin's slab plus
    // its overflow in the single-GPU case): segment s spans records [seg_beg[s*stride + bin], seg_end[s*stride + bin])
    // of `recs` (of `recs_alt` for s >= alt_from).
    __shared__ uint32_t s_segpre[66];           // records of this bin before segment s (flat index space)
    // first record of segment s (so that a record fetch is one global round trip, not two); the one variant whose LDS has no
    // 512 bytes left for it (k >= 56 with colour sets) reads the segment table again instead
    constexpr bool HAVE_SEGBEG = !(NBW == 4 && IS_SET && NT == 512);
    __shared__ uint64_t s_segbeg[HAVE_SEGBEG ? 64 : 1];
    if (tid < 64) {                             // n_src <= 64: one lane per segment, prefix sum by shuffles
        uint32_t len = 0;
        if (tid < n_src) {
            const uint64_t a = seg_beg[tid * seg_stride + (uint64_t)bin_idx * NCLS], b = seg_end[tid * seg_stride + (uint64_t)(bin_idx + 1) * NCLS - 1];
            len = (uint32_t)(b - a);
            if (HAVE_SEGBEG) s_segbeg[tid] = a;
        }
        uint32_t incl = len;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d);
            if (lane >= (uint32_t)d) incl += o;
        }
        if (tid < n_src) s_segpre[tid] = incl - len;
        if (tid == 63) s_segpre[n_src] = incl;
    }
    __syncthreads();
    const uint32_t total_recs = s_segpre[n_src];
    PH(0); MARK("prologue_done");
    if (total_recs == 0 || DBG_ABL_COUNT == 1) return;
    const K128 kmask = k128_mask(k);

    // Work stack of hash-selected passes (P, r): the pass handles the keys with (hash >> 16) % P == r.
    // A pass whose distinct keys overflow the table emits nothing and is replaced by its two children
    // (2P, r) and (2P, r + P), which partition exactly its key set.
    // The stack pointer is kept by every thread (all see the same overflow decisions); only the entries live in LDS, written by
    // one thread when a pass is split and read after the barrier that ends the pass.  A bin spends its time in a sequence of
    // barrier-separated phases: every barrier saved counts (the first pass needs no stack access at all).
    int sp = 1;
    for (uint32_t guard = 0;; guard++) {
        if (sp == 0) break;
        if (guard > 20000u) { if (tid == 0) atomicOr(&gflags[3], 2u); break; }                   // watchdog
        const uint32_t ent = guard == 0 ? 1u : s_st[sp - 1];
        const uint32_t P = ent & 0xffffu, pr = ent >> 16;
        sp--;
        if (tid == 0 && P > 1) { atomicMax(&gflags[1], P); atomicAdd(&gflags[2], 1u); }
        constexpr uint32_t CH = 4;
        // (scalars, not arrays: an indexed private array is placed in scratch memory by the compiler)
        uint64_t W0 = 0, W1 = 0, W2 = 0, W3 = 0, P0 = 0, P1 = 0, P2 = 0, P3 = 0;
        uint64_t meta = 0, pmeta = 0;
        auto base_at = [&](uint32_t q) -> uint32_t {
            uint64_t wd = q < 32 ? W0 : (q < 64 ? W1 : (NBW > 2 && q < 96 ? W2 : (NBW > 3 ? W3 : (NBW > 2 ? W2 : W1))));
            return (uint32_t)(wd >> (62 - 2 * (q & 31))) & 3u;
        };
        auto load_into = [&](uint32_t ridx, uint64_t& A0, uint64_t& A1, uint64_t& A2, uint64_t& A3, uint64_t& am) {   // zero when past the end
            A0 = A1 = A2 = A3 = 0;
            am = 0;
            if (ridx < total_recs) {
                uint32_t sg = 0;
                while (sg + 1 < n_src && ridx >= s_segpre[sg + 1]) sg++;
                const uint64_t sbase = HAVE_SEGBEG ? s_segbeg[sg] : seg_beg[sg * seg_stride + (uint64_t)bin_idx * NCLS];   // first record of the segment
                const uint64_t* g = (sg >= alt_from ? recs_alt : recs) + (sbase + (ridx - s_segpre[sg])) * RW;
                A0 = g[0]; A1 = g[1];
                if (NBW > 2) A2 = g[2];
                if (NBW > 3) A3 = g[3];
                // (the meta bits are taken from the last word where they are used: touching a loaded word here would wait for
                //  the load and expose the latency the prefetch is meant to hide)
            }
        };
        auto load_rec = [&](uint32_t ridx) { load_into(ridx, P0, P1, P2, P3, pmeta); };   // -> P0..P3, pmeta
        // the first two rounds of incoming records are on their way while the table is cleared (a bin normally holds 512 + ~33)
        uint64_t Q0, Q1, Q2, Q3, qmeta;
        load_rec(tid);
        load_into(NT + tid, Q0, Q1, Q2, Q3, qmeta);     // both requests are in flight together
        bool q_valid = true;                            // Q holds record NT + tid
        for (int i = tid; i < T; i += NT) { s_tag[i] = 0; s_cnt[i] = 0; s_aux[i] = 0; if (WIDE) s_aux2[i] = 0; }
        if (WIDE) for (int i = tid; i < T / 4; i += NT) s_ex4[i] = 0;
        if (tid == 0) { s_flag[0] = 0; s_flag[1] = 0; s_bad = 0; s_nst = 0; }
        // (the staging area and the filter of the first fill are reset here too: one barrier for both)
        if (tid < NT / 2) s_w[tid] = 0;
        if (IS_SET) s_cmk[tid] = 0;
        for (uint32_t i = tid; i < DD; i += NT) s_dd[i] = 0;
        lds_barrier();                                  // the prefetched records stay in flight
        PH(1); MARK("cleared");
        if (DBG_ABL_COUNT == 2) break;

        // ---- stream the bin, chunk-parallel.  A super-k-mer record holds 1..W k-mers; handing whole records to
        //      lanes leaves the waves of the workgroup badly balanced (a bin is only ~1.3 records per lane).  So
        //      the workgroup stages a batch of NT records in LDS, cuts every record into ceil(nk/CH) nearly equal
        //      chunks (block prefix sum -> chunk map), and deals the chunks round-robin to all NT lanes.  A lane
        //      re-creates the chunk's first k-mer from the staged record (one funnel shift + one reverse
        //      complement per chunk) and then rolls (extend_right on the forward strand, extend_left of the
        //      complement on the reverse strand).  The next batch is prefetched into registers meanwhile. ----
        bool pass_ovf = false;
        // Staged records persist across rounds of incoming records: DISTINCT records collect in the staging area (identical
        // ones only add to a staged record's weight and colour set), and their k-mers are inserted once the area is nearly
        // full or the bin is exhausted.  A bin of ~545 records holds ~230 distinct ones, so it is normally filled in two rounds
        // (512 + 33 incoming records) and processed ONCE -- the tail batch used to pay for its own staging, chunk map and
        // insertion round, and its records could not be merged with the first batch's.
        //   s_dd   duplicate filter (open addressing; entry = staged slot + 1 | 22 hash bits); lives in s_cmap, which is only
        //          needed once the staged records are cut into chunks -- after that the filter is rebuilt from scratch
        //   s_nst  staged records; a round accepts min(NT - s_nst, remaining) incoming records, so every one of them finds room
        constexpr uint64_t COLOUR_BITS = 63ull << 15;
        constexpr uint32_t MIN_ROOM = NT / 8;           // keep filling while the rest of the bin, or at least this many records, still fit
        // weights are 16-bit: a round adds at most NT x (the largest weight a record carries): 101 x 512 and 7 x 512 x 16 < 65536
        constexpr uint32_t MAX_FILLS = wmask ? 6u : 100u;
        uint32_t rnext = 0;                             // next incoming record of the bin (uniform)
        bool pass_bad = false, bad_bin = false;
        while (rnext < total_recs && !pass_ovf) {
            // ---- reset the staging area and the filter (the first fill's were reset with the table) ----
            if (rnext != 0) {
                if (tid < NT / 2) s_w[tid] = 0;
                if (IS_SET) s_cmk[tid] = 0;
                for (uint32_t i = tid; i < DD; i += NT) s_dd[i] = 0;
                if (tid == 0) s_nst = 0;
                lds_barrier();
            }
            uint32_t nstaged = 0;
            // ---- fill rounds: thread tid holds incoming record rnext + tid (prefetched) ----
            for (uint32_t fills = 0;; fills++) {
                const uint32_t room = NT - nstaged;
                const uint32_t take = total_recs - rnext < room ? total_recs - rnext : room;
                // Claim, barrier, compare -- nobody waits in a loop: the first record of its kind wins its filter slot (CAS), takes
                // the next staging slot, writes the record and publishes the slot number; the others of its kind stop at that
                // filter entry (claimed or published, it carries their hash bits), and after the workgroup's barrier compare
                // themselves with the staged record and add to its weight.  Only a record that met a DIFFERENT record with its
                // hash bits goes round again from the next filter slot.  (The version that re-read a claimed entry until it was
                // published spent a third of the staging phase's lane iterations doing that: a bin's ~18 copies of each piece
                // arrive together.)
                bool pend = tid < take;
                uint32_t mytag = 0, sl = 0;
                cmask_t colour = 0;
                // a record merged on the sending rank stands for 1 + (its weight field) copies; the field is cleared before the record
                // is hashed, compared and staged (wmask = 0: the records carry no weights, the bits may be base bits)
                uint64_t& PLr = NBW == 2 ? P1 : (NBW == 3 ? P2 : P3);                    // word holding the meta bits
                const uint32_t wt = 1u + ((uint32_t)(PLr >> META_BITS) & wmask);
                PLr &= ~((uint64_t)wmask << META_BITS);
                const uint64_t PL0 = PLr;
                if (pend) {
                    pmeta = PL0 & ((1ull << META_BITS) - 1);
                    {   // a record whose length cannot come from the scan (a wrong segment table, an incomplete exchange) must not
                        // be expanded: its k-mer count would be garbage.  The launch is failed instead.
                        const uint32_t rl = (uint32_t)(pmeta & 0x7f);
                        if (rl < (uint32_t)k || rl > (uint32_t)(32 * NBW - (META_BITS + 1) / 2)) { atomicOr(&gflags[3], 4u); pass_bad = true; }
                    }
                    const uint64_t lastw = PL0 & ~COLOUR_BITS;
                    uint64_t ha = P0, hb = NBW == 2 ? lastw : P1;
                    if (NBW == 3) ha += lastw * 0x9E3779B97F4A7C15ull;
                    if (NBW == 4) { ha += P2 * 0x9E3779B97F4A7C15ull; hb += lastw * 0xC2B2AE3D27D4EB4Full; }
                    const uint64_t h = hash_key(ha, hb);
                    mytag = (uint32_t)(h >> 42) << 10;
                    colour = (cmask_t)1 << ((uint32_t)(pmeta >> 15) & (WIDE ? 63u : 31u));
                    sl = (uint32_t)h & (DD - 1);
                }
                bool bad_round = false;
                for (;;) {
                    if (pend) {
                        constexpr uint32_t CLAIMED = 1023u;
                        for (;;) {                                                       // to the first slot that is free or carries my hash bits
                            uint32_t v = s_dd[sl];
                            if (v == 0u) {
                                v = atomicCAS(&s_dd[sl], 0u, CLAIMED | mytag);
                                if (v == 0u) {                                           // first of its kind: stage it
                                    const uint32_t mine = atomicAdd(&s_nst, 1u);         // < NT: at most `room` records allocate
                                    s_slab[mine] = P0; s_slab[NT + mine] = P1;
                                    if (NBW > 2) s_slab[2 * NT + mine] = P2;
                                    if (NBW > 3) s_slab[3 * NT + mine] = P3;
                                    atomicAdd(&s_w[mine >> 1], wt << (16 * (mine & 1u)));
                                    if (IS_SET) atomicOr(&s_cmk[mine], colour);
                                    s_dd[sl] = (mine + 1u) | mytag;
                                    pend = false;
                                    break;
                                }
                            }
                            if ((v & ~1023u) == mytag) break;                            // (being) staged here: compared after the barrier
                            sl = (sl + 1u) & (DD - 1);
                        }
                    }
                    lds_barrier();
                    if (pend) {
                        const uint32_t r = (s_dd[sl] & 1023u) - 1u;
                        bool same = s_slab[r] == P0;
                        if (NBW > 2) same = same && s_slab[NT + r] == P1;
                        if (NBW > 3) same = same && s_slab[2 * NT + r] == P2;
                        same = same && ((s_slab[(NBW - 1) * NT + r] ^ PL0) & ~COLOUR_BITS) == 0;
                        if (same) {                                                      // an equal record is staged: it stands for this one too
                            STAT(fills ? 14 : 15, 1);
                            atomicAdd(&s_w[r >> 1], wt << (16 * (r & 1u)));
                            if (IS_SET) atomicOr(&s_cmk[r], colour);
                            pend = false;
                        } else sl = (sl + 1u) & (DD - 1);
                    }
                    if (pass_bad) s_bad = 1u;
                    const int more = __syncthreads_or(pend ? 1 : 0);
                    if (s_bad) { bad_round = true; break; }
                    if (!more) break;
                }
                if (bad_round) { pass_ovf = true; bad_bin = true; nstaged = 0; rnext = total_recs; break; }     // corrupt input: give up on the bin
                nstaged = s_nst;
#ifdef DBG_COUNT_STATS
                if (tid == 0 && fills == 0) atomicAdd(&s_stat[13], nstaged);
                if (tid == 0 && fills == 1) atomicAdd(&s_stat[12], take);
#endif
                rnext += take;
                if (q_valid && rnext == (uint32_t)NT) { P0 = Q0; P1 = Q1; P2 = Q2; P3 = Q3; pmeta = qmeta; }   // requested at the start
                else load_rec(rnext + tid);                                              // the next incoming records are on their way
                q_valid = false;
                if (rnext >= total_recs || fills >= MAX_FILLS) break;
                if (NT - nstaged < MIN_ROOM && NT - nstaged < total_recs - rnext) break;
            }
            PH(7); MARK("filled");
            if (DBG_ABL_COUNT == 3) continue;
            // ---- cut the staged records into chunks and insert their k-mers; the chunk map may take several rounds when the
            //      records are long and all different ----
            for (uint32_t base = 0; base < nstaged && !pass_ovf;) {
                const bool have = tid >= base && tid < nstaged;
                uint32_t nkr = 0;
                if (have) {
                    const uint32_t wg = (s_w[tid >> 1] >> (16 * (tid & 1u))) & 0xffffu;
                    const uint32_t rl = (uint32_t)(s_slab[(NBW - 1) * NT + tid] & 0x7f);
                    nkr = wg ? rl - (uint32_t)k + 1u : 0u;
#ifdef DBG_ABL_DROP_CUT   // measurement only (WRONG counts): pieces cut by a read end are not expanded -- the bound on what merging them into their whole piece can save
                    { const uint32_t rx = (uint32_t)(s_slab[(NBW - 1) * NT + tid] >> 7) & 0xffu; if (((rx & 0xfu) == 0u) != ((rx & 0xf0u) == 0u)) nkr = 0u; }
#endif
                }
                if (tid == 0) { s_m = nstaged; s_cproc = 0; s_nextq = 0; }
                // Chunks are entered into the map by length (CH, CH-1, ..., 1): the 64 chunks a wave takes then roll the
                // same number of k-mers, where record order mixes lengths cb and cb+1 in every wave (11 % idle lanes).
                // One scan of four packed 16-bit counters gives every record its place in both of its length classes.
                const uint32_t nch = (nkr + CH - 1) / CH;
                const uint32_t cb = nch ? nkr / nch : 0u, cr = nkr - cb * nch;    // chunk c: cb + (c < cr) k-mers
                const uint32_t cl_hi = CH - cb - 1, cl_lo = CH - cb;             // class of the cr longer / nch-cr shorter chunks
                uint64_t contrib = 0;
                if (nch) contrib = (cr ? (uint64_t)cr << (16 * cl_hi) : 0ull) + ((uint64_t)(nch - cr) << (16 * cl_lo));
                uint64_t tot64;
                const uint64_t incl64 = block_inclusive_scan64<NT>(contrib, s_wsum64, &tot64);   // barriers inside
                auto fsum = [](uint64_t x) { return (uint32_t)(x & 0xffff) + (uint32_t)((x >> 16) & 0xffff) + (uint32_t)((x >> 32) & 0xffff) + (uint32_t)(x >> 48); };
                const uint32_t totc = fsum(tot64);
                const bool all_fit = totc <= CAPC;                               // uniform
                if (all_fit) {
                    if (nch) {
                        const uint64_t excl64 = incl64 - contrib;
                        // class c starts after the totals of classes 0..c-1
                        const uint64_t below_lo = tot64 & ((1ull << (16 * cl_lo)) - 1);
                        const uint32_t pos_lo = fsum(below_lo) + (uint32_t)((excl64 >> (16 * cl_lo)) & 0xffff);
                        for (uint32_t c = cr; c < nch; c++) s_cmap[pos_lo + c - cr] = (uint16_t)(tid | (c << 10));
                        if (cr) {
                            const uint64_t below_hi = tot64 & ((1ull << (16 * cl_hi)) - 1);
                            const uint32_t pos_hi = fsum(below_hi) + (uint32_t)((excl64 >> (16 * cl_hi)) & 0xffff);
                            for (uint32_t c = 0; c < cr; c++) s_cmap[pos_hi + c] = (uint16_t)(tid | (c << 10));
                        }
                    }
                } else {
                    // the chunks of the remaining staged records do not fit the map (long records, few copies): as many records
                    // as fit are taken in slot order, the rest in the next round
                    if (tid == 0) atomicAdd(&gflags[12], 1u);
                    uint32_t dummy;
                    const uint32_t incl = block_inclusive_scan<NT>(nch, s_wsum, &dummy);
                    if (incl <= CAPC) {
                        if (nch) atomicMax(&s_cproc, incl);
                        for (uint32_t c = 0; c < nch; c++) s_cmap[incl - nch + c] = (uint16_t)(tid | (c << 10));
                    } else if (have) atomicMin(&s_m, tid);
                }
                __syncthreads();
                MARK("map_done");
                if (DBG_ABL_COUNT == 4) { base = nstaged; continue; }
                const uint32_t mend = all_fit ? nstaged : s_m;                   // records [base, mend) are in the map (at least one)
                const uint32_t cproc = all_fit ? totc : s_cproc;
                // C. chunks, 64 at a time to whichever wave is free (a wave's rounds differ in length: probe retries, chunk sizes)
                for (;;) {
                    uint32_t q0 = 0;
                    if (lane == 0) q0 = atomicAdd(&s_nextq, 64u);
                    q0 = __shfl(q0, 0);
                    if (q0 >= cproc) break;
                    const uint32_t q = q0 + lane;
                    const bool act = q < cproc;
                    const uint32_t e = act ? (uint32_t)s_cmap[q] : 0u;
                    const uint32_t r = e & 1023u, c = e >> 10;
                    W0 = s_slab[r]; W1 = s_slab[NT + r];
                    if (NBW > 2) W2 = s_slab[2 * NT + r];
                    if (NBW > 3) W3 = s_slab[3 * NT + r];
                    {   // the record's meta bits sit below its bases in the last word
                        uint64_t& WL = NBW == 2 ? W1 : (NBW == 3 ? W2 : W3);
                        meta = WL & ((1ull << META_BITS) - 1);
                        WL &= ~((1ull << META_BITS) - 1);
                    }
                    const uint32_t rlen = (uint32_t)(meta & 0x7f), rexts = (uint32_t)(meta >> 7) & 0xffu;
                    const uint32_t wgt = (s_w[r >> 1] >> (16 * (r & 1u))) & 0xffffu;
                    const cmask_t cmr = IS_SET ? s_cmk[r] : (cmask_t)0;
                    const uint32_t cset = WIDE ? (uint32_t)cmr : (uint32_t)cmr << 8, cset2 = WIDE ? (uint32_t)((unsigned long long)cmr >> 32) : 0u;
                    // the record's chunking, as stage A cut it: ceil(nk/4) chunks of cbase (+1 for the first crem) k-mers
                    static_assert(CH == 4, "closed form of nk / ceil(nk / CH)");
                    const uint32_t rnk = rlen - (uint32_t)k + 1u, rnch = (rnk + 3u) >> 2;
                    const uint32_t cbase = rnk < 4u ? rnk : (rnk == 5u ? 2u : ((rnk & 3u) ? 3u : 4u)), crem = rnk - cbase * rnch;
                    uint32_t j = c * cbase + (c < crem ? c : crem);
                    const uint32_t jend = act ? j + cbase + (c < crem ? 1u : 0u) : j;
                    // k-mer j of the record: bases [j, j + k) of the 2-bit stream W[0..NBW)
                    K128 fw;
                    {
                        const uint32_t sft = 2 * j, ws = sft >> 6, bs = sft & 63;
                        const uint64_t A = ws == 0 ? W0 : W1, B = ws == 0 ? W1 : (NBW > 2 ? W2 : 0ull),
                                       C = ws == 0 ? (NBW > 2 ? W2 : 0ull) : (NBW > 3 ? W3 : 0ull);
                        const uint64_t h = bs ? (A << bs) | (B >> (64 - bs)) : A, l = bs ? (B << bs) | (C >> (64 - bs)) : B;
                        fw = k128_shr(K128{h, l}, 128 - 2 * k);
                    }
                    K128 rcw = kmer_rc(fw, k);
                    uint32_t lb = j ? base_at(j - 1) : 0u;
                    // the (at most CH) bases that follow the chunk's first k-mer, top-aligned; zero beyond the record's end
                    uint32_t nx;
                    {
                        const uint32_t sft = 2 * (j + (uint32_t)k), ws = sft >> 6, bs = sft & 63;
                        const uint64_t A = ws == 0 ? W0 : (ws == 1 ? W1 : (NBW > 2 && ws == 2 ? W2 : (NBW > 3 && ws == 3 ? W3 : 0ull)));
                        const uint64_t B = ws == 0 ? W1 : (NBW > 2 && ws == 1 ? W2 : (NBW > 3 && ws == 2 ? W3 : 0ull));
                        const uint64_t v = bs ? (A << bs) | (B >> (64 - bs)) : A;
                        nx = (uint32_t)(v >> 32);
                    }
                    MARK("chunk_setup_done");
                    if (DBG_ABL_COUNT == 5) { if (fw.lo + rcw.hi + lb + nx + wgt + cset + cset2 + rexts == 0x1234567u) s_flag[1] = 7; continue; }
                    while (__any(j < jend)) {
                        const bool alive = j < jend;
#ifdef DBG_COUNT_STATS
                { uint64_t bal = __ballot(alive); if (lane == 0) { atomicAdd(&s_stat[4], 1u); atomicAdd(&s_stat[5], (uint32_t)__popcll(bal)); } }
#endif
                if (alive) {
                    const uint32_t nbase = nx >> 30;                                         // base right of the k-mer (0 past the end)
                    nx <<= 2;
                    {
                        // Exts of k-mer j inside the piece (lib.rs:820-832 with seq_exts = the piece's boundary Exts)
                        uint32_t left = j == 0 ? (rexts & 0xfu) : (1u << lb);
                        uint32_t right = (j + (uint32_t)k == rlen) ? (rexts & 0xf0u) : (16u << nbase);
                        uint32_t ex = left | right;
                        K128 km = fw;
                        if (!stranded && !k128_lt(fw, rcw)) { km = rcw; ex = __brev(ex) >> 24; }   // ties flip (lib.rs:226-230); Exts::rc = byte bit-reversal
                        const uint64_t h = hash_key(km.hi, km.lo);
                        if (P == 1 || ((uint32_t)(h >> 16) & (P - 1)) == pr) {
                            // Bucketised linear probing: 4 tags per 16-byte bucket.  One round = one ds_read_b128 of the
                            // bucket's tags, then at most one dependent LDS operation per lane: the 16-byte key of the one
                            // candidate slot whose tag matches, or a CAS on the first free slot.  (Divergent branches run
                            // one after the other, so every extra dependent LDS access inside a branch costs the whole
                            // wave a round trip; the candidate is therefore chosen with compares and selects only.)
                            // A key lives in the first bucket (in probe order) that had a free slot when it was inserted;
                            // a failed CAS re-reads the bucket, so two lanes can never claim two slots for one key.
                            const uint32_t mytag = ((uint32_t)(h >> 32) & 0x7fffffffu) | 1u;
                            uint32_t bkt = (uint32_t)h & (T / 4 - 1);
                            uint32_t slot = 0, tried = 0, nprobe = 0;
                            const uint32_t rot = (uint32_t)(h >> 12) & 3u;
                            bool hit = false;
                            for (;;) {
                                asm volatile("" ::: "memory");                           // re-read the tags every round
                                const uint4 t4 = *reinterpret_cast<const uint4*>(&s_tag[bkt * 4]);
                                const uint32_t mm = ((t4.x == mytag ? 1u : 0u) | (t4.y == mytag ? 2u : 0u) | (t4.z == mytag ? 4u : 0u) |
                                                     (t4.w == mytag ? 8u : 0u)) & ~tried;
                                const uint32_t bz = mytag | TAG_BUSY;
                                const bool busy = t4.x == bz || t4.y == bz || t4.z == bz || t4.w == bz;
                                const uint32_t em = (t4.x == 0u ? 1u : 0u) | (t4.y == 0u ? 2u : 0u) | (t4.z == 0u ? 4u : 0u) | (t4.w == 0u ? 8u : 0u);
                                STAT(6, 1);
                                if (mm) {                                                // ready entry with my tag: verify the key
                                    const uint32_t i = (uint32_t)__ffs((int)mm) - 1u, sl = bkt * 4 + i;
                                    bool same;
                                    if (KW == 2) {
                                        const ulonglong2 kk = *reinterpret_cast<const ulonglong2*>(&s_key[2 * sl]);
                                        same = kk.x == km.lo && kk.y == km.hi;
                                    } else same = s_key[sl] == km.lo;
                                    if (same) { hit = true; slot = sl; break; }
                                    tried |= 1u << i;                                    // a different key with the same 31-bit tag
                                    STAT(8, 1);
                                    continue;
                                }
                                if (busy) { STAT(9, 1); continue; }                      // a claimer is still writing its key: re-read
                                if (em) {
                                    // first free slot in a cyclic order that depends on the key: lanes that insert DIFFERENT keys into
                                    // one bucket at the same moment then mostly try different slots (with "first free" they all meet on
                                    // the same one and all but one lose a round); two lanes with the SAME key use the same order, so the
                                    // argument above (a failed CAS re-reads the bucket) still keeps a key from getting two slots.
                                    const uint32_t emr = ((em >> rot) | (em << (4u - rot))) & 15u;
                                    const uint32_t sl = bkt * 4 + (((uint32_t)__ffs((int)emr) - 1u + rot) & 3u);
                                    if (atomicCAS(&s_tag[sl], 0u, bz) == 0u) {
                                        if (KW == 2) *reinterpret_cast<ulonglong2*>(&s_key[2 * sl]) = make_ulonglong2(km.lo, km.hi);
                                        else s_key[sl] = km.lo;
                                        asm volatile("" ::: "memory");                   // the key store is issued before the tag store
                                        __hip_atomic_store(&s_tag[sl], mytag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                        atomicAdd(&s_flag[1], 1u);                       // occupancy, polled between batches
                                        hit = true; slot = sl;
                                        STAT(11, 1);
                                        break;
                                    }
                                    STAT(10, 1);
                                    continue;                                            // lost the race: re-read this bucket
                                }
                                bkt = (bkt + 1) & (T / 4 - 1);
                                tried = 0;
                                STAT(7, 1);
                                if (++nprobe >= (uint32_t)(T / 4)) break;                // table full
                            }
                            if (hit) {
                                atomicAdd(&s_cnt[slot], wgt);
                                if (WIDE) {
                                    atomicOr(&s_aux[slot], cset);
                                    if (cset2) atomicOr(&s_aux2[slot], cset2);
                                    atomicOr(&s_ex4[slot >> 2], ex << (8u * (slot & 3u)));
                                } else atomicOr(&s_aux[slot], ex | cset);
                            } else {
                                __hip_atomic_store(&s_flag[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // table full: the pass is re-split
                            }
                        }
                    }
                    // roll to k-mer j+1
                    if (KW == 2) {                                       // 33 <= k <= 64: the first base sits in hi
                        const int sh = 2 * (k - 1) - 64;
                        lb = (uint32_t)(fw.hi >> sh) & 3u;
                        fw.hi = ((fw.hi << 2) | (fw.lo >> 62)) & kmask.hi;
                        fw.lo = (fw.lo << 2) | nbase;
                        rcw.lo = (rcw.lo >> 2) | (rcw.hi << 62);
                        rcw.hi = (rcw.hi >> 2) | ((uint64_t)(3u - nbase) << sh);
                    } else {                                             // k <= 32: everything sits in lo
                        const int sh = 2 * (k - 1);
                        lb = (uint32_t)(fw.lo >> sh) & 3u;
                        fw.lo = ((fw.lo << 2) | nbase) & kmask.lo;
                        rcw.lo = (rcw.lo >> 2) | ((uint64_t)(3u - nbase) << sh);
                    }
                    j++;
                }
                    }   // rolling loop
                    MARK("roll_done");
                }       // chunk loop
                PH(2); MARK("chunks_done");
                __syncthreads();
                PH(3);                 // waiting for the slowest wave
                base = mend;
                // a table more than 7/8 full makes the remaining records probe long chains: give up early and re-split the pass
                // (only when records remain: the last round of a bin needs no second barrier)
                if (base < nstaged || rnext < total_recs) {
                    if (tid == 0 && s_flag[1] > (uint32_t)(T - T / 8)) s_flag[0] = 1;
                    __syncthreads();
                }
                pass_ovf = __hip_atomic_load(&s_flag[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
            }
        }
        if (bad_bin) break;                              // corrupt input: the launch fails (gflags), nothing is emitted
        const bool ovf = pass_ovf;                       // (every thread read the flag after the last barrier above)
        // ---- emit the valid entries of this pass (a pass that overflowed emits nothing) ----
        if (!ovf && !DBG_SKIP_EMIT) {
            // Every wave emits its own T / NWV slots: ballots give the counts and the ranks; the waves' counts meet in LDS, one
            // thread reserves the bin's output range with ONE global atomic per list (same-address device atomics complete at about
            // 10^8 per second: one per wave was measured at twice the kernel time), and each wave writes behind its predecessors.
            constexpr int EMIT_IT = T / NWV / 64;
            const uint32_t wave = tid >> 6;
            const uint32_t slot0 = wave * (T / NWV) + lane;
            uint64_t vb[EMIT_IT], ab[EMIT_IT];
            uint32_t nv = 0, na = 0;
#pragma unroll
            for (int it = 0; it < EMIT_IT; it++) {
                const uint32_t i = slot0 + it * 64;
                const bool occ = s_tag[i] != 0u;
                const uint32_t c = s_cnt[i];
                const bool valid = occ && (IS_SET ? (uint64_t)c >= min_obs : (uint64_t)(c > 65535u ? 65535u : c) >= min_obs);
                vb[it] = __ballot(valid); ab[it] = __ballot(occ);
                nv += (uint32_t)__popcll(vb[it]); na += (uint32_t)__popcll(ab[it]);
            }
            if (lane == 0) s_wsum[wave] = nv | (na << 16);                   // each <= T / NWV
            __syncthreads();
            PH(4);
            uint32_t before = 0, total = 0;
#pragma unroll
            for (int w = 0; w < NWV; w++) { const uint32_t x = s_wsum[w]; before += (uint32_t)w < wave ? x : 0u; total += x; }
            const uint32_t tot_valid = total & 0xffffu, tot_all = total >> 16;
            if (tid == 0) {
                s_base = tot_valid ? atomicAdd(out_cursor, (unsigned long long)tot_valid) : 0ull;
                s_base_all = (out.all_lo && tot_all) ? atomicAdd(out.all_cursor, (unsigned long long)tot_all) : 0ull;
            }
            __syncthreads();
            PH(5);
            const unsigned long long base = s_base + (before & 0xffffu), base_all = s_base_all + (before >> 16);
            nv = tot_valid; na = tot_all;                                     // the fit test is about the whole bin
            const bool fit = !(nv && s_base + nv > out_cap);
            const bool fit_all = !(out.all_lo && na && s_base_all + na > out.all_cap);
            if (!fit && tid == 0) atomicOr(&gflags[0], 1u);                  // output buffer too small: host grows it and retries
            if (!fit_all && tid == 0) atomicOr(&gflags[0], 8u);
            if (fit && fit_all) {
                const uint64_t lt_mask = lanemask_lt();
                uint64_t o = base, oa = base_all;
#pragma unroll
                for (int it = 0; it < EMIT_IT; it++) {
                    const uint32_t i = slot0 + it * 64;
                    if (out.all_lo && ((ab[it] >> lane) & 1ull)) {
                        const uint64_t q = oa + (uint32_t)__popcll(ab[it] & lt_mask);
                        if (KW == 2) { out.all_hi[q] = s_key[2 * i + 1]; out.all_lo[q] = s_key[2 * i]; }
                        else out.all_lo[q] = s_key[i];
                    }
                    if ((vb[it] >> lane) & 1ull) {
                        const uint64_t q = o + (uint32_t)__popcll(vb[it] & lt_mask);
                        const uint32_t c = s_cnt[i];
                        const uint32_t c16 = c > 65535u ? 65535u : c;
                        uint32_t pay = IS_SET ? s_aux[i] : ((s_aux[i] & 0xffu) | (c16 << 8));
                        if (WIDE) {                                          // the payload names the record; Exts and colours wait at that position
                            out.w_rec[q] = make_uint4(s_aux[i], s_aux2[i], (s_ex4[i >> 2] >> (8u * (i & 3u))) & 0xffu, 0u);
                            pay = (uint32_t)q;
                        }
                        if (out.rec16) {
                            const uint64_t klo = KW == 2 ? s_key[2 * i] : s_key[i], khi = KW == 2 ? s_key[2 * i + 1] : 0ull;
                            out.rec16[q] = make_uint4((uint32_t)klo, (uint32_t)(klo >> 32), (uint32_t)khi, pay);
                        } else {
                            if (KW == 2) { out.hi[q] = s_key[2 * i + 1]; out.lo[q] = s_key[2 * i]; }
                            else out.lo[q] = s_key[i];
                            out.pay[q] = pay;
                        }
                    }
                    o += (uint32_t)__popcll(vb[it]); oa += (uint32_t)__popcll(ab[it]);
                }
            }
        } else {
            if (P >= 4096u || sp + 2 >= 24) { if (tid == 0) atomicOr(&gflags[0], 2u); break; }
            if (tid == 0) { s_st[sp] = (2 * P) | (pr << 16); s_st[sp + 1] = (2 * P) | ((pr + P) << 16); }
            sp += 2;
        }
        if (sp != 0) __syncthreads();                    // another pass follows: it clears the table and reads the stack
        PH(6);
    }
#ifdef DBG_COUNT_STATS
    __syncthreads();
    if (threadIdx.x >= 4 && threadIdx.x < 16 && s_stat[threadIdx.x]) atomicAdd(&gflags[threadIdx.x], s_stat[threadIdx.x]);
#endif
}

// The kernel.  Resident workgroups (as many as the chip holds at once, fast_count_bins) take bins from a counter, gflags[BIN_TICKET]; the
// number of the next bin is requested while the current one is counted.  One workgroup per bin in launch order (n_bins == 0: bin =
// blockIdx.x, kept for measurements, DBG_FAST_PERSIST=0) pins every eighth bin to one XCD and pays a workgroup launch per bin:
// 52.9 -> 49.2 ms on the default shape (profiles/r06_bin_count_resident.txt).
constexpr int BIN_TICKET = 16;
template <int KW, int NBW, bool IS_SET, int NT, int T, bool WIDE = false, bool WEIGHTED = false>
__global__ void __launch_bounds__(NT, NT == 512 ? 4 : 1) bin_count_kernel(const uint64_t* __restrict__ recs, const uint64_t* __restrict__ recs_alt, uint32_t alt_from,
                                                       const uint64_t* __restrict__ seg_beg, const uint64_t* __restrict__ seg_end,
                                                       uint32_t n_src, uint64_t seg_stride,
                                                       int k, int stranded, uint64_t min_obs, FastOut out, uint64_t out_cap,
                                                       unsigned long long* __restrict__ out_cursor, uint32_t* __restrict__ gflags, uint32_t n_bins) {
    if (n_bins == 0) {
        bin_count_body<KW, NBW, IS_SET, NT, T, WIDE, WEIGHTED>(blockIdx.x, recs, recs_alt, alt_from, seg_beg, seg_end, n_src, seg_stride, k, stranded, min_obs, out, out_cap, out_cursor, gflags);
        return;
    }
    __shared__ uint32_t s_ticket;
    if (threadIdx.x == 0) s_ticket = atomicAdd(&gflags[BIN_TICKET], 1u);
    __syncthreads();
    for (;;) {
        const uint32_t bin = s_ticket;
        __syncthreads();
        if (bin >= n_bins) break;
        // in flight while this bin is counted; the value stays in a register until the bin is done (stored to LDS right away, the wave
        // would wait for the round trip here, in front of the segment bounds' loads)
        uint32_t next_ticket = 0;
        if (threadIdx.x == 0) next_ticket = atomicAdd(&gflags[BIN_TICKET], 1u);
        bin_count_body<KW, NBW, IS_SET, NT, T, WIDE, WEIGHTED>(bin, recs, recs_alt, alt_from, seg_beg, seg_end, n_src, seg_stride, k, stranded, min_obs, out, out_cap, out_cursor, gflags);
        if (threadIdx.x == 0) s_ticket = next_ticket;
        __syncthreads();
    }
}

// CSR helper for the order-restoring stage
// CountFilterSet output as CSR (sorted, deduplicated labels: filter.rs:97-98) straight from the colour masks: block sums of
// the popcounts, a scan of the block sums, then one pass that writes set_off and set_val together.  Entries are dealt to
// lanes round-robin inside a wave (entry = wave base + r*64 + lane), so loads, set_off stores and -- because neighbouring
// lanes own neighbouring output ranges -- set_val stores are all coalesced.
constexpr int CSR_THREADS = 256, CSR_ITEMS = 8, CSR_TILE = CSR_THREADS * CSR_ITEMS;
__device__ __forceinline__ uint32_t popc_mask(uint32_t m) { return (uint32_t)__popc(m); }
__device__ __forceinline__ uint32_t popc_mask(unsigned long long m) { return (uint32_t)__popcll(m); }
__device__ __forceinline__ uint32_t ffs_mask(uint32_t m) { return (uint32_t)__ffs((int)m) - 1u; }
__device__ __forceinline__ uint32_t ffs_mask(unsigned long long m) { return (uint32_t)__ffsll((long long)m) - 1u; }
template <class M>
__global__ void __launch_bounds__(CSR_THREADS) csr_partials_kernel(const M* __restrict__ msk, uint32_t n, uint64_t* __restrict__ partial) {
    __shared__ uint32_t s_w[CSR_THREADS / 64];
    const uint32_t base = blockIdx.x * CSR_TILE + threadIdx.x;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < CSR_ITEMS; i++) { const uint32_t e = base + i * CSR_THREADS; if (e < n) s += popc_mask(msk[e]); }
    for (int d = 32; d; d >>= 1) s += __shfl_xor(s, d);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (uint64_t)s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
template <class M>
__global__ void __launch_bounds__(CSR_THREADS) csr_apply_kernel(const M* __restrict__ msk, uint32_t n, const uint64_t* __restrict__ partial_scanned,
                                                                uint64_t* __restrict__ set_off, uint32_t* __restrict__ set_val, LabelInv inv) {
    // A wave takes 64 entries per round.  Their labels -- up to 64 x 64 -- are first laid out in entry order in the wave's LDS
    // staging area as colour indices (one byte each: every lane walks the bits of its own mask, scattered LDS byte stores), then
    // leave as coalesced stores (consecutive lanes, consecutive labels).  (Storing straight from the bit walk makes every store
    // instruction touch 64 different cache lines: 26 ms for 6.7e9 labels at 24 colours, against 4 ms at 4.)
    constexpr uint32_t STAGE = sizeof(M) * 8 * 64;
    __shared__ uint32_t s_w[CSR_THREADS / 64];
    __shared__ uint32_t s_inv[64];
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[CSR_THREADS / 64][STAGE + 16];
    if (threadIdx.x < 64) s_inv[threadIdx.x] = inv.on ? inv.v[threadIdx.x] : threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t wbase = blockIdx.x * CSR_TILE + wave * (64 * CSR_ITEMS);
    uint8_t* st = s_stage[wave];
    M m[CSR_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int r = 0; r < CSR_ITEMS; r++) { const uint32_t e = wbase + r * 64 + lane; m[r] = e < n ? msk[e] : (M)0; s += popc_mask(m[r]); }
    uint32_t wtot = s;
    for (int d = 32; d; d >>= 1) wtot += __shfl_xor(wtot, d);
    if (lane == 0) s_w[wave] = wtot;
    __syncthreads();
    uint64_t run = partial_scanned[blockIdx.x];
    for (uint32_t w = 0; w < wave; w++) run += s_w[w];
#pragma unroll
    for (int r = 0; r < CSR_ITEMS; r++) {
        const uint32_t e = wbase + r * 64 + lane;
        const uint32_t c = popc_mask(m[r]);
        const uint32_t incl = wave_inclusive_scan_u32(c);            // DPP: no traffic through the LDS crossbar
        const uint32_t P = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (e < n) set_off[e] = run + incl - c;
        uint32_t o = incl - c;
        M x = m[r];
        while (x) { st[o++] = (uint8_t)ffs_mask(x); x &= x - 1; }    // ascending = sort(); dedup() (the map keeps the labels' order)
        // one wave: its LDS operations are performed in order; the barrier + fence keep the compiler from moving the loads
        // below over the (divergent) stores above, and the loads of this round over the stores of the next
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t jb = 0; jb < P; jb += 256) {
#pragma unroll
            for (uint32_t t = 0; t < 4; t++) {                       // every store instruction covers 256 contiguous bytes
                const uint32_t j = jb + t * 64 + lane;
                if (j < P) set_val[run + j] = s_inv[st[j]];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        run += P;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) set_off[n] = partial_scanned[gridDim.x];      // total = the scan's last entry
}
// WIDE colour sets: after the sort the payload columns spell the record's position in the unsorted output (Exts column = low byte,
// mask column = the other 24 bits); Exts and the 64-bit colour mask are fetched from there
__global__ void __launch_bounds__(256) wide_gather_kernel(uint32_t n, uint8_t* __restrict__ exts_io, const uint32_t* __restrict__ q_hi24,
                                                          const uint4* __restrict__ w_rec, unsigned long long* __restrict__ mask_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t q = (uint32_t)exts_io[i] | (q_hi24[i] << 8);
    const uint4 r = w_rec[q];                                       // one line per entry
    exts_io[i] = (uint8_t)r.z;
    mask_out[i] = (unsigned long long)r.x | ((unsigned long long)r.y << 32);
}
__global__ void __launch_bounds__(256) max_label_kernel(const void* data, uint32_t width, uint64_t n, uint32_t* out) {
    // grid-stride maximum of the D1 labels; one atomic per workgroup
    __shared__ uint32_t s_m[4];
    uint32_t v = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (width == 1) {
        const uint64_t n16 = n / 16;
        const uint4* d16 = (const uint4*)data;
        for (uint64_t q = i; q < n16; q += stride) {
            uint4 x = d16[q];
            uint32_t w4[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int t = 0; t < 4; t++) {
                uint32_t y = w4[t];
                uint32_t mx = max(max(y & 0xffu, (y >> 8) & 0xffu), max((y >> 16) & 0xffu, y >> 24));
                v = max(v, mx);
            }
        }
        for (uint64_t q = n16 * 16 + i; q < n; q += stride) v = max(v, (uint32_t)((const uint8_t*)data)[q]);
    } else {
        for (uint64_t q = i; q < n; q += stride) v = max(v, width == 2 ? (uint32_t)((const uint16_t*)data)[q] : ((const uint32_t*)data)[q]);
    }
    for (int d = 32; d; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t m = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3])); if (m) atomicMax(out, m); }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// host orchestration
// ------------------------------------------------------------------------------------------------
// Length of the internal minimizers.  Longer p-mers mean shorter windows (W = k - p + 1) and so more records per read (density
// 2 / (W + 1)), shorter ones mean fewer distinct minimizers than bins: hot bins, slabs that overflow, tables re-streamed in passes.
// Measured at the C2 shape for every p the record layout allows (profiles/r05_p_sweep.json, round 5): 13 bases are the shortest that
// keep the load of 10^6 bins even (4^13 / 2 = 3.4e7 canonical p-mers), so k <= 25 uses 13 whatever the window -- the earlier rule
// p = k - 8 for k <= 20 kept windows of nine p-mers and paid for it with 65 536 .. 8e6 distinct minimizers: 1.0 / 5.0 / 23.9
// Gkmer/s at k = 16 / 18 / 20, now 24.9 / 39.4 / 43.7 -- 26..29 gain 3-8 % with 14, from 30 on 15 wins.
int fast_internal_p(int k) { return k >= 30 ? 15 : (k >= 26 ? 14 : 13); }

struct FastPlan {
    int k, p, nbw, rw;
    bool stranded, is_set, has_hi;
    bool wide = false;                                  // colour sets of 25..64 colours: two mask words per table entry, payload gathered after the sort
    bool weighted = false;                              // sharded flow: records may carry a weight (sender-side duplicate merge) in the WEIGHT_BITS above the meta bits
    bool lists = false;                                 // label lists (fast_labellists.hpp): rw = nbw + 1, the extra record word carries the read's full D1 label
    uint32_t nbins;
    LabelInv linv = {};
    const uint8_t* lmap = nullptr;                      // device table label -> colour index (owned by the caller of fast_labels_prepare)
};

static bool fast_make_plan(dbg_ctx* c, int k, bool stranded, bool is_set, uint64_t total_kmers, uint32_t force_bins, FastPlan* pl) {
    if (k < 16 || k > 64) return false;
    pl->k = k; pl->p = fast_internal_p(k);
    if (const char* e = c->opt("DBG_FAST_P")) pl->p = std::max(4, std::min(std::min(15, k - 3), atoi(e)));     // measurement: internal minimizer length (windows of >= 4 p-mers)
    pl->nbw = std::max(2, (2 * (2 * k - pl->p) + META_BITS + 63) / 64);   // words per record: bases + META_BITS
    if (pl->nbw > 4) return false;
    pl->rw = pl->nbw;
    pl->stranded = stranded; pl->is_set = is_set; pl->has_hi = k > 32;
    pl->weighted = false;
    // k-mer instances per bin.  The table holds T = 2048 distinct k-mers and works best about half full; the share of
    // distinct k-mers among the instances grows with k (every sequencing error spoils k k-mers), so the bins shrink
    // with k: measured optima on 30x reads with e = 0.1 %: k=31 8-12k, k=47 8k, k=51 7-8k, k=63 6k (a bin that
    // overflows is only re-split, at the price of streaming it again).
    uint64_t target = std::min<uint64_t>(10000, std::max<uint64_t>(5000, 380000 / (uint64_t)k));
    if (const char* e = c->opt("DBG_FAST_TARGET")) target = std::max<uint64_t>(256, strtoull(e, nullptr, 10));
    uint64_t nb64 = force_bins ? force_bins : std::max<uint64_t>(1, total_kmers / target);
    // (the workgroup kernel launches one 512-thread workgroup per bin, and a grid holds fewer than 2^32 threads)
    const uint64_t nb_max = (1ull << 23) - 1;
    if (nb64 > nb_max) nb64 = nb_max;
    pl->nbins = (uint32_t)nb64;
    return true;
}

// scan state carried from fast_scan to fast_scatter
struct FastScan {
    FastPlan pl;
    DBuf<uint32_t> hist, tmp_bin;
    DBuf<uint64_t> tmp_recs;
    uint64_t n_tmp = 0, n_recs = 0, n_kmers = 0;
    uint64_t n_merged_away = 0;                         // records the sender-side merge removed (sharded flow)
    // direct mode: per-bin slabs of slab_cap records + fill counts (records beyond the slab are in tmp_*)
    DBuf<uint64_t> slab;
    DBuf<uint32_t> cursor;
    uint32_t slab_cap = 0;
};

// which D1 labels (< 65536) occur: one bit each; bitmap[2048] != 0 when a label >= 65536 was seen
__global__ void __launch_bounds__(256) label_presence_kernel(const void* data, uint32_t width, uint64_t n, uint32_t* __restrict__ bitmap) {
    __shared__ uint32_t s_b[2049];
    for (int i = threadIdx.x; i < 2049; i += 256) s_b[i] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t v = width == 1 ? ((const uint8_t*)data)[i] : (width == 2 ? ((const uint16_t*)data)[i] : ((const uint32_t*)data)[i]);
        if (v < 65536u) { const uint32_t bit = 1u << (v & 31u); if (!(s_b[v >> 5] & bit)) atomicOr(&s_b[v >> 5], bit); }
        else s_b[2048] = 1u;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2049; i += 256) if (s_b[i]) atomicOr(&bitmap[i], s_b[i]);
}

// The LDS colour bitmask holds 24 colours next to the Exts, or 64 in the WIDE layout of the counting kernel: labels below that are
// their own colour; a sparse alphabet of at most 64 distinct labels (all < 65536) is mapped to colour indices in increasing label
// order and mapped back when the label sets are written.
// lmap_buf receives the device table (it must outlive the scan).
static int fast_labels_prepare(dbg_ctx* c, const SeqDev& s, FastPlan* pl, DBuf<uint8_t>* lmap_buf, bool* ok, std::vector<uint32_t>* many = nullptr) {
    *ok = false;
    pl->linv.on = 0; pl->lmap = nullptr; pl->wide = false;
    if (!s.data) return 0;
    DBuf<uint32_t> mx;
    ALLOC_OR_FAIL(c, mx, 1);
    HIP_TRY(c, hipMemsetAsync(mx.p, 0, 4, c->stream));
    if (s.n) { max_label_kernel<<<(uint32_t)std::min<uint64_t>(cdiv(s.n, 256), 2048), 256, 0, c->stream>>>(s.data, s.data_width, s.n, mx.p); LAUNCH_CHECK(c, "max_label"); }
    uint32_t h = 0;
    HIP_TRY(c, hipMemcpyAsync(&h, mx.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (h < 24) { *ok = true; return 0; }
    if (h < 64) { pl->wide = true; *ok = true; return 0; }
    if (h >= 65536u) return 0;
    DBuf<uint32_t> bm;
    ALLOC_OR_FAIL(c, bm, 2049);
    HIP_TRY(c, hipMemsetAsync(bm.p, 0, 2049 * 4, c->stream));
    label_presence_kernel<<<(uint32_t)std::min<uint64_t>(cdiv(s.n, 256), 1024), 256, 0, c->stream>>>(s.data, s.data_width, s.n, bm.p);
    LAUNCH_CHECK(c, "label_presence");
    std::vector<uint32_t> hb(2049);
    HIP_TRY(c, hipMemcpyAsync(hb.data(), bm.p, 2049 * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (hb[2048]) return 0;
    std::vector<uint8_t> map((size_t)h + 1, 0);
    uint32_t nd = 0;
    for (uint32_t v = 0; v <= h; v++) {
        if (hb[v >> 5] & (1u << (v & 31))) {
            if (nd == 64) {                                 // more than 64 distinct labels: label-group passes (or the generic path)
                if (many) { many->clear(); for (uint32_t u = 0; u <= h; u++) if (hb[u >> 5] & (1u << (u & 31))) many->push_back(u); }
                return 0;
            }
            pl->linv.v[nd] = v;
            map[v] = (uint8_t)nd++;
        }
    }
    ALLOC_OR_FAIL(c, *lmap_buf, map.size());
    HIP_TRY(c, hipMemcpyAsync(lmap_buf->p, map.data(), map.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));            // `map` leaves scope
    pl->linv.on = 1; pl->lmap = lmap_buf->p;
    pl->wide = nd > 24;
    *ok = true;
    return 0;
}
// which D1 labels occur: hb[v >> 5] bit (v & 31) for labels < 65536, hb[2048] != 0 when a larger one was seen
static int seq_label_bitmap(dbg_ctx* c, const SeqDev& s, std::vector<uint32_t>* hb) {
    hb->assign(2049, 0u);
    if (!s.data || !s.n) return 0;
    DBuf<uint32_t> bm;
    ALLOC_OR_FAIL(c, bm, 2049);
    HIP_TRY(c, hipMemsetAsync(bm.p, 0, 2049 * 4, c->stream));
    label_presence_kernel<<<(uint32_t)std::min<uint64_t>(cdiv(s.n, 256), 1024), 256, 0, c->stream>>>(s.data, s.data_width, s.n, bm.p);
    LAUNCH_CHECK(c, "label_presence");
    HIP_TRY(c, hipMemcpyAsync(hb->data(), bm.p, 2049 * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}
int seq_max_label(dbg_ctx* c, const SeqDev& s, uint32_t* out) {
    *out = 0;
    if (!s.data || !s.n) return 0;
    DBuf<uint32_t> mx;
    ALLOC_OR_FAIL(c, mx, 1);
    HIP_TRY(c, hipMemsetAsync(mx.p, 0, 4, c->stream));
    max_label_kernel<<<(uint32_t)std::min<uint64_t>(cdiv(s.n, 256), 2048), 256, 0, c->stream>>>(s.data, s.data_width, s.n, mx.p);
    LAUNCH_CHECK(c, "max_label");
    HIP_TRY(c, hipMemcpyAsync(out, mx.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

// The slabs are ~26 GB written 24 bytes at a time at random addresses, and how fast that goes depends on where the driver put the
// block: the same 1.3e8 random writes take 5.4 .. 7.4 ms, the scan 30.8 .. 35.6 ms, steady to 0.3 ms per block and different in every
// process (profiles/r06_slab_placement.txt).  Rounds 3-4 probed up to four candidate blocks with a synthetic kernel; round 5 mapped the
// slab piecewise from 256 MB physical handles and believed the matter closed -- round 6 measured that no handle size (2 MB .. 2 GB, any
// alignment) changes the lottery (every single 256 MB handle is equally fast; it is the COMBINATION of physical ranges a block is made
// of that decides), and that releasing mapped handles corrupts later mappings on this driver (dbg_ctx::dalloc_fresh).  So the slab is a
// plain block again, and the lottery is played on purpose: the first DBG_SLAB_TRIALS (default 4) calls of a shape each scan into a
// FRESH block while the earlier candidates wait in the pool; the scan itself is the probe (timed with events), and from then on only
// the fastest block stays (the others go back with hipFree).  Six processes: scan 30.9 .. 31.2 ms after the tournament against
// 30.8 .. 34.2 without.  A trial needs room for one more slab (checked); a change of shape starts over; DBG_SLAB_TRIALS=1: no trials.
static bool slab_alloc(dbg_ctx* c, DBuf<uint64_t>* slab, size_t words, uint32_t rec_words, int k, bool allow_trial, bool* trial) {
    const size_t bytes = words * 8;
    *trial = false;
    // what the scan runs on is recorded in the ctx (dbg_ctx_get_stats)
    c->slab_ptr = nullptr; c->slab_bytes = bytes; c->slab_rec_words = rec_words; c->slab_note[0] = 0;
    c->slab_backing = bytes < (4ull << 30) ? DBG_SLAB_PLAIN_SMALL : DBG_SLAB_PLAIN;
    uint32_t n_trials = 4;
    if (const char* e = c->opt("DBG_SLAB_TRIALS")) n_trials = (uint32_t)std::max(1, std::min(8, atoi(e)));
    if (bytes >= (4ull << 30) && n_trials > 1) {
        (void)hipStreamSynchronize(c->stream);                           // (a pooled block may still be in use by queued work of its previous owner)
        const uint64_t key = (uint64_t)bytes ^ ((uint64_t)k << 48) ^ ((uint64_t)rec_words << 56);
        if (key != c->slab_trial_key) {
            c->slab_trial_key = key; c->slab_trials_done = 0; c->slab_cands.clear();
            for (float& x : c->slab_trial_ms) x = 0.f;
        }
        // settle what has been measured so far: only the fastest candidate stays pooled
        if (c->slab_cands.size() > 1) {
            auto cands = c->slab_cands;
            size_t best = 0;
            for (size_t i = 1; i < cands.size(); i++) if (cands[i].ns_per_kmer < cands[best].ns_per_kmer) best = i;
            for (size_t i = 0; i < cands.size(); i++) {
                if (i == best) continue;
                for (auto it = c->free_blocks.begin(); it != c->free_blocks.end(); ++it)
                    if (it->second == cands[i].p) { c->pooled_bytes -= it->first; c->free_blocks.erase(it); c->raw_free(cands[i].p); break; }
            }
        }
        bool fresh = allow_trial && c->slab_trials_done >= 1 && c->slab_trials_done < n_trials;
        if (fresh) {                                                      // room for one more slab next to everything else?
            size_t fr = 0, tot = 0;
            if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < bytes + bytes / 2) { (void)hipGetLastError(); fresh = false; c->slab_trials_done = n_trials; }
        }
        if (fresh) {
            if (void* p = c->dalloc_fresh(bytes)) {
                slab->release();
                slab->ctx = c; slab->p = (uint64_t*)p; slab->n = words;
                c->n_fresh_blocks++;
                c->slab_ptr = p;
                *trial = true;
                return true;
            }
            c->slab_trials_done = n_trials;                              // the device is full: the tournament ends with what it has
        }
        const bool ok = slab->alloc(c, words);
        if (ok) { c->slab_ptr = slab->p; *trial = allow_trial && c->slab_trials_done < n_trials; }
        return ok;
    }
    const bool ok = slab->alloc(c, words);
    if (ok) c->slab_ptr = slab->p;
    return ok;
}

static bool lane_scan_wanted(dbg_ctx* c) { return !(c->opt("DBG_SCAN") && !strcmp(c->opt("DBG_SCAN"), "wave")); }

// scan: super-k-mer records in read order + bin histogram
static int fast_scan(dbg_ctx* c, const SeqDev& s, const FastPlan& pl, uint64_t n_kmers, FastScan* st, bool direct = false) {
    st->pl = pl; st->n_kmers = n_kmers;
    const int k = pl.k, p = pl.p, nbw = pl.nbw, rw = pl.rw;
    const uint32_t nbins = pl.nbins * NCLS;                      // sub-bins (bin, length class)
    FastCfg cfg{k, p, pl.stranded, pl.nbins, pl.lmap, (!pl.stranded && (k & 1) && !c->opt("DBG_NO_STRAND_NORM")) ? 1 : 0};
    SeqDev sd = s;
    if (!pl.is_set) { sd.data = nullptr; sd.data_width = 0; }   // CountFilter ignores D1 (filter.rs:52-62)
    DBuf<uint32_t> sflags;
    DBuf<unsigned long long> tmp_cursor;
    bool slab_trial = false;                                     // this call's scan decides whether its (fresh) slab block stays: slab_alloc
    ALLOC_OR_FAIL(c, st->hist, nbins);
    ALLOC_OR_FAIL(c, tmp_cursor, 1);
    ALLOC_OR_FAIL(c, sflags, 2);
    // expected density of minimizer changes is 2/(W+1) per window plus one piece per read
    uint64_t tmp_cap = (uint64_t)((double)n_kmers * 2.0 / (double)(k - p + 2) * 1.15) + s.n + 1024;
    if (tmp_cap > n_kmers) tmp_cap = n_kmers;
    if (direct) {
        // bins are hash-distributed, so a slab of 1.3 x the mean bin size (+ slack for small bins) holds almost every
        // bin; what does not fit (heavy minimizers of low-complexity sequence) goes through the read-order buffer
        const double mean = (double)tmp_cap / (double)nbins;
        st->slab_cap = ((uint32_t)std::min<double>(mean * 1.3 + 48.0, 4.0e9) + 3u) & ~3u;
        if (const char* e = c->opt("DBG_SLAB_CAP")) st->slab_cap = (uint32_t)std::max(4, atoi(e)) & ~3u;      // measurement: records per slab
        ALLOC_OR_FAIL(c, st->cursor, nbins);
        const bool gated_upload = c->read_gates && c->read_gates->upto.size() > 1;     // (the scan's interval would include the waits for the upload)
        if (!c->opt("DBG_FAST_NO_SLAB") && slab_alloc(c, &st->slab, (uint64_t)nbins * st->slab_cap * rw, (uint32_t)rw, k, !gated_upload, &slab_trial)) tmp_cap = tmp_cap / 16 + 4096;
        else {
            if (!c->opt("DBG_FAST_NO_SLAB")) fprintf(stderr, "[dbg_mi355x] no device memory for per-bin slabs: records take the read-order buffer and the scatter pass\n");
            c->slab_backing = DBG_SLAB_RECORD_ORDER; c->slab_ptr = nullptr;
            snprintf(c->slab_note, sizeof(c->slab_note), "%s", c->opt("DBG_FAST_NO_SLAB") ? "DBG_FAST_NO_SLAB" : "slab allocation failed");
            // not enough memory for slabs (1.3x the records + slack): every record takes the read-order buffer and the
            // scatter pass instead -- slab capacity 0 routes them all there
            st->slab_cap = 0;
            ALLOC_OR_FAIL(c, st->slab, 4);
        }
    }
    const uint32_t scan_blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((s.n + 255) / 256, 256ull * 32));
    // every wave may strand one partial chunk -- of every launch: a gated host-boundary scan launches once per upload chunk
    const bool gated_scan = lane_scan_wanted(c) && direct && c->read_gates && c->read_gates->upto.size() > 1;
    const uint64_t chunk_slack = (uint64_t)scan_blocks * 4 * SCAN_CHUNK * (gated_scan ? c->read_gates->upto.size() : 1);
    tmp_cap += chunk_slack;
    for (int attempt = 0;; attempt++) {
        ALLOC_OR_FAIL(c, st->tmp_recs, tmp_cap * rw);
        ALLOC_OR_FAIL(c, st->tmp_bin, tmp_cap);
        HIP_TRY(c, hipMemsetAsync(st->hist.p, 0, (size_t)nbins * 4, c->stream));
        HIP_TRY(c, hipMemsetAsync(tmp_cursor.p, 0, 8, c->stream));
        HIP_TRY(c, hipMemsetAsync(sflags.p, 0, 8, c->stream));
        HIP_TRY(c, hipMemsetAsync(st->tmp_bin.p, 0xff, tmp_cap * 4, c->stream));
        if (direct) HIP_TRY(c, hipMemsetAsync(st->cursor.p, 0, (size_t)nbins * 4, c->stream));
        // reads of up to SCAN_LANE_MAX bases: lane-per-read kernel; it flags longer reads, which the wave-per-read kernel then
        // takes (DBG_SCAN=wave: everything through the wave-per-read kernel)
        const bool lane_scan = lane_scan_wanted(c);
        unsigned long long cur = 0;
        uint32_t sfl[2] = {0, 0};
        SeqDev cur_reads = sd;                                   // (the reads of one launch: all of them, or one chunk of a gated upload)
#define SCAN_ARGS cur_reads, cfg, st->hist.p, st->tmp_recs.p, st->tmp_bin.p, tmp_cursor.p, tmp_cap, sflags.p, st->slab.p, st->slab_cap, st->cursor.p
        // Host-boundary calls upload the packed words in chunks of reads next to this scan (dbg_ctx::read_gates, api.hip): the first
        // attempt of the lane scan launches once per chunk as the chunks arrive; everything else needs all reads
        dbg_read_gates* gates = (lane_scan_wanted(c) && direct && attempt == 0 && c->read_gates && c->read_gates->upto.size() > 1) ? c->read_gates : nullptr;
        if (!gates) DBG_TRY(c->wait_all_reads());
        if (lane_scan) {
            const uint32_t W = (uint32_t)(k - p + 1);
#ifndef DBG_SCAN_LDS_PAD
#define DBG_SCAN_LDS_PAD 0
#endif
            const size_t lds = (size_t)scan_lane_lds_words(W) * sizeof(uint32_t) + DBG_SCAN_LDS_PAD;
            uint32_t lane_blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((s.n + 63) / 64, 256ull * 16 * 4));
            if (!gates) c->t_begin("sk_scan", n_kmers);           // (gated: one interval per launch, without the waits for the upload)
            hipEvent_t tr_a = nullptr, tr_b = nullptr;
            if (slab_trial && attempt == 0) { tr_a = c->get_event(); tr_b = c->get_event(); (void)hipEventRecord(tr_a, c->stream); }
#define SCANL(NBW_, D_, L_) do { HIP_TRY(c, hipFuncSetAttribute((const void*)sk_scan_lane_kernel<NBW_, D_, L_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            sk_scan_lane_kernel<NBW_, D_, L_><<<lane_blocks, 64, lds, c->stream>>>(SCAN_ARGS); } while (0)
#define SCANL_D(NBW_, L_) do { if (direct) SCANL(NBW_, true, L_); else SCANL(NBW_, false, L_); } while (0)
#define SCANL_GO() do { if (pl.lists) { if (nbw == 2) SCANL_D(2, true); else if (nbw == 3) SCANL_D(3, true); else SCANL_D(4, true); } \
            else { if (nbw == 2) SCANL_D(2, false); else if (nbw == 3) SCANL_D(3, false); else SCANL_D(4, false); } } while (0)
            if (!gates) SCANL_GO();
            else {
                uint64_t r0 = 0;
                for (size_t g = 0; g < gates->upto.size(); g++) {
                    if (const int e = gates->wait(g)) return c->fail(e, gates->msg);
                    const uint64_t r1 = std::min<uint64_t>(gates->upto[g], s.n);
                    if (r1 <= r0) continue;
                    cur_reads = sd;
                    cur_reads.start += r0; cur_reads.length += r0; cur_reads.n = r1 - r0;
                    if (cur_reads.exts) cur_reads.exts += r0;
                    if (cur_reads.data) cur_reads.data = (const uint8_t*)cur_reads.data + r0 * cur_reads.data_width;
                    lane_blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((cur_reads.n + 63) / 64, 256ull * 16 * 4));
                    c->t_begin("sk_scan", (uint64_t)((double)n_kmers * (double)(r1 - r0) / (double)std::max<uint64_t>(s.n, 1)));
                    SCANL_GO();
                    c->t_end();
                    LAUNCH_CHECK(c, "sk_scan_lane");
                    r0 = r1;
                }
                cur_reads = sd;
            }
#undef SCANL_GO
#undef SCANL_D
#undef SCANL
            if (!gates) c->t_end();
            if (tr_b) (void)hipEventRecord(tr_b, c->stream);
            LAUNCH_CHECK(c, "sk_scan_lane");
            HIP_TRY(c, hipMemcpyAsync(sfl, sflags.p, 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipMemcpyAsync(&cur, tmp_cursor.p, 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (tr_b) {                                          // one more candidate of the slab tournament measured
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, tr_a, tr_b) == hipSuccess && ms > 0.f && c->slab_ptr == (void*)st->slab.p) {
                    if (c->slab_trials_done < 8) c->slab_trial_ms[c->slab_trials_done] = ms;
                    c->slab_trials_done++;
                    bool known = false;
                    for (auto& cd : c->slab_cands) if (cd.p == c->slab_ptr) { cd.ns_per_kmer = std::min(cd.ns_per_kmer, (double)ms * 1e6 / (double)std::max<uint64_t>(n_kmers, 1)); known = true; }
                    if (!known) c->slab_cands.push_back({c->slab_ptr, c->slab_bytes, (double)ms * 1e6 / (double)std::max<uint64_t>(n_kmers, 1)});
                } else (void)hipGetLastError();
                c->event_pool.push_back(tr_a); c->event_pool.push_back(tr_b);
            }
        }
        if (!lane_scan || sfl[1]) {
            const uint32_t long_min = lane_scan ? SCAN_LANE_MAX : 0u;
            c->t_begin(lane_scan ? "sk_scan_long" : "sk_scan", lane_scan ? 0 : n_kmers);
#define SCAN(NBW_, D_, L_) sk_scan_kernel<NBW_, D_, L_><<<scan_blocks, 256, 0, c->stream>>>(SCAN_ARGS, long_min)
#define SCAN_D(NBW_, L_) do { if (direct) SCAN(NBW_, true, L_); else SCAN(NBW_, false, L_); } while (0)
            if (pl.lists) { if (nbw == 2) SCAN_D(2, true); else if (nbw == 3) SCAN_D(3, true); else SCAN_D(4, true); }
            else { if (nbw == 2) SCAN_D(2, false); else if (nbw == 3) SCAN_D(3, false); else SCAN_D(4, false); }
#undef SCAN_D
#undef SCAN
            c->t_end();
            LAUNCH_CHECK(c, "sk_scan");
            HIP_TRY(c, hipMemcpyAsync(&cur, tmp_cursor.p, 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
#undef SCAN_ARGS
        if (c->opt("DBG_DEBUG")) fprintf(stderr, "[fastpath] scan done: nbw=%d rw=%d bins=%u slab_cap=%u tmp used %llu of %llu slab=%p cursor=%p words=%p\n", nbw, rw, nbins, st->slab_cap,
                                         cur, (unsigned long long)tmp_cap, (void*)st->slab.p, (void*)st->cursor.p, (const void*)s.words);
        if (cur > tmp_cap) {                                  // low-complexity input: more pieces than estimated
            if (attempt >= 2) return c->fail(133, "fast path: super-k-mer buffer estimate failed");
            tmp_cap = std::min<uint64_t>(n_kmers, cur) + chunk_slack;
            continue;
        }
        st->n_tmp = cur;
        break;
    }
    return 0;
}

// scatter into bin order: recs_out [n_recs * rw] and bin_off_out [nbins + 1] are caller-provided device buffers.
// n_recs must first be obtained with fast_bin_offsets.
static int fast_bin_offsets(dbg_ctx* c, FastScan* st, uint64_t* bin_off_out, bool book = true) {
    DBG_TRY(scan_exclusive_u32_u64(c, st->hist.p, bin_off_out, (uint64_t)st->pl.nbins * NCLS));
    HIP_TRY(c, hipMemcpyAsync(&st->n_recs, bin_off_out + (uint64_t)st->pl.nbins * NCLS, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (book) {
        c->t_begin("sk_records", st->n_recs);    // bookkeeping entry: units = super-k-mer records (no kernel)
        c->t_end();
    }
    return 0;
}
static int fast_scatter(dbg_ctx* c, FastScan* st, const uint64_t* bin_off, uint64_t* recs_out) {
    const uint32_t nbins = st->pl.nbins * NCLS;
    DBuf<uint32_t> cursor;
    ALLOC_OR_FAIL(c, cursor, nbins);
    HIP_TRY(c, hipMemsetAsync(cursor.p, 0, (size_t)nbins * 4, c->stream));
    if (st->n_tmp) {
        const uint64_t n_tmp = st->n_tmp;
        c->t_begin("sk_scatter", st->n_recs);
        if (st->pl.rw == 2) sk_scatter_kernel<2><<<cdiv(n_tmp, 256), 256, 0, c->stream>>>(st->tmp_recs.p, st->tmp_bin.p, n_tmp, bin_off, cursor.p, recs_out);
        else if (st->pl.rw == 3) sk_scatter_kernel<3><<<cdiv(n_tmp, 256), 256, 0, c->stream>>>(st->tmp_recs.p, st->tmp_bin.p, n_tmp, bin_off, cursor.p, recs_out);
        else if (st->pl.rw == 4) sk_scatter_kernel<4><<<cdiv(n_tmp, 256), 256, 0, c->stream>>>(st->tmp_recs.p, st->tmp_bin.p, n_tmp, bin_off, cursor.p, recs_out);
        else sk_scatter_kernel<5><<<cdiv(n_tmp, 256), 256, 0, c->stream>>>(st->tmp_recs.p, st->tmp_bin.p, n_tmp, bin_off, cursor.p, recs_out);
        c->t_end();
        LAUNCH_CHECK(c, "sk_scatter");
    }
    st->tmp_recs.release(); st->tmp_bin.release(); st->hist.release();
    return 0;
}

// Workgroups of `kern` the device holds at once (the grid of a kernel whose workgroups take their work from a counter).
static uint32_t resident_grid(dbg_ctx* c, const void* kern, int nt, size_t dyn_lds, const char* forced) {
    if (forced && atoi(forced) > 0) return (uint32_t)atoi(forced);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, nt, dyn_lds) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 2; }
    return (uint32_t)per_cu * (uint32_t)std::max(1, c->n_cu);
}
// per-bin LDS hash tables over `nbins_local` bins whose records arrive as n_src segments, then the
// order-restoring sort and the output table
// Counting state that outlives one bin_count launch: the unsorted (key, payload) records of the valid k-mers found so
// far.  The multi-GPU flow counts its bins in several chunks (while the next chunk is still on the wire) and sorts once.
struct FastCountState {
    FastPlan pl;
    uint64_t min_obs = 0, cap = 0, n_out = 0, n_kmers_hint = 0;
    DBuf<uint64_t> u_hi, u_lo;
    DBuf<uint32_t> u_pay, gflags;
    DBuf<uint4> u16;                       // k <= 48: 16-byte records instead of the three arrays
    bool use16 = false;
    DBuf<uint4> w_rec;                     // WIDE colour sets: 64-bit colour mask + Exts by output position
    DBuf<unsigned long long> out_cursor;
    // report_all_kmers: every distinct key
    DBuf<unsigned long long>* keep_masks = nullptr;   // WIDE colour sets: hand the sorted 64-bit masks to the caller instead of building the CSR
    bool report_all = false;
    uint64_t all_cap = 0, n_all = 0;
    DBuf<uint64_t> a_hi, a_lo;
    DBuf<unsigned long long> all_cursor;
};

static int fast_count_alloc_all(dbg_ctx* c, FastCountState* st, uint64_t cap) {
    if (cap >= (1ull << 32)) cap = (1ull << 32) - 1;
    DBuf<uint64_t> n_hi, n_lo;
    if (st->pl.has_hi) ALLOC_OR_FAIL(c, n_hi, cap);
    ALLOC_OR_FAIL(c, n_lo, cap);
    if (st->n_all) {
        if (st->pl.has_hi) HIP_TRY(c, hipMemcpyAsync(n_hi.p, st->a_hi.p, st->n_all * 8, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(n_lo.p, st->a_lo.p, st->n_all * 8, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    std::swap(st->a_hi, n_hi); std::swap(st->a_lo, n_lo);
    st->all_cap = cap;
    return 0;
}

static int fast_count_alloc(dbg_ctx* c, FastCountState* st, uint64_t cap) {
    if (cap >= (1ull << 32)) cap = (1ull << 32) - 1;
    if (st->pl.wide) {
        DBuf<uint4> nr;
        ALLOC_OR_FAIL(c, nr, cap);
        if (st->n_out) {
            HIP_TRY(c, hipMemcpyAsync(nr.p, st->w_rec.p, st->n_out * 16, hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
        std::swap(st->w_rec, nr);
    }
    if (st->use16) {
        DBuf<uint4> n16;
        ALLOC_OR_FAIL(c, n16, cap);
        if (st->n_out) {
            HIP_TRY(c, hipMemcpyAsync(n16.p, st->u16.p, st->n_out * 16, hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
        std::swap(st->u16, n16);
        st->cap = cap;
        return 0;
    }
    DBuf<uint64_t> n_hi, n_lo;
    DBuf<uint32_t> n_pay;
    if (st->pl.has_hi) ALLOC_OR_FAIL(c, n_hi, cap);
    ALLOC_OR_FAIL(c, n_lo, cap);
    ALLOC_OR_FAIL(c, n_pay, cap);
    if (st->n_out) {                                             // keep what earlier chunks produced
        if (st->pl.has_hi) HIP_TRY(c, hipMemcpyAsync(n_hi.p, st->u_hi.p, st->n_out * 8, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(n_lo.p, st->u_lo.p, st->n_out * 8, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(n_pay.p, st->u_pay.p, st->n_out * 4, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    std::swap(st->u_hi, n_hi); std::swap(st->u_lo, n_lo); std::swap(st->u_pay, n_pay);
    st->cap = cap;
    return 0;
}

static int fast_count_begin(dbg_ctx* c, const FastPlan& pl, uint64_t min_obs, uint64_t n_kmers_hint, FastCountState* st,
                            bool report_all = false) {
    st->pl = pl; st->min_obs = min_obs; st->n_out = 0; st->n_kmers_hint = n_kmers_hint;
    st->report_all = report_all; st->n_all = 0;
    st->use16 = 2 * pl.k <= 96 && !c->opt("DBG_NO_REC16");
    ALLOC_OR_FAIL(c, st->out_cursor, 1);
    if (report_all) {
        ALLOC_OR_FAIL(c, st->all_cursor, 1);
        DBG_TRY(fast_count_alloc_all(c, st, std::max<uint64_t>(std::min<uint64_t>(n_kmers_hint, std::max<uint64_t>(n_kmers_hint / 4, 1u << 20)), 1)));
    }
    ALLOC_OR_FAIL(c, st->gflags, 32);
    HIP_TRY(c, hipMemsetAsync(st->out_cursor.p, 0, 8, c->stream));
    return fast_count_alloc(c, st, std::max<uint64_t>(std::min<uint64_t>(n_kmers_hint, std::max<uint64_t>(n_kmers_hint / 8, 1u << 20)), 1));
}

// per-bin LDS hash tables over `nbins_local` bins whose records arrive as n_src segments; valid k-mers are appended to the state
static int fast_count_bins(dbg_ctx* c, FastCountState* st, const uint64_t* recs, const uint64_t* recs_alt, uint32_t alt_from,
                           const uint64_t* seg_beg, const uint64_t* seg_end, uint32_t n_src, uint64_t seg_stride,
                           uint32_t nbins_local, uint64_t n_kmers_units, uint64_t n_recs_hint) {
    const FastPlan& pl = st->pl;
    const int k = pl.k, nbw = pl.nbw;
    const bool is_set = pl.is_set, has_hi = pl.has_hi;
    const uint64_t min_obs = st->min_obs;
    constexpr uint32_t TABLE = 2048;
    for (int attempt = 0;; attempt++) {
        const uint64_t cap = st->cap;
        unsigned long long start = st->n_out, start_all = st->n_all;
        HIP_TRY(c, hipMemcpyAsync(st->out_cursor.p, &start, 8, hipMemcpyHostToDevice, c->stream));
        if (st->report_all) HIP_TRY(c, hipMemcpyAsync(st->all_cursor.p, &start_all, 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemsetAsync(st->gflags.p, 0, 128, c->stream));
        FastOut fo{st->u_hi.p, st->u_lo.p, st->u_pay.p, st->use16 ? st->u16.p : nullptr, st->w_rec.p,
                   st->report_all ? st->a_hi.p : nullptr, st->report_all ? st->a_lo.p : nullptr,
                   st->report_all ? st->all_cursor.p : nullptr, st->all_cap};
        unsigned long long* out_cursor_p = st->out_cursor.p;
        uint32_t* gflags_p = st->gflags.p;
        if (nbins_local) {
            if ((uint64_t)nbins_local * 512 >= (1ull << 32)) return c->fail(135, "fast path: too many bins for the workgroup counting kernel (a grid holds < 2^32 threads)");
            c->t_begin("bin_count", n_kmers_units);
            const int nt_env = c->opt("DBG_FAST_NT") ? atoi(c->opt("DBG_FAST_NT")) : 512;
            const int tb_env = c->opt("DBG_FAST_TABLE") ? atoi(c->opt("DBG_FAST_TABLE")) : 2048;
            const size_t dyn_lds = c->opt("DBG_DYN_LDS") ? (size_t)atoi(c->opt("DBG_DYN_LDS")) : 0;   // measurement: extra LDS per workgroup (8192 leaves room for only one per CU)
            // resident workgroups: as many as the device holds at once (DBG_FAST_PERSIST=<n>: that many; 0: one workgroup per bin)
            const bool resident = !(c->opt("DBG_FAST_PERSIST") && atoi(c->opt("DBG_FAST_PERSIST")) == 0);
            uint32_t grid_wg = nbins_local;
            const uint32_t n_bins_persist = resident ? nbins_local : 0u;
#define ARGS_ recs, recs_alt, alt_from, seg_beg, seg_end, n_src, seg_stride, k, pl.stranded ? 1 : 0, min_obs, fo, cap, out_cursor_p, gflags_p, n_bins_persist
#define LK(NTT, ...) do { auto kern = bin_count_kernel<__VA_ARGS__>; \
            if (resident) grid_wg = std::min(nbins_local, resident_grid(c, (const void*)kern, NTT, dyn_lds, c->opt("DBG_FAST_PERSIST"))); \
            kern<<<grid_wg, NTT, dyn_lds, c->stream>>>(ARGS_); } while (0)
#define L(KW, NBW, SET, NTT, TT) do { if (pl.weighted && NTT == 512) LK(512, KW, NBW, SET, 512, TT, false, true); \
            else LK(NTT, KW, NBW, SET, NTT, TT); } while (0)
#define LW(KW, NBW) do { if (pl.weighted) LK(512, KW, NBW, true, 512, 1024, true, true); \
            else LK(512, KW, NBW, true, 512, 1024, true); } while (0)
#define GO(KW, NBW, SET) do { \
            if (SET && pl.wide) LW(KW, NBW); \
            else if (tb_env == 1024 && nt_env == 256 && !pl.weighted) L(KW, NBW, SET, 256, 1024); \
            else L(KW, NBW, SET, 512, TABLE); } while (0)
            if (!has_hi) { if (is_set) GO(1, 2, true); else GO(1, 2, false); }
            else if (nbw == 2) { if (is_set) GO(2, 2, true); else GO(2, 2, false); }
            else if (nbw == 3) { if (is_set) GO(2, 3, true); else GO(2, 3, false); }
            else { if (is_set) GO(2, 4, true); else GO(2, 4, false); }
#undef GO
#undef LW
#undef L
#undef LK
#undef ARGS_
            c->t_end();
            LAUNCH_CHECK(c, "bin_count");
        }
        unsigned long long cur = 0, cur_all = 0;
        uint32_t flv[16] = {0};
        HIP_TRY(c, hipMemcpyAsync(&cur, st->out_cursor.p, 8, hipMemcpyDeviceToHost, c->stream));
        if (st->report_all) HIP_TRY(c, hipMemcpyAsync(&cur_all, st->all_cursor.p, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(flv, st->gflags.p, 64, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        const uint32_t fl = flv[0];
        if (c->opt("DBG_DEBUG")) fprintf(stderr, "[fastpath] bins=%u srcs=%u recs=%llu valid=%llu flags=%u maxP=%u split_passes=%u filter_undone=%u wd=%u\n",
                                         nbins_local, n_src, (unsigned long long)n_recs_hint, cur, fl, flv[1], flv[2], flv[12], flv[3]);
#ifdef DBG_PHASE_TIMES
        if (c->opt("DBG_DEBUG")) {
            unsigned long long ph[8];
            (void)hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_phase_cycles), sizeof(ph));
            fprintf(stderr, "[fastpath-phases] (100MHz ticks, summed over bins) prologue=%llu clear=%llu stage+filter+map=%llu chunks(w0)=%llu wait=%llu scan=%llu cursor=%llu write=%llu\n",
                    ph[0], ph[1], ph[7], ph[2], ph[3], ph[4], ph[5], ph[6]);
            unsigned long long z[8] = {0};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof(z));
        }
#endif
#ifdef DBG_COUNT_STATS
        if (c->opt("DBG_DEBUG")) fprintf(stderr, "[fastpath-stats] wave-iterations=%u active lanes=%u (%.1f/64); lane-rounds=%u next-bucket=%u tag-collision=%u busy=%u cas-lost=%u inserts=%u; staged after round 1=%u, round-2 incoming=%u, merged in round 1=%u, later=%u\n",
                                         flv[4], flv[5], flv[4] ? (double)flv[5] / flv[4] : 0.0, flv[6], flv[7], flv[8], flv[9], flv[10], flv[11], flv[13], flv[12], flv[15], flv[14]);
#endif
        if (flv[3] & 4u) return c->fail(134, "fast path: corrupt super-k-mer record (record buffer or segment table of the counting stage is wrong)");
        if (flv[3]) return c->fail(132, "fast path: internal watchdog fired");
        if (fl & 6u) return c->fail(130, "fast path: a bin exceeded the multi-pass limit");
        if (fl & 9u) {                                            // an output buffer is too small: grow (earlier chunks are kept) and redo this launch
            if (attempt >= 3 || cur >= (1ull << 32) || cur_all >= (1ull << 32)) return c->fail(131, "fast path: more than 2^32-1 k-mers in one table");
            if (fl & 1u) DBG_TRY(fast_count_alloc(c, st, cur + cur / 16 + 1024));
            if (fl & 8u) DBG_TRY(fast_count_alloc_all(c, st, cur_all + cur_all / 16 + 1024));
            continue;
        }
        st->n_out = cur;
        st->n_all = cur_all;
        break;
    }
    return 0;
}

// order-restoring sort of everything counted so far -> the output table
static int fast_count_finish(dbg_ctx* c, FastCountState* st, dbg_kmer_table* out) {
    const FastPlan& pl = st->pl;
    const int k = pl.k;
    const bool is_set = pl.is_set, has_hi = pl.has_hi;
    const uint64_t n_out = st->n_out, n_kmers_hint = st->n_kmers_hint;
    DBuf<uint64_t>& u_hi = st->u_hi; DBuf<uint64_t>& u_lo = st->u_lo; DBuf<uint32_t>& u_pay = st->u_pay;
    // ---- order-restoring sort: ascending key (filter.rs:205-206 bucket order + stable sort = global order) ----
    DBuf<uint32_t> t_pay;
    DBuf<uint64_t> t_hi, t_lo;
    size_t na = std::max<uint64_t>(n_out, 1);
    if (!st->use16) {
        ALLOC_OR_FAIL(c, t_pay, na); ALLOC_OR_FAIL(c, t_lo, na);
        if (has_hi) ALLOC_OR_FAIL(c, t_hi, na);
    }
    RecArrays A{has_hi ? u_hi.p : nullptr, u_lo.p, u_pay.p}, B{has_hi ? t_hi.p : nullptr, t_lo.p, t_pay.p};
    DBuf<uint64_t> o_hi, o_lo, o_set_off;
    DBuf<uint8_t> o_exts;
    DBuf<uint16_t> o_count;
    DBuf<uint32_t> o_set_val, setn, msk_sorted;
    ALLOC_OR_FAIL(c, o_hi, na); ALLOC_OR_FAIL(c, o_lo, na); ALLOC_OR_FAIL(c, o_exts, na);
    if (is_set) { ALLOC_OR_FAIL(c, msk_sorted, na); ALLOC_OR_FAIL(c, o_set_off, na + 1); }
    else ALLOC_OR_FAIL(c, o_count, na);
    uint64_t n_setval = 0;
    if (st->use16) {
        DBuf<uint4> t16;
        ALLOC_OR_FAIL(c, t16, na);
        DBG_TRY(sort_table_hybrid16(c, n_out, st->u16.p, t16.p, 2 * k, is_set, !c->opt("DBG_NO_HYBRID_SORT"), o_hi.p, o_lo.p, o_exts.p, o_count.p,
                                    nullptr, msk_sorted.p));
    } else
    DBG_TRY(sort_table_hybrid(c, n_out, A, B, 2 * k, is_set, !c->opt("DBG_NO_HYBRID_SORT"), o_hi.p, o_lo.p, o_exts.p, o_count.p,
                              nullptr, msk_sorted.p));
    DBuf<unsigned long long> msk64;
    if (is_set && pl.wide) {
        ALLOC_OR_FAIL(c, msk64, na);
        c->t_begin("wide_gather", n_out);
        wide_gather_kernel<<<cdiv(na, 256), 256, 0, c->stream>>>((uint32_t)n_out, o_exts.p, msk_sorted.p, st->w_rec.p, msk64.p);
        c->t_end();
        LAUNCH_CHECK(c, "wide_gather");
    }
    if (is_set && st->keep_masks) {
        if (!pl.wide) return c->fail(136, "fast path: keep_masks needs the wide colour layout");
        *st->keep_masks = std::move(msk64);
        o_set_off.release();
    } else if (is_set) {
        const uint32_t nb = cdiv(std::max<uint64_t>(n_out, 1), CSR_TILE);
        DBuf<uint64_t> part, part_sc;
        ALLOC_OR_FAIL(c, part, nb); ALLOC_OR_FAIL(c, part_sc, (size_t)nb + 1);
        if (pl.wide) csr_partials_kernel<unsigned long long><<<nb, CSR_THREADS, 0, c->stream>>>(msk64.p, (uint32_t)n_out, part.p);
        else csr_partials_kernel<uint32_t><<<nb, CSR_THREADS, 0, c->stream>>>(msk_sorted.p, (uint32_t)n_out, part.p);
        LAUNCH_CHECK(c, "csr_partials");
        DBG_TRY(scan_exclusive_u64(c, part.p, part_sc.p, nb));
        HIP_TRY(c, hipMemcpyAsync(&n_setval, part_sc.p + nb, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        ALLOC_OR_FAIL(c, o_set_val, std::max<uint64_t>(n_setval, 1));
        c->t_begin("set_csr", n_out);
        if (pl.wide) csr_apply_kernel<unsigned long long><<<nb, CSR_THREADS, 0, c->stream>>>(msk64.p, (uint32_t)n_out, part_sc.p, o_set_off.p, o_set_val.p, pl.linv);
        else csr_apply_kernel<uint32_t><<<nb, CSR_THREADS, 0, c->stream>>>(msk_sorted.p, (uint32_t)n_out, part_sc.p, o_set_off.p, o_set_val.p, pl.linv);
        c->t_end();
        LAUNCH_CHECK(c, "csr_apply");
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    memset(out, 0, sizeof(*out));
    out->n = n_out;
    out->key_hi = o_hi.take(); out->key_lo = o_lo.take(); out->exts = o_exts.take(); out->count = o_count.take();
    out->set_off = o_set_off.take(); out->set_val = o_set_val.take(); out->n_set_val = n_setval;
    out->n_kmer_instances = n_kmers_hint; out->n_passes = 1; out->on_device = 1;
    if (st->report_all) {
        // all_kmers (filter.rs:208-212): every distinct key in the same ascending order; keys only, so the payload
        // columns of the sorter are scratch
        const uint64_t n_all = st->n_all;
        const size_t naa = std::max<uint64_t>(n_all, 1);
        DBuf<uint64_t> b_hi, b_lo, f_hi, f_lo;
        DBuf<uint32_t> a_pay, b_pay;
        DBuf<uint8_t> x_exts;
        DBuf<uint16_t> x_count;
        ALLOC_OR_FAIL(c, a_pay, naa); ALLOC_OR_FAIL(c, b_pay, naa); ALLOC_OR_FAIL(c, b_lo, naa);
        ALLOC_OR_FAIL(c, f_hi, naa); ALLOC_OR_FAIL(c, f_lo, naa); ALLOC_OR_FAIL(c, x_exts, naa); ALLOC_OR_FAIL(c, x_count, naa);
        if (has_hi) ALLOC_OR_FAIL(c, b_hi, naa);
        HIP_TRY(c, hipMemsetAsync(a_pay.p, 0, naa * 4, c->stream));
        RecArrays A2{has_hi ? st->a_hi.p : nullptr, st->a_lo.p, a_pay.p}, B2{has_hi ? b_hi.p : nullptr, b_lo.p, b_pay.p};
        DBG_TRY(sort_table_hybrid(c, n_all, A2, B2, 2 * k, false, !c->opt("DBG_NO_HYBRID_SORT"), f_hi.p, f_lo.p, x_exts.p, x_count.p, nullptr, nullptr));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        out->n_all = n_all;
        out->all_hi = f_hi.take(); out->all_lo = f_lo.take();
    }
    return 0;
}

static int fast_count(dbg_ctx* c, const FastPlan& pl, uint64_t min_obs, const uint64_t* recs, const uint64_t* recs_alt, uint32_t alt_from,
                      const uint64_t* seg_beg, const uint64_t* seg_end,
                      uint32_t n_src, uint64_t seg_stride, uint32_t nbins_local, uint64_t n_kmers_hint, uint64_t n_recs_hint,
                      dbg_kmer_table* out, bool report_all = false, DBuf<unsigned long long>* keep_masks = nullptr) {
    FastCountState st;
    DBG_TRY(fast_count_begin(c, pl, min_obs, n_kmers_hint, &st, report_all));
    st.keep_masks = keep_masks;
    DBG_TRY(fast_count_bins(c, &st, recs, recs_alt, alt_from, seg_beg, seg_end, n_src, seg_stride, nbins_local, n_kmers_hint, n_recs_hint));
    return fast_count_finish(c, &st, out);
}

// segment bounds of the direct layout: segment 0 = the bin's slab, segment 1 = its overflow records (bin order, own buffer)
__global__ void slab_bounds_kernel(const uint32_t* __restrict__ cursor, uint32_t slab_cap, const uint64_t* __restrict__ ovf_off,
                                   uint32_t nb, uint64_t* __restrict__ beg, uint64_t* __restrict__ end, unsigned long long* __restrict__ total) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t cnt = 0;
    if (b < nb) {
        cnt = cursor[b];
        beg[b] = (uint64_t)b * slab_cap;
        end[b] = (uint64_t)b * slab_cap + (cnt < slab_cap ? cnt : slab_cap);
        beg[nb + b] = ovf_off[b];
        end[nb + b] = ovf_off[b + 1];
    }
    unsigned long long v = cnt;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(total, v);
}

// Sharded flow: the all-to-all wants every destination's records contiguous and in bin order, so the slabs are
// compacted (one wave per bin, 16-byte moves: a coalesced copy) and the few overflow records are scattered behind them.
template <int RW>
__global__ void __launch_bounds__(256) slab_compact_kernel(const uint64_t* __restrict__ slab, uint32_t slab_cap, const uint32_t* __restrict__ cursor,
                                                            const uint64_t* __restrict__ bin_off, uint32_t nb, uint64_t* __restrict__ out,
                                                            uint64_t* __restrict__ ovf_base) {
    const uint32_t b = (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (b >= nb) return;
    const uint32_t cnt = cursor[b], n = cnt < slab_cap ? cnt : slab_cap;
    const uint64_t o = bin_off[b];
    if (lane == 0) ovf_base[b] = o + n;
    const uint64_t* src = slab + (uint64_t)b * slab_cap * RW;
    uint64_t* dst = out + o * RW;
    const uint32_t words = n * RW;
    if (RW % 2 == 0 && ((o * RW) & 1) == 0) {                       // 16-byte aligned on both sides (slab_cap is a multiple of 4)
        for (uint32_t i = lane * 2; i < words; i += 128) *(ulonglong2*)(dst + i) = *(const ulonglong2*)(src + i);
    } else {
        for (uint32_t i = lane; i < words; i += 64) dst[i] = src[i];
    }
}

// Sender-side duplicate merge of the sharded flow (round 3).  Reads that cover the same stretch of the genome produce
// identical records (the cut points depend on the sequence alone), and a rank that holds 1/W of the reads still holds
// coverage/W copies of each: before the exchange every bin's identical records are merged IN PLACE in its slab into one record
// that carries (copies - 1) in the WEIGHT_BITS above its meta bits (more than 16 copies: several records), and the bin's fill
// count shrinks.  One wave per bin: distinct records collect in a per-wave LDS staging area behind a small hash filter (the
// same claim / compare scheme as bin_count's staging, without barriers: a wave's LDS operations are performed in order) and
// are written back behind the wave's read position when the area fills up or the bin ends.  Any grouping of equal records is
// correct -- duplicates that meet in different fills just stay separate.  Bins that overflowed their slab are left alone.
// MG_S staged records per wave: 256 merge a two-rank job's bins (270 records, 130 distinct) as well as can be done; bins of
// under ~130 records (eight ranks) take the 128-slot instantiation, whose smaller LDS footprint keeps 28 waves per CU resident
// instead of 12 (1.85 ms against 2.87 per 2*10^7 reads).
constexpr uint32_t MG_WAVES = 4, MG_COUNTERS = 4096;
template <int RW, uint32_t MG_S>
__global__ void __launch_bounds__(64 * MG_WAVES) slab_merge_kernel(uint64_t* __restrict__ slab, uint32_t slab_cap, uint32_t* __restrict__ cursor, uint32_t nb,
                                                                   unsigned long long* __restrict__ merged_away) {
    constexpr uint32_t MG_F = 2 * MG_S;                      // the filter stays at most half full; its entries hold a 10-bit slot number
    static_assert(MG_S < 1023, "slot field");
    __shared__ uint64_t s_rec[MG_WAVES][RW * MG_S];          // staged records, word-major
    __shared__ uint32_t s_wt[MG_WAVES][MG_S];
    __shared__ uint32_t s_f[MG_WAVES][MG_F];                 // filter: (staged slot + 1, or 1023 = being staged) | hash bits << 10
    __shared__ uint32_t s_n[MG_WAVES];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t b = blockIdx.x * MG_WAVES + wave;
    if (b >= nb) return;
    const uint32_t cnt = cursor[b];
    if (cnt < 2 || cnt > slab_cap) return;
    uint64_t* const rec = slab + (uint64_t)b * slab_cap * RW;
    uint64_t* const R = s_rec[wave];
    uint32_t* const WT = s_wt[wave];
    uint32_t* const F = s_f[wave];
    constexpr uint32_t WCAP = 1u << WEIGHT_BITS;
    uint32_t m = 0, nst = 0;                                 // records written back / staged (wave-uniform)
    for (uint32_t i = lane; i < MG_F; i += 64) F[i] = 0;
    if (lane == 0) s_n[wave] = 0;
    __builtin_amdgcn_wave_barrier();
    auto flush = [&]() {
        for (uint32_t i0 = 0; i0 < nst; i0 += 64) {
            const uint32_t i = i0 + lane;
            const bool on = i < nst;
            uint32_t w = on ? WT[i] : 0u;
            const uint32_t ne = (w + WCAP - 1u) / WCAP;      // output records of this staged record
            uint32_t incl = ne;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= (uint32_t)d) incl += o; }
            const uint32_t tot = __shfl(incl, 63);
            if (on) {
                uint64_t a[RW];
#pragma unroll
                for (int j = 0; j < RW; j++) a[j] = R[j * MG_S + i];
                uint64_t* dst = rec + (uint64_t)(m + incl - ne) * RW;
                for (uint32_t e = 0; e < ne; e++, dst += RW) {
                    const uint32_t part = w < WCAP ? w : WCAP;
                    w -= part;
#pragma unroll
                    for (int j = 0; j < RW - 1; j++) dst[j] = a[j];
                    dst[RW - 1] = a[RW - 1] | ((uint64_t)(part - 1u) << META_BITS);
                }
            }
            m += tot;
        }
        nst = 0;
        for (uint32_t i = lane; i < MG_F; i += 64) F[i] = 0;
        if (lane == 0) s_n[wave] = 0;
        __builtin_amdgcn_wave_barrier();
    };
    for (uint32_t r0 = 0; r0 < cnt; r0 += 64) {
        // everything written back so far lies below r0 (one output record stands for at least one consumed record)
        if (nst + 64 > MG_S) flush();
        bool pend = r0 + lane < cnt;
        uint64_t a[RW];
#pragma unroll
        for (int j = 0; j < RW; j++) a[j] = pend ? rec[(uint64_t)(r0 + lane) * RW + j] : 0ull;
        uint64_t ha = a[0], hb = a[1];
        if (RW > 2) ha += a[2] * 0x9E3779B97F4A7C15ull;
        if (RW > 3) hb += a[3] * 0xC2B2AE3D27D4EB4Full;
        const uint64_t h = hash_key(ha, hb);
        const uint32_t tag = (uint32_t)(h >> 42) << 10;
        uint32_t sl = (uint32_t)h & (MG_F - 1);
        while (__ballot(pend) != 0ull) {
            if (pend) {
                for (;;) {                                   // to the first filter slot that is free or carries my hash bits
                    uint32_t v = F[sl];
                    if (v == 0u) {
                        v = atomicCAS(&F[sl], 0u, 1023u | tag);
                        if (v == 0u) {                       // first of its kind: stage it
                            const uint32_t mine = atomicAdd(&s_n[wave], 1u);
#pragma unroll
                            for (int j = 0; j < RW; j++) R[j * MG_S + mine] = a[j];
                            WT[mine] = 1u;
                            F[sl] = (mine + 1u) | tag;
                            pend = false;
                            break;
                        }
                    }
                    if ((v & ~1023u) == tag) break;
                    sl = (sl + 1u) & (MG_F - 1);
                }
            }
            __builtin_amdgcn_wave_barrier();                 // (every lane is out of the loop: what was claimed is now staged)
            if (pend) {
                const uint32_t r = (F[sl] & 1023u) - 1u;
                bool same = true;
#pragma unroll
                for (int j = 0; j < RW; j++) same = same && R[j * MG_S + r] == a[j];
                if (same) { atomicAdd(&WT[r], 1u); pend = false; }
                else sl = (sl + 1u) & (MG_F - 1);
            }
            __builtin_amdgcn_wave_barrier();
        }
        nst = s_n[wave];
    }
    flush();
    // (one counter would be a single hot address: same-address atomics complete at ~1e8/s, and there are millions of bins)
    if (lane == 0) { cursor[b] = m; if (cnt != m) atomicAdd(&merged_away[b & (MG_COUNTERS - 1)], (unsigned long long)(cnt - m)); }
}

// scan -> slabs -> count -> sort for one plan (colour layout and label map already chosen)
static int fast_run(dbg_ctx* c, const SeqDev& s, FastPlan pl, uint64_t min_obs, bool report_all, uint64_t n_kmers, dbg_kmer_table* out,
                    DBuf<unsigned long long>* keep_masks = nullptr) {
    FastScan st;
    DBG_TRY(fast_scan(c, s, pl, n_kmers, &st, true));
    const uint32_t nb = pl.nbins * NCLS;
    DBuf<uint64_t> ovf_off, ovf_recs, seg;
    DBuf<unsigned long long> total;
    ALLOC_OR_FAIL(c, ovf_off, (size_t)nb + 1);
    DBG_TRY(fast_bin_offsets(c, &st, ovf_off.p, false));          // st.n_recs = records that did not fit their slab
    ALLOC_OR_FAIL(c, ovf_recs, std::max<uint64_t>(st.n_recs * pl.rw, 1));
    DBuf<uint64_t> slab;
    DBuf<uint32_t> cursor;
    std::swap(slab, st.slab); std::swap(cursor, st.cursor);
    const uint32_t slab_cap = st.slab_cap;
    DBG_TRY(fast_scatter(c, &st, ovf_off.p, ovf_recs.p));
    ALLOC_OR_FAIL(c, seg, (size_t)nb * 4);
    ALLOC_OR_FAIL(c, total, 1);
    HIP_TRY(c, hipMemsetAsync(total.p, 0, 8, c->stream));
    slab_bounds_kernel<<<cdiv(nb, 256), 256, 0, c->stream>>>(cursor.p, slab_cap, ovf_off.p, nb, seg.p, seg.p + 2 * (size_t)nb, total.p);
    LAUNCH_CHECK(c, "slab_bounds");
    unsigned long long n_recs_total = 0;
    HIP_TRY(c, hipMemcpyAsync(&n_recs_total, total.p, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->t_begin("sk_records", n_recs_total);    // bookkeeping entry: units = super-k-mer records (no kernel)
    c->t_end();
    if (c->opt("DBG_DEBUG")) fprintf(stderr, "[fastpath] direct slabs: cap=%u records/bin, %llu records, %llu took the overflow path\n", slab_cap,
                                     n_recs_total, (unsigned long long)st.n_recs);
    const int rc_count = fast_count(c, pl, min_obs, slab.p, ovf_recs.p, 1, seg.p, seg.p + 2 * (size_t)nb, st.n_recs ? 2u : 1u, (uint64_t)nb,
                                    pl.nbins, n_kmers, n_recs_total, out, report_all, keep_masks);
    return rc_count;
}

#include "fast_manylabels.hpp"
#include "fast_labellists.hpp"

// returns 0 and sets *used = true when the fast path produced the table; *used = false means the
// caller must take the generic path (unsupported shape), nothing was written.
int filter_kmers_fast(dbg_ctx* c, const SeqDev& s, const dbg_filter_params* prm, uint64_t n_kmers, dbg_kmer_table* out,
                      bool* used) {
    *used = false;
    const bool is_set = prm->summarizer == DBG_COUNT_FILTER_SET;
    if (n_kmers == 0) return 0;
    FastPlan pl;
    if (!fast_make_plan(c, (int)prm->k, prm->stranded != 0, is_set, n_kmers, 0, &pl)) return 0;
    DBuf<uint8_t> lmap_buf;
    if (is_set) {
        bool ok;
        std::vector<uint32_t> many;                 // more than 64 distinct labels (all < 65536): the label-group passes
        DBG_TRY(fast_labels_prepare(c, s, &pl, &lmap_buf, &ok, &many));
        if (!ok) {
            // more than 64 distinct labels, or labels >= 65536: label lists (any alphabet, one pass, 33 Gkmer/s at the C2 shape whatever
            // the alphabet: fast_labellists.hpp).  Label groups (fast_manylabels.hpp: 65..1024 labels < 65536; 26 / 21 / 12.6 Gkmer/s at
            // 65 / 100 / 250 labels) stay for DBG_LABEL_LISTS=0 and for a device too full for the label buffer.
            const char* ll = c->opt("DBG_LABEL_LISTS");
            const bool groups_ok = !(many.empty() || many.size() > 64u * ML_MAX_GROUPS || c->opt("DBG_NO_LABEL_GROUPS"));
            if (!(ll && !strcmp(ll, "0"))) {                 // (its scan follows the gates of a host-boundary upload like fast_run's)
                DBG_TRY(filter_kmers_fast_lists(c, s, prm, n_kmers, out, used));
                if (*used) return 0;
            }
            DBG_TRY(c->wait_all_reads());
            if (!groups_ok) return 0;
            return filter_kmers_fast_many(c, s, prm, n_kmers, many, out, used);
        }
    }
    // the WIDE colour-set layout keeps two more words per table entry: with 1024-entry tables two workgroups still share a CU's
    // LDS, so its bins are half the size
    if (pl.wide && !c->opt("DBG_FAST_TARGET")) pl.nbins = (uint32_t)std::min<uint64_t>((uint64_t)pl.nbins * 2, (1ull << 23) - 1);
    DBG_TRY(fast_run(c, s, pl, prm->min_kmer_obs, prm->report_all_kmers != 0, n_kmers, out));
    *used = true;
    return 0;
}


// ------------------------------------------------------------------------------------------------
// Sharded counting for multi-GPU runs.  The bin space is global (every rank uses the same plan);
// rank r owns a contiguous range of bins.  Stage 1 (scan + scatter) runs on every rank over its own
// reads and leaves its super-k-mer records in bin order, so the slab for each destination rank is
// contiguous.  The caller exchanges the slabs (RCCL all-to-all over xGMI) and hands stage 2 the
// received records as n_src bin-ordered segments.
// ------------------------------------------------------------------------------------------------
#include <memory>
// The state one call leaves for the next lives in the ctx (dbg_state_slot): the library has no globals.
template <class T> static void slot_delete(void* p) { delete static_cast<T*>(p); }

extern "C" int dbg_count_kmer_instances_dev(dbg_ctx* c, const dbg_seqset* ds, uint32_t k, uint64_t* n_out) {
    HIP_TRY(c, hipSetDevice(c->device));
    SeqDev s{ds->words, ds->start, ds->length, ds->exts, ds->data, ds->data ? ds->data_width : 0u, ds->n_seqs, ds->n_words};
    return kmer_total(c, s, (int)k, n_out);
}

static int plan_from(dbg_ctx* c, const dbg_shard_plan* sp, FastPlan* pl) {
    if (sp->n_bins % NCLS) return c->fail(144, "n_bins must be a multiple of bin_group");
    if (!fast_make_plan(c, (int)sp->k, sp->stranded != 0, sp->summarizer == DBG_COUNT_FILTER_SET, sp->total_kmers, sp->n_bins / NCLS, pl))
        return c->fail(140, "sharded counting supports 16 <= k <= 64");
    // every rank must use the same colour layout: it follows from the plan's global max_label (no per-rank label map here)
    if (pl->is_set && sp->n_labels) {
        // a sparse alphabet agreed on by all ranks: colour i stands for labels[i] (ascending)
        if (sp->n_labels > 64) return c->fail(141, "sharded CountFilterSet: at most 64 distinct labels (dbg_shard_plan.n_labels)");
        for (uint32_t i = 0; i < sp->n_labels; i++) {
            if (sp->labels[i] >= 65536u || (i && sp->labels[i] <= sp->labels[i - 1]))
                return c->fail(141, "sharded CountFilterSet: dbg_shard_plan.labels must be ascending and < 65536");
            pl->linv.v[i] = sp->labels[i];
        }
        pl->linv.on = 1;
        pl->wide = sp->n_labels > 24;
    } else {
        if (pl->is_set && sp->max_label >= 64)
            return c->fail(141, "sharded CountFilterSet needs labels < 64 (dbg_shard_plan.max_label) or the list of at most 64 distinct labels (dbg_shard_plan.labels)");
        pl->wide = pl->is_set && sp->max_label >= 24;
    }
    // records of the sharded flow may carry weights when the record has the spare bits (every rank derives the same answer)
    pl->weighted = 64 * pl->nbw - 2 * (2 * pl->k - pl->p) - META_BITS >= WEIGHT_BITS;
    if (pl->wide && !sp->n_bins && !c->opt("DBG_FAST_TARGET")) pl->nbins = (uint32_t)std::min<uint64_t>((uint64_t)pl->nbins * 2, (1ull << 23) - 1);
    return 0;
}

extern "C" int dbg_seqset_max_label_dev(dbg_ctx* c, const dbg_seqset* ds, uint32_t* out) {
    if (!ds || !out) return c->fail(10, "null argument");
    HIP_TRY(c, hipSetDevice(c->device));
    SeqDev s{ds->words, ds->start, ds->length, ds->exts, ds->data, ds->data ? ds->data_width : 0u, ds->n_seqs, ds->n_words};
    return seq_max_label(c, s, out);
}

extern "C" int dbg_seqset_label_bitmap_dev(dbg_ctx* c, const dbg_seqset* ds, uint32_t* bitmap_out) {
    if (!ds || !bitmap_out) return c->fail(10, "null argument");
    HIP_TRY(c, hipSetDevice(c->device));
    SeqDev s{ds->words, ds->start, ds->length, ds->exts, ds->data, ds->data ? ds->data_width : 0u, ds->n_seqs, ds->n_words};
    std::vector<uint32_t> hb;
    DBG_TRY(seq_label_bitmap(c, s, &hb));
    std::copy(hb.begin(), hb.end(), bitmap_out);
    return 0;
}

extern "C" int dbg_shard_plan_make(dbg_ctx* c, dbg_shard_plan* sp) {
    FastPlan pl;
    DBG_TRY(plan_from(c, sp, &pl));                 // n_bins = 0: derived from total_kmers; otherwise the caller's bin count is kept
    sp->n_bins = pl.nbins * NCLS;
    sp->rec_words = (uint32_t)pl.rw;
    sp->bin_group = NCLS;
    return 0;
}

// scan of this rank's reads into the global bin space (+ the sender-side merge): leaves the slabs, the per-bin record counts
// (st->cursor) and the overflow records in *st
static int shard_scan_core(dbg_ctx* c, const dbg_seqset* ds, const dbg_shard_plan* sp, FastPlan pl, FastScan* st) {
    SeqDev s{ds->words, ds->start, ds->length, ds->exts, ds->data, ds->data ? ds->data_width : 0u, ds->n_seqs, ds->n_words};
    DBuf<uint8_t> lmap_buf;                          // label -> colour index (sparse alphabets); lives until the scan is done
    if (pl.lists) {}                                 // (label lists: the labels travel as they are, checked < 2^24 by the caller)
    else if (pl.is_set && sp->n_labels) {
        std::vector<uint32_t> hb;
        DBG_TRY(seq_label_bitmap(c, s, &hb));
        const uint32_t top = sp->labels[sp->n_labels - 1];
        std::vector<uint8_t> map((size_t)top + 1, 0);
        for (uint32_t i = 0; i < sp->n_labels; i++) { map[sp->labels[i]] = (uint8_t)i; hb[sp->labels[i] >> 5] &= ~(1u << (sp->labels[i] & 31)); }
        for (uint32_t w : hb)
            if (w) return c->fail(141, "sharded CountFilterSet: a label of this rank is not in dbg_shard_plan.labels (the list must hold the labels of ALL ranks)");
        ALLOC_OR_FAIL(c, lmap_buf, map.size());
        HIP_TRY(c, hipMemcpyAsync(lmap_buf.p, map.data(), map.size(), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));    // `map` leaves scope
        pl.lmap = lmap_buf.p;
    } else if (pl.is_set) {
        uint32_t mx = 0;
        DBG_TRY(seq_max_label(c, s, &mx));
        if (mx >= 64 || mx > std::max(sp->max_label, 23u))
            return c->fail(141, "sharded CountFilterSet: a label of this rank exceeds dbg_shard_plan.max_label (labels must be < 64, and max_label the maximum over all ranks)");
    }
    uint64_t n_kmers = 0;
    DBG_TRY(dbg_count_kmer_instances_dev(c, ds, sp->k, &n_kmers));
    const uint32_t nb = pl.nbins * NCLS;
    if (n_kmers) DBG_TRY(fast_scan(c, s, pl, n_kmers, st, true));
    else {
        st->pl = pl; st->slab_cap = 4;
        ALLOC_OR_FAIL(c, st->hist, nb); ALLOC_OR_FAIL(c, st->cursor, nb); ALLOC_OR_FAIL(c, st->slab, 4);
        HIP_TRY(c, hipMemsetAsync(st->hist.p, 0, (size_t)nb * 4, c->stream));
        HIP_TRY(c, hipMemsetAsync(st->cursor.p, 0, (size_t)nb * 4, c->stream));
    }
    if (n_kmers && sp->merge_dups && pl.weighted && st->slab_cap) {
        DBuf<unsigned long long> away;
        ALLOC_OR_FAIL(c, away, MG_COUNTERS);
        HIP_TRY(c, hipMemsetAsync(away.p, 0, 8 * MG_COUNTERS, c->stream));
        c->t_begin("slab_merge", 0);
        const uint32_t blocks = cdiv(nb, MG_WAVES);
        const bool small = st->slab_cap <= 176;            // slab_cap = 1.3 x the mean bin + 48: bins of up to ~100 records on average
#define MERGE(RW_) do { if (small) slab_merge_kernel<RW_, 128><<<blocks, 64 * MG_WAVES, 0, c->stream>>>(st->slab.p, st->slab_cap, st->cursor.p, nb, away.p); \
                        else slab_merge_kernel<RW_, 256><<<blocks, 64 * MG_WAVES, 0, c->stream>>>(st->slab.p, st->slab_cap, st->cursor.p, nb, away.p); } while (0)
        if (pl.rw == 2) MERGE(2); else if (pl.rw == 3) MERGE(3); else MERGE(4);
#undef MERGE
        c->t_end();
        LAUNCH_CHECK(c, "slab_merge");
        std::vector<unsigned long long> h_away(MG_COUNTERS);
        HIP_TRY(c, hipMemcpyAsync(h_away.data(), away.p, 8 * MG_COUNTERS, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        unsigned long long n_away = 0;
        for (unsigned long long v : h_away) n_away += v;
        st->n_merged_away = n_away;
        c->t_begin("sk_merged_away", n_away);  // bookkeeping entry: units = records the merge removed (no kernel)
        c->t_end();
    }
    return 0;
}

extern "C" int dbg_shard_scan_dev(dbg_ctx* c, const dbg_seqset* ds, const dbg_shard_plan* sp, uint64_t* n_recs,
                                  uint64_t* bin_off_dev) {
    HIP_TRY(c, hipSetDevice(c->device));
    c->t_clear();
    FastPlan pl;
    DBG_TRY(plan_from(c, sp, &pl));
    std::unique_ptr<FastScan> st(new FastScan());
    DBG_TRY(shard_scan_core(c, ds, sp, pl, st.get()));
    const uint32_t nb = pl.nbins * NCLS;
    // per-bin record counts (slab + overflow) -> exclusive offsets of the bin-ordered layout
    DBG_TRY(scan_exclusive_u32_u64(c, st->cursor.p, bin_off_dev, nb));
    HIP_TRY(c, hipMemcpyAsync(&st->n_recs, bin_off_dev + nb, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->t_begin("sk_records", st->n_recs);    // bookkeeping entry: units = super-k-mer records (no kernel)
    c->t_end();
    *n_recs = st->n_recs;
    c->shard_scan.reset(st.release(), slot_delete<FastScan>);
    return 0;
}

// slabs + overflow records -> recs_out in the order bin_off gives (any placement of the bins: bin b's records go to
// [bin_off[b], bin_off[b] + count[b])); releases the scan state's buffers
static int shard_scatter_core(dbg_ctx* c, FastScan* st, const uint64_t* bin_off_dev, uint64_t* recs_out_dev) {
    const uint32_t nb = st->pl.nbins * NCLS;
    DBuf<uint64_t> ovf_base;
    ALLOC_OR_FAIL(c, ovf_base, (size_t)nb + 1);
    c->t_begin("slab_compact", st->n_recs);
    const uint32_t blocks = cdiv((uint64_t)nb * 64, 256);
    if (st->pl.rw == 2) slab_compact_kernel<2><<<blocks, 256, 0, c->stream>>>(st->slab.p, st->slab_cap, st->cursor.p, bin_off_dev, nb, recs_out_dev, ovf_base.p);
    else if (st->pl.rw == 3) slab_compact_kernel<3><<<blocks, 256, 0, c->stream>>>(st->slab.p, st->slab_cap, st->cursor.p, bin_off_dev, nb, recs_out_dev, ovf_base.p);
    else slab_compact_kernel<4><<<blocks, 256, 0, c->stream>>>(st->slab.p, st->slab_cap, st->cursor.p, bin_off_dev, nb, recs_out_dev, ovf_base.p);
    c->t_end();
    LAUNCH_CHECK(c, "slab_compact");
    st->slab.release(); st->cursor.release();
    DBG_TRY(fast_scatter(c, st, ovf_base.p, recs_out_dev));           // records that did not fit their slab go behind it
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int dbg_shard_scatter_dev(dbg_ctx* c, const uint64_t* bin_off_dev, uint64_t* recs_out_dev) {
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->shard_scan.p) return c->fail(142, "dbg_shard_scatter_dev without a preceding dbg_shard_scan_dev");
    std::unique_ptr<FastScan> st(static_cast<FastScan*>(c->shard_scan.release()));
    return shard_scatter_core(c, st.get(), bin_off_dev, recs_out_dev);
}

extern "C" int dbg_shard_count_dev(dbg_ctx* c, const dbg_shard_plan* sp, const uint64_t* recs_dev, const uint64_t* seg_off_dev,
                                   uint32_t n_src, uint32_t n_bins_local, uint64_t n_kmers_hint, dbg_kmer_table* out) {
    HIP_TRY(c, hipSetDevice(c->device));
    FastPlan pl;
    DBG_TRY(plan_from(c, sp, &pl));
    if (n_src == 0 || n_src > 64) return c->fail(143, "n_src must be in 1..=64 (one record segment per source rank)");
    if (n_bins_local % NCLS) return c->fail(144, "n_bins_local must be a multiple of bin_group");
    DBG_TRY(fast_count(c, pl, sp->min_kmer_obs, recs_dev, recs_dev, n_src, seg_off_dev, seg_off_dev + 1, n_src, (uint64_t)n_bins_local + 1, n_bins_local / NCLS,
                       std::max<uint64_t>(n_kmers_hint, 1), 0, out));
    return 0;
}

// Chunked form of dbg_shard_count_dev: begin, then any number of bin ranges (each with its own record buffer and segment
// table -- e.g. one per all-to-all chunk, counted while the next chunk is on the wire), then one sort in finish.
extern "C" int dbg_shard_count_begin(dbg_ctx* c, const dbg_shard_plan* sp, uint64_t n_kmers_hint) {
    HIP_TRY(c, hipSetDevice(c->device));
    FastPlan pl;
    DBG_TRY(plan_from(c, sp, &pl));
    std::unique_ptr<FastCountState> st(new FastCountState());
    DBG_TRY(fast_count_begin(c, pl, sp->min_kmer_obs, std::max<uint64_t>(n_kmers_hint, 1), st.get()));
    c->shard_count.reset(st.release(), slot_delete<FastCountState>);
    return 0;
}

extern "C" int dbg_shard_count_bins_dev(dbg_ctx* c, const uint64_t* recs_dev, const uint64_t* seg_off_dev, uint32_t n_src,
                                        uint32_t n_bins_chunk, uint64_t n_kmers_units) {
    HIP_TRY(c, hipSetDevice(c->device));
    FastCountState* st = static_cast<FastCountState*>(c->shard_count.p);
    if (!st) return c->fail(145, "dbg_shard_count_bins_dev without dbg_shard_count_begin");
    if (n_src == 0 || n_src > 64) return c->fail(143, "n_src must be in 1..=64 (one record segment per source rank)");
    if (n_bins_chunk % NCLS) return c->fail(144, "n_bins_chunk must be a multiple of bin_group");
    return fast_count_bins(c, st, recs_dev, recs_dev, n_src, seg_off_dev, seg_off_dev + 1, n_src, (uint64_t)n_bins_chunk + 1,
                           n_bins_chunk / NCLS, n_kmers_units, 0);
}

extern "C" int dbg_shard_count_finish(dbg_ctx* c, dbg_kmer_table* out) {
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->shard_count.p) return c->fail(145, "dbg_shard_count_finish without dbg_shard_count_begin");
    std::unique_ptr<FastCountState> st(static_cast<FastCountState*>(c->shard_count.release()));
    return fast_count_finish(c, st.get(), out);
}

#include "fast_exchange.hpp"
