// Minimum-substring partitioning on the device: Scanner::scan + msp_sequence
// (src/msp.rs:207-276, :279-324) with the reference's exact tie rule:
//   * initial / expiry rescans take the RIGHTMOST minimal p-mer of the window (MinPos::cmp,
//     msp.rs:127-141, via std::cmp::min in find_min, msp.rs:218-228);
//   * a p-mer entering the window replaces the current minimizer only if STRICTLY smaller
//     (msp.rs:244-246); expiry is tested first (msp.rs:241).
// One lane scans one sequence (the event chain is sequential per read, ~2/(k-p+2) events per
// k-mer); a first pass counts pieces, an exclusive scan places them, a second pass emits
// (bucket, Exts::from_slice_bounds, start, len, minimizer_pos [, Lmer words]) in input order.
#include "dbg_internal.hpp"
#include "dbg_msp_device.hpp"
#include <algorithm>

namespace {

struct MspScore {
    MspCfg c;
    __device__ __forceinline__ uint32_t operator()(uint32_t pm) const { return pmer_score(c, pm); }
};

// reads with at most MSP_WAVE_POS p-mer positions are handled by the wave-per-read kernel below
constexpr uint32_t MSP_WAVE_POS = 256;
__device__ __forceinline__ bool wave_handles(const MspCfg& c, uint32_t m) { return m >= (uint32_t)c.k && m - (uint32_t)c.p + 1 <= MSP_WAVE_POS; }

__global__ void msp_count_kernel(SeqDev s, MspCfg c, uint32_t* __restrict__ counts) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.n) return;
    uint32_t m = s.length[i];
    if (wave_handles(c, m)) return;
    uint32_t n = 0;
    if (m >= (uint32_t)c.k) scan_sequence(c.k, c.p, MspScore{c}, s.words, s.start[i], m, [&](uint32_t, uint32_t, const MinPosD&) { n++; });
    counts[i] = n;                                                                     // m < k: empty (msp.rs:294-296)
}

__global__ void msp_emit_kernel(SeqDev s, MspCfg c, const uint64_t* __restrict__ piece_off,
                                uint32_t* __restrict__ bucket, uint8_t* __restrict__ exts, uint32_t* __restrict__ pstart,
                                uint16_t* __restrict__ plen, uint32_t* __restrict__ mpos, uint64_t* __restrict__ lmer) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.n) return;
    uint32_t m = s.length[i];
    if (m < (uint32_t)c.k || wave_handles(c, m)) return;
    const uint64_t st = s.start[i];
    const uint64_t* __restrict__ w = s.words;
    uint64_t o = piece_off[i];
    scan_sequence(c.k, c.p, MspScore{c}, w, st, m, [&](uint32_t start, uint32_t len, const MinPosD& mp) {
        uint32_t r = pmer_rc(mp.pmer, c.p);
        bucket[o] = mp.pmer < r ? mp.pmer : r;                                        // min_rc().to_u64() (msp.rs:115-117)
        uint32_t le = start > 0 ? (1u << packed_get(w, st + start - 1)) : 0u;         // lib.rs:645-660
        uint32_t re = start + len < m ? (1u << packed_get(w, st + start + len)) : 0u;
        exts[o] = (uint8_t)((re << 4) | le);
        pstart[o] = start;
        plen[o] = (uint16_t)len;
        mpos[o] = mp.pos;
        if (lmer) {                                                                    // V::from_slice (vmer.rs:125-134, lib.rs:356-363)
            uint64_t* lw = lmer + o * (uint64_t)c.lmer_words;
            for (int q = 0; q < c.lmer_words; q++) {
                uint32_t b0 = (uint32_t)q * 32;
                uint64_t v = 0;
                if (b0 < len) {
                    uint32_t nb = len - b0 < 32 ? len - b0 : 32;
                    v = packed_get_kmer(w, st + start + b0, (int)nb).lo << (64 - 2 * nb);   // left-align nb bases
                }
                if (q == c.lmer_words - 1) v |= (uint64_t)(len & 0xff);
                lw[q] = v;
            }
        }
        o++;
    });
}

// ------------------------------------------------------------------------------------------------
// Wave-per-read form of the same scanner for reads of up to MSP_WAVE_POS p-mer positions.
//
// Scanner::scan looks sequential, but the position it tracks only ever changes in two ways (msp.rs:237-263):
//   * ENTRY : a p-mer entering the window at e is STRICTLY smaller than the tracked one (msp.rs:244) -- the first such
//             e in (m, m + W - 1] (everything right of m that was already inside the window is >= v[m]: m is either the
//             rightmost minimum of a rescan or the newest position);
//   * EXPIRY: otherwise the window start passes m (msp.rs:241) and the rescan takes the RIGHTMOST minimum of
//             [m + 1, m + W] (msp.rs:218-228, MinPos::cmp :127-141).
// Both depend on m alone, so the tracked positions are the orbit of the first window's rightmost minimum under a function
// next(m) that the whole wave evaluates in a handful of instructions.  Pieces are then emitted one per lane.  W = k - p + 1.
// ------------------------------------------------------------------------------------------------
template <bool EMIT>
__global__ void __launch_bounds__(256) msp_wave_kernel(SeqDev s, MspCfg c, uint32_t* __restrict__ counts, const uint64_t* __restrict__ piece_off,
                                                       uint32_t* __restrict__ bucket, uint8_t* __restrict__ exts, uint32_t* __restrict__ pstart,
                                                       uint16_t* __restrict__ plen, uint32_t* __restrict__ mpos, uint64_t* __restrict__ lmer) {
    __shared__ uint32_t s_v[4][MSP_WAVE_POS];
    __shared__ uint16_t s_ps[4][MSP_WAVE_POS], s_pl[4][MSP_WAVE_POS], s_pm[4][MSP_WAVE_POS];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t* v = s_v[wv];
    uint16_t *ps = s_ps[wv], *pl = s_pl[wv], *pmn = s_pm[wv];
    const uint64_t gwave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t W = (uint32_t)(c.k - c.p + 1), win = W - 1;
    const uint64_t* __restrict__ w = s.words;
    const uint16_t NONE = 0xffffu;
    for (uint64_t i = gwave; i < s.n; i += nwaves) {
        const uint32_t m = s.length[i];
        if (!wave_handles(c, m)) { if (!EMIT && m < (uint32_t)c.k && lane == 0) counts[i] = 0; continue; }
        const uint64_t st = s.start[i];
        const uint32_t npos = m - (uint32_t)c.p + 1, nwin = m - (uint32_t)c.k + 1;
        // 1. scores of all p-mers
        for (uint32_t j = lane; j < npos; j += 64) v[j] = pmer_score(c, (uint32_t)packed_get_kmer(w, st + j, c.p).lo);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // 2. orbit of the first window's rightmost minimum.  next(m) is evaluated for the tracked position only, by the
        //    whole wave: lanes look at the W positions right of m -- the first strictly smaller entry within W - 1 is an
        //    ENTRY (ballot + count-trailing-zeros), otherwise the rightmost minimum of [m+1, m+W] follows the EXPIRY
        //    (wave min of (value, -position)).
        auto rightmost_min = [&](uint32_t lo, uint32_t hi) -> uint32_t {                   // positions [lo, hi], hi < npos
            uint64_t best = ~0ull;
            for (uint32_t j = lo + lane; j <= hi; j += 64) {
                const uint64_t key = ((uint64_t)v[j] << 32) | (uint32_t)(0xffffu - j);       // smaller value, then larger position
                best = key < best ? key : best;
            }
            for (int d = 32; d; d >>= 1) { const uint64_t o2 = __shfl_xor(best, d); best = o2 < best ? o2 : best; }
            return 0xffffu - (uint32_t)(best & 0xffffu);
        };
        uint32_t np = 0, cur = rightmost_min(0, win), cur_start = 0;
        for (;;) {
            const uint32_t vm = v[cur];
            uint32_t e = NONE;
            const uint32_t last_entry = cur + W - 1 < npos - 1 ? cur + W - 1 : npos - 1;     // entries tested before m expires
            for (uint32_t j0 = cur + 1; j0 <= last_entry && e == NONE; j0 += 64) {
                const uint32_t j = j0 + lane;
                const uint64_t sm = __ballot(j <= last_entry && v[j < npos ? j : npos - 1] < vm);
                if (sm) e = j0 + (uint32_t)__ffsll((long long)sm) - 1u;
            }
            uint32_t i_change, nxt;
            if (e != NONE) { i_change = e - win; nxt = e; }                                 // ENTRY at window e - win (msp.rs:244)
            else if (cur + 1 < nwin) { i_change = cur + 1; nxt = rightmost_min(cur + 1, cur + W); }   // EXPIRY (msp.rs:241)
            else break;
            if (lane == 0) { ps[np] = (uint16_t)cur_start; pl[np] = (uint16_t)(i_change + (uint32_t)c.k - 1 - cur_start); pmn[np] = (uint16_t)cur; }
            np++;
            cur_start = i_change; cur = nxt;
        }
        if (lane == 0) { ps[np] = (uint16_t)cur_start; pl[np] = (uint16_t)(m - cur_start); pmn[np] = (uint16_t)cur; }   // msp.rs:266-273
        np++;
        if (!EMIT) { if (lane == 0) counts[i] = np; continue; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // 4. one piece per lane
        const uint64_t o0 = piece_off[i];
        for (uint32_t t = lane; t < np; t += 64) {
            const uint32_t start = ps[t], len = pl[t], mp = pmn[t];
            const uint64_t o = o0 + t;
            const uint32_t pmer = (uint32_t)packed_get_kmer(w, st + mp, c.p).lo;
            const uint32_t r = pmer_rc(pmer, c.p);
            bucket[o] = pmer < r ? pmer : r;                                          // min_rc().to_u64() (msp.rs:115-117)
            const uint32_t le = start > 0 ? (1u << packed_get(w, st + start - 1)) : 0u;   // lib.rs:645-660
            const uint32_t re = start + len < m ? (1u << packed_get(w, st + start + len)) : 0u;
            exts[o] = (uint8_t)((re << 4) | le);
            pstart[o] = start;
            plen[o] = (uint16_t)len;
            mpos[o] = mp;
            if (lmer) {                                                                // V::from_slice (vmer.rs:125-134, lib.rs:356-363)
                uint64_t* lw = lmer + o * (uint64_t)c.lmer_words;
                for (int q = 0; q < c.lmer_words; q++) {
                    const uint32_t b0 = (uint32_t)q * 32;
                    uint64_t x = 0;
                    if (b0 < len) {
                        const uint32_t nb = len - b0 < 32 ? len - b0 : 32;
                        x = packed_get_kmer(w, st + start + b0, (int)nb).lo << (64 - 2 * nb);   // left-align nb bases
                    }
                    if (q == c.lmer_words - 1) x |= (uint64_t)(len & 0xff);
                    lw[q] = x;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}
}  // namespace

static int validate_msp(dbg_ctx* c, const dbg_seqset* s, const dbg_msp_params* p) {
    if (!s || !p) return c->fail(10, "null argument");
    if (p->k < 1 || p->k > 64 * 1024) return c->fail(30, "k out of range");
    if (p->p < 1 || p->p > 16) return c->fail(31, "p must be in 1..=16 (bucket is a u32, msp.rs:321)");
    if (p->p > p->k) return c->fail(32, "p must not exceed k");
    if (p->lmer_words) {
        if (p->lmer_words > 6) return c->fail(33, "Lmer is implemented for [u64; 1..=6] (vmer.rs:182-206)");
        uint64_t max_len = ((uint64_t)p->lmer_words * 64 - 8) / 2;                     // vmer.rs:119-121
        if (max_len < 2ull * p->k - p->p) return c->fail(34, "assertion failed: V::max_len() >= 2 * k - p (msp.rs:292)");
    }
    if (2ull * p->k - p->p > 65535) return c->fail(35, "piece length does not fit MspIntervalP.len (u16, msp.rs:105)");
    return 0;
}

extern "C" int dbg_msp_sequence_dev(dbg_ctx* c, const dbg_seqset* ds, const dbg_msp_params* p, dbg_msp_pieces* out) {
    DBG_TRY(validate_msp(c, ds, p));
    HIP_TRY(c, hipSetDevice(c->device));
    c->t_clear();
    memset(out, 0, sizeof(*out));
    SeqDev s{ds->words, ds->start, ds->length, ds->exts, ds->data, ds->data ? ds->data_width : 0u, ds->n_seqs, ds->n_words};
    MspCfg cfg{(int)p->k, (int)p->p, p->permutation, p->rc, (int)p->lmer_words};
    DBuf<uint32_t> counts;
    DBuf<uint64_t> off;
    ALLOC_OR_FAIL(c, counts, std::max<uint64_t>(s.n, 1));
    ALLOC_OR_FAIL(c, off, s.n + 1);
    if (s.n) {
        const uint32_t wblocks = (uint32_t)std::min<uint64_t>(cdiv(s.n, 4), 256ull * 64);
        c->t_begin("msp_count", s.n);
        msp_count_kernel<<<cdiv(s.n, 64), 64, 0, c->stream>>>(s, cfg, counts.p);            // long reads only
        msp_wave_kernel<false><<<wblocks, 256, 0, c->stream>>>(s, cfg, counts.p, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        c->t_end();
        LAUNCH_CHECK(c, "msp_count");
    }
    DBG_TRY(scan_exclusive_u32_u64(c, counts.p, off.p, s.n));
    uint64_t np = 0;
    HIP_TRY(c, hipMemcpyAsync(&np, off.p + s.n, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    DBuf<uint32_t> bucket, pstart, mpos;
    DBuf<uint8_t> exts;
    DBuf<uint16_t> plen;
    DBuf<uint64_t> lmer;
    ALLOC_OR_FAIL(c, bucket, np); ALLOC_OR_FAIL(c, pstart, np); ALLOC_OR_FAIL(c, mpos, np);
    ALLOC_OR_FAIL(c, exts, np); ALLOC_OR_FAIL(c, plen, np);
    if (p->lmer_words) ALLOC_OR_FAIL(c, lmer, np * p->lmer_words);
    if (s.n && np) {
        c->t_begin("msp_emit", s.n);
        msp_emit_kernel<<<cdiv(s.n, 64), 64, 0, c->stream>>>(s, cfg, off.p, bucket.p, exts.p, pstart.p, plen.p, mpos.p, lmer.p);
        msp_wave_kernel<true><<<(uint32_t)std::min<uint64_t>(cdiv(s.n, 4), 256ull * 64), 256, 0, c->stream>>>(s, cfg, nullptr, off.p, bucket.p, exts.p,
                                                                                                       pstart.p, plen.p, mpos.p, lmer.p);
        c->t_end();
        LAUNCH_CHECK(c, "msp_emit");
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    out->n_pieces = np;
    out->piece_off = off.take(); out->bucket = bucket.take(); out->exts = exts.take(); out->start = pstart.take();
    out->len = plen.take(); out->minimizer_pos = mpos.take(); out->lmer = lmer.take();
    out->on_device = 1;
    return 0;
}

extern "C" void dbg_free_pieces(dbg_ctx* c, dbg_msp_pieces* pc) {
    if (!pc) return;
    void* ptrs[] = {pc->piece_off, pc->bucket, pc->exts, pc->start, pc->len, pc->minimizer_pos, pc->lmer};
    for (void* q : ptrs) {
        if (!q) continue;
        if (pc->on_device) { if (c) c->dfree(q); } else free(q);
    }
    memset(pc, 0, sizeof(*pc));
}

struct DevSeqSet;
int upload_seqset(dbg_ctx* c, const dbg_seqset* hs, DevSeqSet* d);

// host-pointer form (the Rust call site's shape, one call for a whole batch of reads)
extern "C" int dbg_msp_sequence(dbg_ctx* c, const dbg_seqset* hs, const dbg_msp_params* p, dbg_msp_pieces* out) {
    DBG_TRY(validate_msp(c, hs, p));
    HIP_TRY(c, hipSetDevice(c->device));
    for (uint64_t i = 0; i < hs->n_seqs; i++) {
        if (hs->length[i] && (hs->start[i] + hs->length[i] + 31) / 32 > hs->n_words) return c->fail(16, "sequence runs past n_words");
    }
    uint64_t n = hs->n_seqs;
    DBuf<uint64_t> words, start;
    DBuf<uint32_t> length, perm;
    ALLOC_OR_FAIL(c, words, std::max<uint64_t>(hs->n_words, 1));
    ALLOC_OR_FAIL(c, start, std::max<uint64_t>(n, 1));
    ALLOC_OR_FAIL(c, length, std::max<uint64_t>(n, 1));
    if (hs->n_words) HIP_TRY(c, hipMemcpyAsync(words.p, hs->words, hs->n_words * 8, hipMemcpyHostToDevice, c->stream));
    if (n) {
        HIP_TRY(c, hipMemcpyAsync(start.p, hs->start, n * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(length.p, hs->length, n * 4, hipMemcpyHostToDevice, c->stream));
    }
    dbg_msp_params dp = *p;
    if (p->permutation) {
        uint64_t np = 1ull << (2 * p->p);
        ALLOC_OR_FAIL(c, perm, np);
        HIP_TRY(c, hipMemcpyAsync(perm.p, p->permutation, np * 4, hipMemcpyHostToDevice, c->stream));
        dp.permutation = perm.p;
    }
    dbg_seqset dv = *hs;
    dv.words = words.p; dv.start = start.p; dv.length = length.p; dv.exts = nullptr; dv.data = nullptr;
    dbg_msp_pieces dev;
    DBG_TRY(dbg_msp_sequence_dev(c, &dv, &dp, &dev));
    memset(out, 0, sizeof(*out));
    out->n_pieces = dev.n_pieces;
    uint64_t np = dev.n_pieces;
#define CP(field, T, cnt)                                                                             \
    if (dev.field) {                                                                                  \
        out->field = (T*)malloc(std::max<size_t>((size_t)(cnt) * sizeof(T), 1));                      \
        if ((cnt)) HIP_TRY(c, hipMemcpyAsync(out->field, dev.field, (size_t)(cnt) * sizeof(T), hipMemcpyDeviceToHost, c->stream)); \
    }
    CP(piece_off, uint64_t, n + 1) CP(bucket, uint32_t, np) CP(exts, uint8_t, np) CP(start, uint32_t, np)
    CP(len, uint16_t, np) CP(minimizer_pos, uint32_t, np) CP(lmer, uint64_t, np * p->lmer_words)
#undef CP
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    dbg_free_pieces(c, &dev);
    return 0;
}
