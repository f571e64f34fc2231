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

__global__ void msp_count_kernel(SeqDev s, MspCfg c, uint32_t* __restrict__ counts) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.n) return;
    uint32_t m = s.length[i];
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
    if (m < (uint32_t)c.k) return;
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
        c->t_begin("msp_count", s.n);
        msp_count_kernel<<<cdiv(s.n, 64), 64, 0, c->stream>>>(s, cfg, counts.p);
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
