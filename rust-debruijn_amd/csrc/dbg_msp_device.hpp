// Device-side minimizer scanner shared by the reference-exact msp_sequence kernels (msp.hip) and the
// internal super-k-mer partitioner of the fast counting path (fastpath.hip).
// Restates Scanner::scan (src/msp.rs:207-276) with its exact tie rule; the score function is a
// template parameter (the reference passes a closure, msp.rs:177,305-311).
#pragma once
#include "dbg_device.hpp"

struct MspCfg {
    int k, p;
    const uint32_t* perm;   // device, [4^p] or null
    int rc;
    int lmer_words;
};

// reverse complement of a right-aligned p-mer (p <= 16) held in 32 bits: reverse the bit pairs, complement, re-align
__device__ __forceinline__ uint32_t pmer_rc(uint32_t pm, int p) {
    uint32_t r = __brev(pm);
    r = ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
    return (~r) >> (32 - 2 * p);
}

__device__ __forceinline__ uint32_t pmer_score(const MspCfg& c, uint32_t pm) {          // msp.rs:305-311
    uint32_t a = c.perm ? c.perm[pm] : pm;
    if (c.rc) {
        uint32_t r = pmer_rc(pm, c.p);
        uint32_t b = c.perm ? c.perm[r] : r;
        a = a < b ? a : b;
    }
    return a;
}

struct MinPosD { uint32_t val, pos, pmer; };

// find_min(start, stop) (msp.rs:218-228): rightmost minimal position in [start, stop]
template <class Score>
__device__ __forceinline__ MinPosD find_min(int p, const Score& score, const uint64_t* __restrict__ w, uint64_t last_word, uint64_t st,
                                            uint32_t start, uint32_t stop, uint32_t pmask) {
    uint32_t pm = (uint32_t)packed_get_pmer64(w, st + start, p, last_word);
    MinPosD best{score(pm), start, pm};
    for (uint32_t pos = start + 1; pos <= stop; pos++) {
        pm = ((pm << 2) | packed_get(w, st + pos + p - 1)) & pmask;
        uint32_t v = score(pm);
        if (v <= best.val) best = MinPosD{v, pos, pm};      // equal value: larger pos is "Less" (msp.rs:134-139)
    }
    return best;
}

// Runs Scanner::scan over one sequence and calls emit(start, len, minpos) per interval.
template <class Score, class Emit>
__device__ __forceinline__ void scan_sequence(int k, int p, const Score& score, const uint64_t* __restrict__ w, uint64_t last_word, uint64_t st,
                                              uint32_t m, Emit emit) {
    const uint32_t win = (uint32_t)(k - p);
    const uint32_t pmask = p >= 16 ? 0xffffffffu : ((1u << (2 * p)) - 1);
    MinPosD minp = find_min(p, score, w, last_word, st, 0, win, pmask);                                  // msp.rs:232
    uint32_t end_pm = (uint32_t)packed_get_pmer64(w, st + win, p, last_word);                    // msp.rs:233
    uint32_t cur_start = 0;
    const uint32_t nwin = m - (uint32_t)k + 1;
    for (uint32_t i = 1; i < nwin; i++) {                                              // msp.rs:237
        end_pm = ((end_pm << 2) | packed_get(w, st + i + win + p - 1)) & pmask;        // incr, msp.rs:239
        if (i > minp.pos) {                                                            // msp.rs:241
            MinPosD nm = find_min(p, score, w, last_word, st, i, i + win, pmask);
            emit(cur_start, i + (uint32_t)k - 1 - cur_start, minp);
            cur_start = i; minp = nm;
        } else {
            uint32_t v = score(end_pm);
            if (v < minp.val) {                                                        // msp.rs:244
                emit(cur_start, i + (uint32_t)k - 1 - cur_start, minp);
                cur_start = i; minp = MinPosD{v, i + win, end_pm};
            }
        }
    }
    emit(cur_start, m - cur_start, minp);                                              // msp.rs:266-273
}

