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

// Three kernels share the work by read length and shape (all three restate the same Scanner::scan):
//   lane kernel    (default)  reads of up to MSP_LANE_MAX bases when p <= 11 and the window is at most MSP_LANE_W positions:
//                             one read per lane, in-place van Herk window minimum in LDS, wave-uniform control flow
//   wave kernel               other reads with at most MSP_WAVE_POS p-mer positions: one read per wavefront
//   literal kernel            everything longer: one read per lane, the reference's loop as it stands
constexpr uint32_t MSP_WAVE_POS = 256;
constexpr uint32_t MSP_LANE_MAX = 1024, MSP_LANE_W = 96;
__host__ __device__ __forceinline__ bool lane_cfg_ok(const MspCfg& c) { return c.p <= 11 && (uint32_t)(c.k - c.p + 1) <= MSP_LANE_W && c.k - c.p + 1 >= 2; }
__device__ __forceinline__ bool lane_handles(const MspCfg& c, uint32_t m, bool lane_on) { return lane_on && m >= (uint32_t)c.k && m <= MSP_LANE_MAX; }
__device__ __forceinline__ bool wave_handles(const MspCfg& c, uint32_t m, bool lane_on) {
    return !lane_handles(c, m, lane_on) && m >= (uint32_t)c.k && m - (uint32_t)c.p + 1 <= MSP_WAVE_POS;
}

__global__ void msp_count_kernel(SeqDev s, MspCfg c, uint32_t* __restrict__ counts, bool lane_on) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.n) return;
    uint32_t m = s.length[i];
    if (wave_handles(c, m, lane_on) || lane_handles(c, m, lane_on)) return;
    uint32_t n = 0;
    if (m >= (uint32_t)c.k) scan_sequence(c.k, c.p, MspScore{c}, s.words, (s.n_words ? s.n_words - 1 : 0), s.start[i], m, [&](uint32_t, uint32_t, const MinPosD&) { n++; });
    counts[i] = n;                                                                     // m < k: empty (msp.rs:294-296)
}

__global__ void msp_emit_kernel(SeqDev s, MspCfg c, const uint64_t* __restrict__ piece_off,
                                uint32_t* __restrict__ bucket, uint8_t* __restrict__ exts, uint32_t* __restrict__ pstart,
                                uint16_t* __restrict__ plen, uint32_t* __restrict__ mpos, uint64_t* __restrict__ lmer, bool lane_on) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.n) return;
    uint32_t m = s.length[i];
    if (m < (uint32_t)c.k || wave_handles(c, m, lane_on) || lane_handles(c, m, lane_on)) return;
    const uint64_t st = s.start[i];
    const uint64_t* __restrict__ w = s.words;
    uint64_t o = piece_off[i];
    scan_sequence(c.k, c.p, MspScore{c}, w, (s.n_words ? s.n_words - 1 : 0), st, m, [&](uint32_t start, uint32_t len, const MinPosD& mp) {
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
                    v = packed_get_pmer64(w, st + start + b0, (int)nb, (s.n_words ? s.n_words - 1 : 0)) << (64 - 2 * nb);   // left-align nb bases
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
                                                       uint16_t* __restrict__ plen, uint32_t* __restrict__ mpos, uint64_t* __restrict__ lmer, bool lane_on) {
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
        if (!wave_handles(c, m, lane_on)) { if (!EMIT && m < (uint32_t)c.k && lane == 0) counts[i] = 0; continue; }
        const uint64_t st = s.start[i];
        const uint32_t npos = m - (uint32_t)c.p + 1, nwin = m - (uint32_t)c.k + 1;
        // 1. scores of all p-mers
        for (uint32_t j = lane; j < npos; j += 64) v[j] = pmer_score(c, (uint32_t)packed_get_pmer64(w, st + j, c.p, (s.n_words ? s.n_words - 1 : 0)));
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
            const uint32_t pmer = (uint32_t)packed_get_pmer64(w, st + mp, c.p, (s.n_words ? s.n_words - 1 : 0));
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
                        x = packed_get_pmer64(w, st + start + b0, (int)nb, (s.n_words ? s.n_words - 1 : 0)) << (64 - 2 * nb);   // left-align nb bases
                    }
                    if (q == c.lmer_words - 1) x |= (uint64_t)(len & 0xff);
                    lw[q] = x;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}
// ------------------------------------------------------------------------------------------------
// Lane-per-read form (the default for sequencing reads).  A wavefront takes 64 reads, one per lane; every lane walks its own
// read base by base with wave-uniform control flow.  What makes Scanner::scan fit that shape:
//   * the rescan of msp.rs:218-228 -- the RIGHTMOST minimum of a window -- is the window minimum of the packed key
//     (score << 10 | 1023 - position), and the minima of ALL windows cost 4 LDS operations per position with the in-place
//     van Herk / Gil-Werman scheme (block of W keys per lane in LDS; window = suffix of the previous block + prefix of the
//     current one); the scanner simply reads the one it needs when its minimizer expires (msp.rs:241);
//   * the other transition -- an entering p-mer STRICTLY smaller than the tracked one (msp.rs:244) -- compares two registers.
// A first launch counts the pieces of every read, the second writes (start, len, minimizer_pos) at piece_off[read] + t, and
// msp_finish_kernel derives bucket, Exts and the Lmer words of every piece from those.
// ------------------------------------------------------------------------------------------------
// MODE 0: count the pieces of every read.  MODE 1: write them at piece_off[read] + t.  MODE 2 (round 3): count AND keep the first
// MSP_STASH pieces of every read in stash[read * MSP_STASH + t] -- the second scan of the reads is then only needed for the reads
// with more pieces than that (MODE 3: MODE 1 restricted to them; a 150-base read has ~7), and msp_finish_kernel takes the pieces
// from the stash: one pass over the reads instead of two.
constexpr uint32_t MSP_STASH = 16;
template <int MODE>
__global__ void __launch_bounds__(64) msp_lane_kernel(SeqDev s, MspCfg c, uint32_t* __restrict__ counts, const uint64_t* __restrict__ piece_off,
                                                      uint32_t* __restrict__ packed, uint32_t* __restrict__ stash, uint32_t* __restrict__ flag) {
    constexpr bool EMIT = MODE == 1 || MODE == 3;
    extern __shared__ uint32_t s_dyn[];
    const uint32_t lane = threadIdx.x;
    const int k = c.k, p = c.p;
    const uint32_t W = (uint32_t)(k - p + 1);
    uint32_t* SV = s_dyn + lane;                                    // slot j of this lane: SV[j * 64]
    SV[W * 64] = 0xffffffffu;                                       // S[W]: the empty suffix
    const uint64_t* __restrict__ w = s.words;
    const uint64_t last_word = s.n_words ? s.n_words - 1 : 0;
    const uint32_t pmask = (1u << (2 * p)) - 1u, top = 2u * (uint32_t)(p - 1);      // p <= 11
    for (uint64_t rb = (uint64_t)blockIdx.x * 64; rb < s.n; rb += (uint64_t)gridDim.x * 64) {
        const uint64_t my = rb + lane;
        uint32_t m = 0;
        uint64_t st = 0;
        if (my < s.n) { m = s.length[my]; st = s.start[my]; }
        if (!lane_handles(c, m, true)) m = 0;                       // other kernels (or nothing: m < k gives no pieces)
        if (MODE == 3 && m && counts[my] <= MSP_STASH) m = 0;       // its pieces are in the stash
        const uint32_t nwin = m ? m - (uint32_t)k + 1 : 0u;
        uint32_t mmax = m;
#pragma unroll
        for (int d = 32; d; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)mmax, d); mmax = o > mmax ? o : mmax; }
        mmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)mmax);
        if (mmax == 0) continue;
        const uint64_t w_first = st >> 5;
        const uint64_t* __restrict__ wr = w + w_first;
        const uint32_t sb2 = (uint32_t)(st & 31) * 2u;
        const uint32_t last_rel = (uint32_t)(last_word - w_first < 0x7fffffffull ? last_word - w_first : 0x7fffffffull);
        uint64_t prev = wr[0];
        uint32_t widx = 1, fw = 0, rcp = 0, P = 0xffffffffu, cur = 0, lo_half = 0;
        uint32_t m_key = 0, cur_start = 0, np = 0;                  // tracked minimizer as packed key; open piece; pieces so far
        uint32_t j = 0;
        const uint64_t o0 = EMIT && m ? piece_off[my] : 0ull;

        auto roll = [&]() {
            const uint32_t b = cur >> 30;
            cur <<= 2;
            fw = ((fw << 2) | b) & pmask;
            rcp = (rcp >> 2) | ((3u - b) << top);
        };
        auto key_of = [&](uint32_t e) -> uint32_t {                 // msp.rs:305-311 score, then (score, rightmost first)
            uint32_t a = c.perm ? c.perm[fw] : fw;
            if (c.rc) { const uint32_t b = c.perm ? c.perm[rcp] : rcp; a = a < b ? a : b; }
            return (a << 10) | (1023u - e);
        };
        auto close = [&](uint32_t i_change) {                       // the piece [cur_start, i_change + k - 1) ends (msp.rs:250-262)
            // one 4-byte store per piece: start (10 bits) | len (11: at most 2k - p <= 2 * 106) << 10 | minimizer_pos (10) << 21
            const uint32_t x = cur_start | ((i_change + (uint32_t)k - 1 - cur_start) << 10) | ((1023u - (m_key & 1023u)) << 21);
            if (EMIT) packed[o0 + np] = x;
            if (MODE == 2 && np < MSP_STASH) stash[my * MSP_STASH + np] = x;
            np++;
            cur_start = i_change;
        };
        // window sw of every lane is complete: its rightmost minimum is `out`, the p-mer that just entered has key `ke`
        auto window = [&](uint32_t sw, uint32_t out, uint32_t ke) {
            if (sw == 0) { m_key = out; return; }                   // wave-uniform: msp.rs:232
            const bool live = sw < nwin;
            const bool expired = sw > 1023u - (m_key & 1023u);      // msp.rs:241
            const bool enters = (ke >> 10) < (m_key >> 10);         // msp.rs:244 (strict)
            if (live && (expired || enters)) { close(sw); m_key = expired ? out : ke; }
        };

        for (uint32_t t = 0; t < mmax;) {
            if ((t & 31u) == 0) {
                const uint64_t nw = wr[widx < last_rel ? widx : last_rel];
                widx++;
                const uint64_t chunk = (prev << sb2) | ((nw >> 1) >> (63u - sb2));
                prev = nw;
                cur = (uint32_t)(chunk >> 32);
                lo_half = (uint32_t)chunk;
            } else if ((t & 15u) == 0) cur = lo_half;
            uint32_t n = 16u - (t & 15u);
            n = mmax - t < n ? mmax - t : n;
            if (t + 1 < (uint32_t)p) {
                n = (uint32_t)p - 1 - t < n ? (uint32_t)p - 1 - t : n;
                for (uint32_t u = 0; u < n; u++) roll();
                t += n;
                continue;
            }
            n = W - j < n ? W - j : n;
            const uint32_t e = t + 1 - (uint32_t)p;
            if (e + 1 < W) {                                        // block 0 is being filled: no window yet
                n = W - 1 - e < n ? W - 1 - e : n;
                for (uint32_t u = 0; u < n; u++) {
                    roll();
                    const uint32_t h = key_of(e + u);
                    P = h < P ? h : P;
                    SV[(j + u) * 64] = h;
                }
            } else {
                uint32_t u = 0;
                while (u < n) {
                    uint32_t* q = SV + (j + u) * 64;
                    if (n - u >= 4) {
                        const uint32_t s1 = q[64], s2 = q[128], s3 = q[192], s4 = q[256];
                        uint32_t h;
                        roll(); h = key_of(e + u);     P = h < P ? h : P; q[0] = h;   window(e + u + 1 - W, s1 < P ? s1 : P, h);
                        roll(); h = key_of(e + u + 1); P = h < P ? h : P; q[64] = h;  window(e + u + 2 - W, s2 < P ? s2 : P, h);
                        roll(); h = key_of(e + u + 2); P = h < P ? h : P; q[128] = h; window(e + u + 3 - W, s3 < P ? s3 : P, h);
                        roll(); h = key_of(e + u + 3); P = h < P ? h : P; q[192] = h; window(e + u + 4 - W, s4 < P ? s4 : P, h);
                        u += 4;
                    } else {
                        const uint32_t s1 = q[64];
                        roll();
                        const uint32_t h = key_of(e + u);
                        P = h < P ? h : P;
                        q[0] = h;
                        window(e + u + 1 - W, s1 < P ? s1 : P, h);
                        u++;
                    }
                }
            }
            t += n;
            j += n;
            if (j == W) {                                           // block complete: keys -> suffix minima, in place
                uint32_t run = 0xffffffffu;
                uint32_t q = W;
                for (; q >= 5; q -= 4) {
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
        if (m) {                                                    // the last piece runs to the end of the read (msp.rs:266-273)
            const uint32_t x = cur_start | ((m - cur_start) << 10) | ((1023u - (m_key & 1023u)) << 21);
            if (EMIT) packed[o0 + np] = x;
            if (MODE == 2 && np < MSP_STASH) stash[my * MSP_STASH + np] = x;
            np++;
            if (!EMIT) counts[my] = np;
            if (MODE == 2 && np > MSP_STASH) *flag = 1u;            // (any number of lanes may store the same value)
        }
    }
}

__global__ void __launch_bounds__(256) perm_max_kernel(const uint32_t* __restrict__ v, uint64_t n, uint32_t* __restrict__ out) {
    uint32_t m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) m = max(m, v[i]);
    for (int d = 32; d; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// The pieces the lane kernel placed, unpacked and completed (bucket, Exts::from_slice_bounds, Lmer words): a wavefront takes 64
// reads, its lanes the reads' pieces in output order -- loads of the packed words and stores of all seven columns are
// coalesced; a piece finds its read by a 6-step search over the 64 piece offsets held one per lane.
__global__ void __launch_bounds__(256) msp_finish_kernel(SeqDev s, MspCfg c, const uint64_t* __restrict__ piece_off, uint32_t* __restrict__ bucket,
                                                         uint8_t* __restrict__ exts, uint32_t* __restrict__ pstart, uint16_t* __restrict__ plen,
                                                         uint32_t* __restrict__ mpos, uint64_t* __restrict__ lmer, const uint32_t* __restrict__ stash) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t rb = (((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 64;
    if (rb >= s.n) return;
    const uint64_t my = rb + lane < s.n ? rb + lane : s.n - 1;
    const uint64_t off_l = piece_off[rb + lane < s.n ? rb + lane : s.n];            // lanes past the end hold the total
    const uint32_t m_l = s.length[my];
    const uint64_t st_l = s.start[my];
    const bool lane_read = lane_handles(c, m_l, true);
    const uint64_t o_begin = __shfl(off_l, 0), o_end = piece_off[rb + 64 < s.n ? rb + 64 : s.n];
    const uint64_t* __restrict__ w = s.words;
    for (uint64_t ob = o_begin; ob < o_end; ob += 64) {                             // wave-uniform trip count: the shuffles need every lane
        const uint64_t o = ob + lane;
        uint32_t r = 0;                                                             // largest r with piece_off[rb + r] <= o
#pragma unroll
        for (int step = 32; step; step >>= 1) { const uint64_t x = __shfl(off_l, (int)(r + step)); if (x <= o) r += step; }
        const uint32_t m = __shfl(m_l, (int)r);
        const uint64_t st = __shfl(st_l, (int)r);
        const bool mine = __shfl((int)lane_read, (int)r) != 0;                      // else: a piece of another kernel's read
        const uint64_t off_r = __shfl(off_l, (int)r);                               // (before any lane leaves: shuffles need every lane)
        if (o >= o_end || !mine) continue;
        // the packed piece: from the read's stash (its first MSP_STASH pieces), else where the second scan left it
        const uint64_t t = o - off_r;
        const uint32_t x = stash && t < MSP_STASH ? stash[(rb + r) * MSP_STASH + t] : mpos[o];
        const uint32_t start = x & 1023u, len = (x >> 10) & 2047u, mp = x >> 21;
        pstart[o] = start; plen[o] = (uint16_t)len; mpos[o] = mp;
        const uint32_t pmer = (uint32_t)packed_get_pmer64(w, st + mp, c.p, (s.n_words ? s.n_words - 1 : 0));
        const uint32_t rcv = pmer_rc(pmer, c.p);
        bucket[o] = pmer < rcv ? pmer : rcv;                                      // min_rc().to_u64() (msp.rs:115-117)
        const uint32_t le = start > 0 ? (1u << packed_get(w, st + start - 1)) : 0u;   // lib.rs:645-660
        const uint32_t re = start + len < m ? (1u << packed_get(w, st + start + len)) : 0u;
        exts[o] = (uint8_t)((re << 4) | le);
        if (lmer) {                                                                // V::from_slice (vmer.rs:125-134, lib.rs:356-363)
            uint64_t* lw = lmer + o * (uint64_t)c.lmer_words;
            for (int q = 0; q < c.lmer_words; q++) {
                const uint32_t b0 = (uint32_t)q * 32;
                uint64_t v = 0;
                if (b0 < len) {
                    const uint32_t nb = len - b0 < 32 ? len - b0 : 32;
                    v = packed_get_pmer64(w, st + start + b0, (int)nb, (s.n_words ? s.n_words - 1 : 0)) << (64 - 2 * nb);   // left-align nb bases
                }
                if (q == c.lmer_words - 1) v |= (uint64_t)(len & 0xff);
                lw[q] = v;
            }
        }
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
    // DBG_MSP=wave: everything the lane kernel would take goes to the wave-per-read kernel instead (A/B measurements, tests)
    bool lane_on = lane_cfg_ok(cfg) && !(c->opt("DBG_MSP") && !strcmp(c->opt("DBG_MSP"), "wave"));
    if (lane_on && cfg.perm) {
        // the lane kernel packs (score, position) into 32 bits: scores must stay below 2^22 (true of any permutation of 0..4^p, p <= 11)
        DBuf<uint32_t> mx;
        ALLOC_OR_FAIL(c, mx, 1);
        HIP_TRY(c, hipMemsetAsync(mx.p, 0, 4, c->stream));
        perm_max_kernel<<<64, 256, 0, c->stream>>>(cfg.perm, 1ull << (2 * cfg.p), mx.p);
        LAUNCH_CHECK(c, "perm_max");
        uint32_t h = 0;
        HIP_TRY(c, hipMemcpyAsync(&h, mx.p, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        lane_on = h < (1u << 22);
    }
    const uint32_t W = p->k - p->p + 1;
    const size_t lane_lds = (size_t)(W + 1) * 64 * sizeof(uint32_t);
    const uint32_t lane_blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((s.n + 63) / 64, 256ull * 16 * 8));
    // the lane kernel's stash (64 bytes per read); without the memory for it, or with DBG_MSP=twopass, the reads are scanned twice
    DBuf<uint32_t> stash, stash_flag;
    uint32_t h_stash_flag = 0;
    if (s.n && lane_on && !(c->opt("DBG_MSP") && !strcmp(c->opt("DBG_MSP"), "twopass"))) {
        if (!stash.alloc(c, s.n * MSP_STASH)) stash.p = nullptr;
        ALLOC_OR_FAIL(c, stash_flag, 1);
        HIP_TRY(c, hipMemsetAsync(stash_flag.p, 0, 4, c->stream));
    }
    if (s.n) {
        const uint32_t wblocks = (uint32_t)std::min<uint64_t>(cdiv(s.n, 4), 256ull * 64);
        HIP_TRY(c, hipMemsetAsync(counts.p, 0, s.n * 4, c->stream));
        c->t_begin("msp_count", s.n);
        msp_count_kernel<<<cdiv(s.n, 64), 64, 0, c->stream>>>(s, cfg, counts.p, lane_on);            // long reads only
        msp_wave_kernel<false><<<wblocks, 256, 0, c->stream>>>(s, cfg, counts.p, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, lane_on);
        if (lane_on && stash.p) msp_lane_kernel<2><<<lane_blocks, 64, lane_lds, c->stream>>>(s, cfg, counts.p, nullptr, nullptr, stash.p, stash_flag.p);
        else if (lane_on) msp_lane_kernel<0><<<lane_blocks, 64, lane_lds, c->stream>>>(s, cfg, counts.p, nullptr, nullptr, nullptr, nullptr);
        c->t_end();
        LAUNCH_CHECK(c, "msp_count");
    }
    DBG_TRY(scan_exclusive_u32_u64(c, counts.p, off.p, s.n));
    uint64_t np = 0;
    HIP_TRY(c, hipMemcpyAsync(&np, off.p + s.n, 8, hipMemcpyDeviceToHost, c->stream));
    if (stash.p) HIP_TRY(c, hipMemcpyAsync(&h_stash_flag, stash_flag.p, 4, hipMemcpyDeviceToHost, c->stream));
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
        msp_emit_kernel<<<cdiv(s.n, 64), 64, 0, c->stream>>>(s, cfg, off.p, bucket.p, exts.p, pstart.p, plen.p, mpos.p, lmer.p, lane_on);
        msp_wave_kernel<true><<<(uint32_t)std::min<uint64_t>(cdiv(s.n, 4), 256ull * 64), 256, 0, c->stream>>>(s, cfg, nullptr, off.p, bucket.p, exts.p,
                                                                                                       pstart.p, plen.p, mpos.p, lmer.p, lane_on);
        if (lane_on) {
            // packed words wait in mpos (second scan) or in the stash
            if (!stash.p) msp_lane_kernel<1><<<lane_blocks, 64, lane_lds, c->stream>>>(s, cfg, nullptr, off.p, mpos.p, nullptr, nullptr);
            else if (h_stash_flag) msp_lane_kernel<3><<<lane_blocks, 64, lane_lds, c->stream>>>(s, cfg, counts.p, off.p, mpos.p, nullptr, nullptr);
            msp_finish_kernel<<<cdiv(s.n, 256), 256, 0, c->stream>>>(s, cfg, off.p, bucket.p, exts.p, pstart.p, plen.p, mpos.p, lmer.p, stash.p);
        }
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
