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
// The internal partition uses its own minimizer scheme (p up to 13, bijectively hashed canonical p-mers
// for balance -- a `permutation` in the reference's terms, msp.rs:55-59); it is invisible in the result.
#include "dbg_internal.hpp"
#include "dbg_msp_device.hpp"
#include <algorithm>
#include <cstdlib>

namespace {

// ------------------------------------------------------------------------------------------------
// internal minimizer order + bin hash
// ------------------------------------------------------------------------------------------------
struct FastCfg {
    int k, p;
    int stranded;
    uint32_t nbins;
};

// bijection on 2p-bit integers (odd multiply and xorshift are both invertible mod 2^(2p))
__device__ __forceinline__ uint32_t mix_pmer(uint32_t x, int p) {
    const uint32_t mask = p >= 16 ? 0xffffffffu : ((1u << (2 * p)) - 1);
    x = (x * 0x9E3779B1u) & mask;
    x ^= x >> p;
    x = (x * 0x85EBCA6Bu) & mask;
    x ^= x >> p;
    return x;
}
struct FastScore {
    int p, stranded;
    __device__ __forceinline__ uint32_t operator()(uint32_t pm) const {
        if (!stranded) { uint32_t r = pmer_rc(pm, p); pm = pm < r ? pm : r; }
        return mix_pmer(pm, p);
    }
};
__device__ __forceinline__ uint32_t bin_of(const FastCfg& c, uint32_t pm) {
    if (!c.stranded) { uint32_t r = pmer_rc(pm, c.p); pm = pm < r ? pm : r; }
    uint32_t h = pm * 0xC2B2AE35u + 0x27D4EB2Fu;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return (uint32_t)(((uint64_t)h * c.nbins) >> 32);
}

// ------------------------------------------------------------------------------------------------
// pass 1: count super-k-mers per bin        pass 2: write the records
// record = NBW base words (bases left-aligned, MSB first) + 1 meta word: len | exts << 8 | D1 << 16
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) sk_count_kernel(SeqDev s, FastCfg c, uint32_t* __restrict__ hist) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.n) return;
    uint32_t m = s.length[i];
    if (m < (uint32_t)c.k) return;
    FastScore sc{c.p, c.stranded};
    scan_sequence(c.k, c.p, sc, s.words, s.start[i], m, [&](uint32_t, uint32_t, const MinPosD& mp) {
        atomicAdd(&hist[bin_of(c, mp.pmer)], 1u);
    });
}

template <int NBW>
__global__ void __launch_bounds__(64) sk_emit_kernel(SeqDev s, FastCfg c, const uint64_t* __restrict__ bin_off,
                                                     uint32_t* __restrict__ cursor, uint64_t* __restrict__ recs) {
    constexpr int RW = NBW + 1;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.n) return;
    uint32_t m = s.length[i];
    if (m < (uint32_t)c.k) return;
    const uint64_t st = s.start[i];
    const uint64_t* __restrict__ w = s.words;
    const uint32_t sexts = s.exts ? s.exts[i] : 0u;
    uint32_t d1 = 0;
    if (s.data) d1 = s.data_width == 1 ? ((const uint8_t*)s.data)[i] : (s.data_width == 2 ? ((const uint16_t*)s.data)[i] : ((const uint32_t*)s.data)[i]);
    FastScore sc{c.p, c.stranded};
    scan_sequence(c.k, c.p, sc, w, st, m, [&](uint32_t start, uint32_t len, const MinPosD& mp) {
        uint32_t b = bin_of(c, mp.pmer);
        uint64_t r = bin_off[b] + atomicAdd(&cursor[b], 1u);
        uint64_t* o = recs + r * RW;
        // boundary Exts of the piece inside its read: neighbouring base, or the read's own seq_exts at the ends
        uint32_t le = start > 0 ? (1u << packed_get(w, st + start - 1)) : (sexts & 0xfu);
        uint32_t re = start + len < m ? (1u << packed_get(w, st + start + len)) : (sexts >> 4);
#pragma unroll
        for (int q = 0; q < NBW; q++) {
            uint32_t b0 = (uint32_t)q * 32;
            uint64_t v = 0;
            if (b0 < len) {
                uint32_t nb = len - b0 < 32 ? len - b0 : 32;
                v = packed_get_kmer(w, st + start + b0, (int)nb).lo << (64 - 2 * nb);
            }
            o[q] = v;
        }
        o[NBW] = (uint64_t)len | ((uint64_t)((re << 4) | le) << 8) | ((uint64_t)d1 << 16);
    });
}

// ------------------------------------------------------------------------------------------------
// per-bin LDS hash-table counter
// ------------------------------------------------------------------------------------------------
struct FastOut {
    uint64_t* hi;       // null when k <= 32
    uint64_t* lo;
    uint32_t* pay;      // exts | min(count, 65535) << 8
    uint32_t* msk;      // CountFilterSet: bit d set iff label d observed (labels < 32); else null
    uint32_t* nobs;     // CountFilterSet: raw observation count (validity uses nobs, filter.rs:99); else null
};

__device__ __forceinline__ uint64_t hash_key(uint64_t hi, uint64_t lo) {
    uint64_t h = lo ^ (hi * 0x9E3779B97F4A7C15ull);
    h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull;
    h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull;
    h ^= h >> 32;
    return h;
}

template <int NT>
__device__ __forceinline__ uint32_t block_inclusive_scan(uint32_t v, uint32_t* s_wsum, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
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

constexpr int IT_DONE = 0, IT_PROBE = 1, IT_VERIFY = 2, IT_HIT = 3;

template <int KW, int NBW, bool IS_SET, int NT, int T, int CH, int ITEMS>
__global__ void __launch_bounds__(NT) bin_count_kernel(const uint64_t* __restrict__ recs, const uint64_t* __restrict__ bin_off,
                                                       int k, int stranded, uint64_t min_obs, FastOut out, uint64_t out_cap,
                                                       unsigned long long* __restrict__ out_cursor, uint32_t* __restrict__ gflags) {
    constexpr int RW = NBW + 1;
    __shared__ uint32_t s_tag[T];
    __shared__ uint64_t s_lo[T];
    __shared__ uint64_t s_hi[KW == 2 ? T : 1];
    __shared__ uint32_t s_cnt[T];
    __shared__ uint32_t s_aux[T];
    __shared__ uint32_t s_msk[IS_SET ? T : 1];
    __shared__ uint64_t s_rec[CH * RW];
    __shared__ uint32_t s_off[CH + 1];
    __shared__ uint32_t s_wsum[NT / 64];
    __shared__ uint32_t s_flag[2];              // [0] table overflow, [1] claimed entries
    __shared__ unsigned long long s_base;

    const uint32_t tid = threadIdx.x;
    const uint64_t r0 = bin_off[blockIdx.x], r1 = bin_off[blockIdx.x + 1];
    if (r0 == r1) return;
    const K128 kmask = k128_mask(k);

    // Work stack of hash-selected passes (P, r): the pass handles the keys with (hash >> 16) % P == r.
    // A pass whose distinct keys overflow the table emits nothing and is replaced by its two children
    // (2P, r) and (2P, r + P), which partition exactly its key set.
    __shared__ uint32_t s_stP[40], s_stR[40];
    __shared__ int s_sp;
    if (tid == 0) { s_stP[0] = 1; s_stR[0] = 0; s_sp = 1; }
    __syncthreads();
    for (uint32_t guard = 0;; guard++) {
        const int sp = s_sp;
        if (sp == 0) break;
        if (guard > 20000u) { if (tid == 0) atomicOr(&gflags[3], 2u); break; }                   // watchdog
        const uint32_t P = s_stP[sp - 1], pr = s_stR[sp - 1];
        __syncthreads();
        if (tid == 0) { s_sp = sp - 1; atomicMax(&gflags[1], P); atomicAdd(&gflags[2], 1u); }
        bool ovf = false;
        {
            for (int i = tid; i < T; i += NT) { s_tag[i] = 0; s_cnt[i] = 0; s_aux[i] = 0; if (IS_SET) s_msk[i] = 0; }
            if (tid == 0) { s_flag[0] = 0; s_flag[1] = 0; }
            __syncthreads();
            for (uint64_t rb = r0; rb < r1 && !ovf; rb += CH) {
                const uint32_t nrec = (uint32_t)((r1 - rb) < (uint64_t)CH ? (r1 - rb) : (uint64_t)CH);
                for (uint32_t w = tid; w < nrec * RW; w += NT) s_rec[w] = recs[rb * RW + w];
                __syncthreads();
                uint32_t nk = 0;
                if (tid < nrec) nk = (uint32_t)(s_rec[tid * RW + NBW] & 0xff) - (uint32_t)k + 1;
                uint32_t tot;
                uint32_t incl = block_inclusive_scan<NT>(nk, s_wsum, &tot);
                if (tid < nrec) s_off[tid + 1] = incl;
                if (tid == 0) s_off[0] = 0;
                __syncthreads();
                for (uint32_t base = 0; base < tot && !ovf; base += NT * ITEMS) {
                    uint64_t it_lo[ITEMS], it_hi[ITEMS];
                    uint32_t it_slot[ITEMS], it_tag[ITEMS], it_aux[ITEMS];
                    int it_state[ITEMS];
                    // ---- materialise up to ITEMS consecutive k-mer instances ----
                    const uint32_t i0 = base + tid * ITEMS;
                    uint32_t rec = 0, j = 0, rlen = 0, rexts = 0, rd = 0;
                    K128 fw{0, 0};
                    const uint64_t* rw = s_rec;
                    if (i0 < tot) {
                        uint32_t lo_i = 0, hi_i = nrec;                 // largest rec with s_off[rec] <= i0
                        while (hi_i - lo_i > 1) { uint32_t mid = (lo_i + hi_i) >> 1; if (s_off[mid] <= i0) lo_i = mid; else hi_i = mid; }
                        rec = lo_i; j = i0 - s_off[rec];
                    }
#pragma unroll
                    for (int it = 0; it < ITEMS; it++) {
                        it_state[it] = IT_DONE;
                        it_lo[it] = it_hi[it] = 0; it_slot[it] = it_tag[it] = it_aux[it] = 0;
                        if (i0 + it >= tot) continue;
                        bool fresh = it == 0;
                        if (it > 0) { j++; if (j + (uint32_t)k > rlen) { rec++; j = 0; fresh = true; } }
                        if (fresh) {
                            rw = s_rec + rec * RW;
                            uint64_t meta = rw[NBW];
                            rlen = (uint32_t)(meta & 0xff); rexts = (uint32_t)(meta >> 8) & 0xffu; rd = (uint32_t)(meta >> 16);
                            fw = packed_get_kmer(rw, j, k);
                        } else {
                            fw = k128_shl(fw, 2);
                            fw.hi &= kmask.hi; fw.lo &= kmask.lo;
                            fw.lo |= packed_get(rw, j + k - 1);
                        }
                        // Exts of k-mer j inside the piece (lib.rs:820-832 with seq_exts = the piece's boundary Exts)
                        uint32_t left = j == 0 ? (rexts & 0xfu) : (1u << packed_get(rw, j - 1));
                        uint32_t right = (j + (uint32_t)k == rlen) ? (rexts & 0xf0u) : (16u << packed_get(rw, j + k));
                        uint32_t ex = left | right;
                        K128 km = fw;
                        if (!stranded) {
                            K128 rc = kmer_rc(fw, k);
                            if (!k128_lt(fw, rc)) { km = rc; ex = exts_rc(ex); }  // ties flip (lib.rs:226-230)
                        }
                        uint64_t h = hash_key(km.hi, km.lo);
                        if (P > 1 && ((uint32_t)(h >> 16) & (P - 1)) != pr) continue;
                        it_lo[it] = km.lo; it_hi[it] = km.hi;
                        it_slot[it] = (uint32_t)h & (T - 1);
                        it_tag[it] = (uint32_t)(h >> 32) | 1u;
                        it_aux[it] = ex | (rd << 8);
                        it_state[it] = IT_PROBE;
                    }
                    // ---- insert: claim-or-match by 32-bit tag, verify full keys after a barrier ----
                    // Every item has a total probe budget of T slots over all rounds; claims stop at
                    // 7/8 occupancy.  Either limit raises the overflow flag and the pass is re-split.
                    uint32_t it_left[ITEMS];
#pragma unroll
                    for (int it = 0; it < ITEMS; it++) it_left[it] = T;
                    for (uint32_t round = 0;; round++) {
                        if (round > 2u * T + 8u) { if (tid == 0) atomicOr(&gflags[3], 1u); break; }   // watchdog: cannot happen
                        bool pending = false;
#pragma unroll
                        for (int it = 0; it < ITEMS; it++) {
                            if (it_state[it] != IT_PROBE) continue;
                            uint32_t slot = it_slot[it];
                            uint32_t left = it_left[it];
                            int st = IT_PROBE;
                            while (left) {
                                left--;
                                uint32_t t = ((volatile uint32_t*)s_tag)[slot];
                                if (t == 0) {
                                    if (((volatile uint32_t*)s_flag)[0]) { left = 0; break; }     // pass already overflowed
                                    t = atomicCAS(&s_tag[slot], 0u, it_tag[it]);
                                    if (t == 0) {
                                        s_lo[slot] = it_lo[it];
                                        if (KW == 2) s_hi[slot] = it_hi[it];
                                        if (atomicAdd(&s_flag[1], 1u) + 1u > (uint32_t)(T - T / 8)) s_flag[0] = 1;
                                        st = IT_HIT;
                                        break;
                                    }
                                }
                                if (t == it_tag[it]) { st = IT_VERIFY; break; }
                                slot = (slot + 1) & (T - 1);
                            }
                            if (st == IT_PROBE) { s_flag[0] = 1; st = IT_DONE; }      // budget exhausted / table full
                            it_slot[it] = slot; it_state[it] = st; it_left[it] = left;
                        }
                        __syncthreads();
                        const uint32_t ovf_now = s_flag[0];          // stable: nobody writes it between the two barriers
#pragma unroll
                        for (int it = 0; it < ITEMS; it++) {
                            if (it_state[it] != IT_VERIFY) continue;
                            uint32_t slot = it_slot[it];
                            bool same = s_lo[slot] == it_lo[it];
                            if (KW == 2) same = same && s_hi[slot] == it_hi[it];
                            if (same) it_state[it] = IT_HIT;
                            else { it_slot[it] = (slot + 1) & (T - 1); it_state[it] = IT_PROBE; pending = true; }
                        }
                        if (!__syncthreads_or((pending && !ovf_now) ? 1 : 0)) break;
                    }
                    ovf = s_flag[0] != 0;
                    if (!ovf) {
#pragma unroll
                        for (int it = 0; it < ITEMS; it++) {
                            if (it_state[it] != IT_HIT) continue;
                            uint32_t slot = it_slot[it];
                            atomicAdd(&s_cnt[slot], 1u);
                            atomicOr(&s_aux[slot], it_aux[it] & 0xffu);
                            if (IS_SET) atomicOr(&s_msk[slot], 1u << ((it_aux[it] >> 8) & 31u));
                        }
                    }
                    __syncthreads();        // flags / table settle before the next batch re-reads them
                }
                __syncthreads();            // s_rec is overwritten by the next chunk
            }
            // ---- emit the valid entries of this pass (a pass that overflowed emits nothing) ----
            if (!ovf) {
            uint32_t nvalid = 0;
            for (int i = tid; i < T; i += NT) {
                if (!s_tag[i]) continue;
                uint32_t c = s_cnt[i];
                bool valid = IS_SET ? (uint64_t)c >= min_obs : (uint64_t)(c > 65535u ? 65535u : c) >= min_obs;
                nvalid += valid ? 1u : 0u;
            }
            uint32_t tot_valid;
            uint32_t incl = block_inclusive_scan<NT>(nvalid, s_wsum, &tot_valid);
            if (tid == 0) s_base = tot_valid ? atomicAdd(out_cursor, (unsigned long long)tot_valid) : 0ull;
            __syncthreads();
            uint64_t o = s_base + (incl - nvalid);
            if (tot_valid && s_base + tot_valid > out_cap) {
                if (tid == 0) atomicOr(&gflags[0], 1u);                    // output buffer too small: host retries
            } else {
                for (int i = tid; i < T; i += NT) {
                    if (!s_tag[i]) continue;
                    uint32_t c = s_cnt[i];
                    uint32_t c16 = c > 65535u ? 65535u : c;
                    bool valid = IS_SET ? (uint64_t)c >= min_obs : (uint64_t)c16 >= min_obs;
                    if (!valid) continue;
                    if (KW == 2) out.hi[o] = s_hi[i];
                    out.lo[o] = s_lo[i];
                    out.pay[o] = (s_aux[i] & 0xffu) | (c16 << 8);
                    if (IS_SET) { out.msk[o] = s_msk[i]; }
                    o++;
                }
            }
            }
            __syncthreads();
        }
        if (ovf) {
            if (P >= 4096u || sp + 1 >= 40) { if (tid == 0) atomicOr(&gflags[0], 2u); break; }
            if (tid == 0) { s_stP[sp - 1] = 2 * P; s_stR[sp - 1] = pr; s_stP[sp] = 2 * P; s_stR[sp] = pr + P; s_sp = sp + 1; }
        }
        __syncthreads();
    }
}

// iota, gather and CSR helpers for the order-restoring stage
__global__ void iota_kernel(uint32_t* p, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}
template <bool IS_SET>
__global__ void finalize_kernel(uint32_t n, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ pay,
                                const uint32_t* __restrict__ msk, uint8_t* __restrict__ exts, uint16_t* __restrict__ count,
                                uint32_t* __restrict__ setn, uint32_t* __restrict__ msk_sorted) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t j = idx[i];
    uint32_t p = pay[j];
    exts[i] = (uint8_t)(p & 0xffu);
    if (IS_SET) { uint32_t m = msk[j]; msk_sorted[i] = m; setn[i] = __popc(m); }
    else count[i] = (uint16_t)(p >> 8);
}
__global__ void set_values_kernel(uint32_t n, const uint32_t* __restrict__ msk_sorted, const uint64_t* __restrict__ set_off,
                                  uint32_t* __restrict__ set_val) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t m = msk_sorted[i];
    uint64_t o = set_off[i];
    while (m) { uint32_t b = __ffs(m) - 1; set_val[o++] = b; m &= m - 1; }   // ascending = sort(); dedup() (filter.rs:97-98)
}
__global__ void max_label_kernel(const void* data, uint32_t width, uint64_t n, uint32_t* out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t v = 0;
    if (i < n) v = width == 1 ? ((const uint8_t*)data)[i] : (width == 2 ? ((const uint16_t*)data)[i] : ((const uint32_t*)data)[i]);
    for (int d = 32; d; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d));
    if ((threadIdx.x & 63) == 0 && v) atomicMax(out, v);
}
}  // namespace

// ------------------------------------------------------------------------------------------------
// host orchestration
// ------------------------------------------------------------------------------------------------
int fast_internal_p(int k) { return k >= 21 ? 13 : std::max(4, k - 8); }

// returns 0 and sets *used = true when the fast path produced the table; *used = false means the
// caller must take the generic path (unsupported shape), nothing was written.
int filter_kmers_fast(dbg_ctx* c, const SeqDev& s, const dbg_filter_params* prm, uint64_t n_kmers, dbg_kmer_table* out,
                      bool* used) {
    *used = false;
    const int k = (int)prm->k;
    const bool is_set = prm->summarizer == DBG_COUNT_FILTER_SET;
    const bool has_hi = k > 32;
    if (k < 16 || prm->report_all_kmers || n_kmers == 0) return 0;
    if (is_set) {
        if (!s.data) return 0;
        DBuf<uint32_t> mx;
        ALLOC_OR_FAIL(c, mx, 1);
        HIP_TRY(c, hipMemsetAsync(mx.p, 0, 4, c->stream));
        max_label_kernel<<<cdiv(s.n, 256), 256, 0, c->stream>>>(s.data, s.data_width, s.n, mx.p);
        LAUNCH_CHECK(c, "max_label");
        uint32_t h = 0;
        HIP_TRY(c, hipMemcpyAsync(&h, mx.p, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (h >= 32) return 0;                          // label domain too wide for the LDS bitmask
    }
    const int p = fast_internal_p(k);
    const int nbw = std::max(2, (2 * k - p + 31) / 32);   // base words per record
    if (nbw < 2 || nbw > 4) return 0;
    constexpr uint32_t TABLE = 2048;
    const uint64_t target = 9000;                        // k-mer instances per bin (about 0.15 distinct per instance)
    uint64_t nb64 = std::max<uint64_t>(1, n_kmers / target);
    if (nb64 > (1u << 24)) nb64 = 1u << 24;
    FastCfg cfg{k, p, prm->stranded != 0, (uint32_t)nb64};
    const uint32_t nbins = cfg.nbins;
    SeqDev sd = s;
    if (!is_set) { sd.data = nullptr; sd.data_width = 0; }   // CountFilter ignores D1 (filter.rs:52-62)

    // ---- pass 1: bin histogram ----
    DBuf<uint32_t> hist, cursor;
    DBuf<uint64_t> bin_off;
    ALLOC_OR_FAIL(c, hist, nbins);
    ALLOC_OR_FAIL(c, cursor, nbins);
    ALLOC_OR_FAIL(c, bin_off, (size_t)nbins + 1);
    HIP_TRY(c, hipMemsetAsync(hist.p, 0, (size_t)nbins * 4, c->stream));
    HIP_TRY(c, hipMemsetAsync(cursor.p, 0, (size_t)nbins * 4, c->stream));
    c->t_begin("sk_count", n_kmers);
    sk_count_kernel<<<cdiv(s.n, 64), 64, 0, c->stream>>>(sd, cfg, hist.p);
    c->t_end();
    LAUNCH_CHECK(c, "sk_count");
    DBG_TRY(scan_exclusive_u32_u64(c, hist.p, bin_off.p, nbins));
    uint64_t n_recs = 0;
    HIP_TRY(c, hipMemcpyAsync(&n_recs, bin_off.p + nbins, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));

    // ---- pass 2: records ----
    const int rw = nbw + 1;
    DBuf<uint64_t> recs;
    ALLOC_OR_FAIL(c, recs, std::max<uint64_t>(n_recs * rw, 1));
    c->t_begin("sk_emit", n_kmers);
    if (nbw == 2) sk_emit_kernel<2><<<cdiv(s.n, 64), 64, 0, c->stream>>>(sd, cfg, bin_off.p, cursor.p, recs.p);
    else if (nbw == 3) sk_emit_kernel<3><<<cdiv(s.n, 64), 64, 0, c->stream>>>(sd, cfg, bin_off.p, cursor.p, recs.p);
    else sk_emit_kernel<4><<<cdiv(s.n, 64), 64, 0, c->stream>>>(sd, cfg, bin_off.p, cursor.p, recs.p);
    c->t_end();
    LAUNCH_CHECK(c, "sk_emit");

    // ---- per-bin LDS hash tables ----
    DBuf<unsigned long long> out_cursor;
    DBuf<uint32_t> gflags;
    ALLOC_OR_FAIL(c, out_cursor, 1);
    ALLOC_OR_FAIL(c, gflags, 4);
    uint64_t cap = std::max<uint64_t>(std::min<uint64_t>(n_kmers, std::max<uint64_t>(n_kmers / 8, 1u << 20)), 1);
    DBuf<uint64_t> u_hi, u_lo;
    DBuf<uint32_t> u_pay, u_msk;
    uint64_t n_out = 0;
    for (int attempt = 0;; attempt++) {
        if (cap >= (1ull << 32)) cap = (1ull << 32) - 1;
        if (has_hi) ALLOC_OR_FAIL(c, u_hi, cap);
        ALLOC_OR_FAIL(c, u_lo, cap);
        ALLOC_OR_FAIL(c, u_pay, cap);
        if (is_set) ALLOC_OR_FAIL(c, u_msk, cap);
        HIP_TRY(c, hipMemsetAsync(out_cursor.p, 0, 8, c->stream));
        HIP_TRY(c, hipMemsetAsync(gflags.p, 0, 16, c->stream));
        FastOut fo{u_hi.p, u_lo.p, u_pay.p, u_msk.p, nullptr};
        c->t_begin("bin_count", n_kmers);
#define GO(KW, NBW, SET) bin_count_kernel<KW, NBW, SET, 512, TABLE, 128, 4><<<nbins, 512, 0, c->stream>>>( \
        recs.p, bin_off.p, k, cfg.stranded, prm->min_kmer_obs, fo, cap, out_cursor.p, gflags.p)
        if (!has_hi) { if (is_set) GO(1, 2, true); else GO(1, 2, false); }
        else if (nbw == 2) { if (is_set) GO(2, 2, true); else GO(2, 2, false); }
        else if (nbw == 3) { if (is_set) GO(2, 3, true); else GO(2, 3, false); }
        else { if (is_set) GO(2, 4, true); else GO(2, 4, false); }
#undef GO
        c->t_end();
        LAUNCH_CHECK(c, "bin_count");
        unsigned long long cur = 0;
        uint32_t flv[4] = {0, 0, 0, 0};
        HIP_TRY(c, hipMemcpyAsync(&cur, out_cursor.p, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(flv, gflags.p, 16, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        const uint32_t fl = flv[0];
        if (getenv("DBG_DEBUG")) fprintf(stderr, "[fastpath] bins=%u recs=%llu valid=%llu flags=%u maxP=%u passes=%u wd=%u\n",
                                         nbins, (unsigned long long)n_recs, cur, fl, flv[1], flv[2], flv[3]);
        if (flv[3]) return c->fail(132, "fast path: internal watchdog fired");
        if (fl & 6u) return c->fail(130, "fast path: a bin exceeded the multi-pass limit");
        if (fl & 1u) {
            if (attempt >= 3 || cur >= (1ull << 32)) return c->fail(131, "fast path: more than 2^32-1 valid k-mers");
            cap = cur + cur / 16 + 1024;
            continue;
        }
        n_out = cur;
        break;
    }
    recs.release();

    // ---- order-restoring sort: ascending key (filter.rs:205-206 bucket order + stable sort = global order) ----
    DBuf<uint32_t> idx, t_pay;
    DBuf<uint64_t> t_hi, t_lo;
    size_t na = std::max<uint64_t>(n_out, 1);
    ALLOC_OR_FAIL(c, idx, na); ALLOC_OR_FAIL(c, t_pay, na); ALLOC_OR_FAIL(c, t_lo, na);
    if (has_hi) ALLOC_OR_FAIL(c, t_hi, na);
    if (n_out) { iota_kernel<<<cdiv(n_out, 256), 256, 0, c->stream>>>(idx.p, (uint32_t)n_out); LAUNCH_CHECK(c, "iota"); }
    RecArrays A{has_hi ? u_hi.p : nullptr, u_lo.p, idx.p}, B{has_hi ? t_hi.p : nullptr, t_lo.p, t_pay.p};
    bool in_b = false;
    DBG_TRY(radix_sort_records(c, n_out, A, B, 2 * k, 0, 0, &in_b));
    RecArrays S = in_b ? B : A;

    DBuf<uint64_t> o_hi, o_lo, o_set_off;
    DBuf<uint8_t> o_exts;
    DBuf<uint16_t> o_count;
    DBuf<uint32_t> o_set_val, setn, msk_sorted;
    ALLOC_OR_FAIL(c, o_hi, na); ALLOC_OR_FAIL(c, o_lo, na); ALLOC_OR_FAIL(c, o_exts, na);
    if (is_set) { ALLOC_OR_FAIL(c, setn, na); ALLOC_OR_FAIL(c, msk_sorted, na); ALLOC_OR_FAIL(c, o_set_off, na + 1); }
    else ALLOC_OR_FAIL(c, o_count, na);
    uint64_t n_setval = 0;
    if (n_out) {
        if (has_hi) HIP_TRY(c, hipMemcpyAsync(o_hi.p, S.hi, n_out * 8, hipMemcpyDeviceToDevice, c->stream));
        else HIP_TRY(c, hipMemsetAsync(o_hi.p, 0, n_out * 8, c->stream));
        HIP_TRY(c, hipMemcpyAsync(o_lo.p, S.lo, n_out * 8, hipMemcpyDeviceToDevice, c->stream));
        c->t_begin("finalize", n_out);
        if (is_set) finalize_kernel<true><<<cdiv(n_out, 256), 256, 0, c->stream>>>((uint32_t)n_out, S.pay, u_pay.p, u_msk.p, o_exts.p, nullptr, setn.p, msk_sorted.p);
        else finalize_kernel<false><<<cdiv(n_out, 256), 256, 0, c->stream>>>((uint32_t)n_out, S.pay, u_pay.p, nullptr, o_exts.p, o_count.p, nullptr, nullptr);
        c->t_end();
        LAUNCH_CHECK(c, "finalize");
    }
    if (is_set) {
        DBG_TRY(scan_exclusive_u32_u64(c, setn.p, o_set_off.p, n_out));
        HIP_TRY(c, hipMemcpyAsync(&n_setval, o_set_off.p + n_out, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        ALLOC_OR_FAIL(c, o_set_val, std::max<uint64_t>(n_setval, 1));
        if (n_out) {
            set_values_kernel<<<cdiv(n_out, 256), 256, 0, c->stream>>>((uint32_t)n_out, msk_sorted.p, o_set_off.p, o_set_val.p);
            LAUNCH_CHECK(c, "set_values");
        }
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    memset(out, 0, sizeof(*out));
    out->n = n_out;
    out->key_hi = o_hi.take(); out->key_lo = o_lo.take(); out->exts = o_exts.take(); out->count = o_count.take();
    out->set_off = o_set_off.take(); out->set_val = o_set_val.take(); out->n_set_val = n_setval;
    out->n_kmer_instances = n_kmers; out->n_passes = 1; out->on_device = 1;
    *used = true;
    return 0;
}
