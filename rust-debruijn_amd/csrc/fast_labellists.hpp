// CountFilterSet over ANY label alphabet on the fast path (round 5; included by fastpath.hip).
//
// The colour masks of the counting kernel hold 64 labels; label groups (fast_manylabels.hpp) stretch that to 1024 at one pass per 64
// labels; everything beyond -- thousands of sample / transcript ids, labels >= 65536 -- used to take the generic path (sort every
// k-mer instance: ~4 Gkmer/s).  CountFilterSet::summarize (filter.rs:85-100) collects the labels of a k-mer's observations, sorts
// and de-duplicates them; here that happens bin by bin, next to the LDS hash table that already groups the bin's k-mers:
//   * the scan's records carry one more word, which holds the read's full D1 label (FastPlan::lists: rw = nbw + 1);
//   * bin_labels_kernel, one workgroup per bin, streams the bin's records TWICE:
//       A  count: k-mer -> table slot, count, Exts (the probing scheme of bin_count_kernel);
//       B  validity (filter.rs:88 compares the unsaturated number of observations) and an exclusive scan of the valid slots'
//          counts: every valid k-mer owns a segment of `count` labels in the bin's stretch of the label buffer (one global atomic
//          per bin reserves the stretch);
//       C  append: every instance of a valid k-mer stores its read's label at the next free place of its segment (LDS cursor);
//       D  every segment is sorted and de-duplicated in place: segments of up to 64 labels -- packed several to a wavefront --
//          by a bitonic network over (segment, label) in registers, longer ones by a wave-level network over the segment itself;
//       E  emit (key, position) records for the order-restoring sort, and {segment offset, labels, Exts} at that position;
//   * after the sort the lists are copied into CSR order (ll_meta_kernel, ll_csr_kernel).
// One pass whatever the alphabet; labels are compared as the u32 values they are (no colour map).  Tables that overflow are
// re-streamed in hash-selected passes like bin_count_kernel's.
#pragma once

struct ListOut {
    uint32_t* lab;                      // label segments
    uint64_t cap;
    unsigned long long* cursor;
};

__device__ __forceinline__ uint64_t ll_range_mask(uint32_t start, uint32_t len) {      // len <= 64, start + len <= 64
    const uint64_t m = len >= 64u ? ~0ull : ((1ull << len) - 1ull);
    return m << start;
}

// Ascending bitonic sort of one 32-bit key per lane.  18 of the 21 compare-exchange steps reach their partner (lane ^ 1, 2, 4, 8)
// with DPP modifiers -- no trip through the LDS crossbar, which the probing waves of the CU's other workgroup keep busy; lane ^ 16
// is a ds_swizzle, lane ^ 32 a ds_bpermute.  (First form: 64-bit keys exchanged with two ds_bpermute per step -- 42 crossbar
// operations per batch of 64 labels; the sort phase took 26 of the kernel's 60 ms per 2*10^9 labels.)
template <int J2>
__device__ __forceinline__ uint32_t ll_lane_xor(uint32_t x) {
    if (J2 == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, true);           // quad_perm:[1,0,3,2]
    if (J2 == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, true);           // quad_perm:[2,3,0,1]
    if (J2 == 4) {                                                                                       // banks 0, 2 read lane + 4, banks 1, 3 lane - 4
        const int t = __builtin_amdgcn_update_dpp((int)x, (int)x, 0x104, 0xf, 0x5, false);               // row_shl:4
        return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)x, 0x114, 0xf, 0xa, false);                 // row_shr:4
    }
    if (J2 == 8) return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x128, 0xf, 0xf, false);    // row_ror:8
    if (J2 == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, 0x401F);                           // and 0x1f, or 0, xor 0x10
    return (uint32_t)__shfl_xor((int)x, 32);
}
// lanes that keep the smaller key in the (K2, J2) step: a compile-time constant, handed to v_cndmask_b32 as a literal SGPR pair (as
// values computed from the lane number the 21 masks were kept in SGPRs, spilled to VGPR lanes, and fetched back with two
// v_readlane per step -- 42 VALU instructions per group of 64 labels)
constexpr uint64_t ll_min_lanes(uint32_t k2, uint32_t j2) {
    uint64_t m = 0;
    for (uint32_t l = 0; l < 64; l++) if (((l & k2) == 0u) == ((l & j2) == 0u)) m |= 1ull << l;
    return m;
}
template <int K2, int J2>
__device__ __forceinline__ uint32_t ll_cmpx(uint32_t key, uint32_t lane) {
    const uint32_t other = ll_lane_xor<J2>(key);
    const uint32_t mn = key < other ? key : other, mx = key < other ? other : key;
    constexpr uint64_t M = ll_min_lanes((uint32_t)K2, (uint32_t)J2);
    uint32_t r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mx), "v"(mn), "s"(M));
    return r;
}
// The network, for four independent keys per lane, step by step (a step's three instructions depend on one another, and a DPP operand wants its producer
// several cycles old): -10 % of the sort phase's time, -1 % of the kernel's -- the CU as a whole is bound by VALU issue, what one
// phase leaves idle the other workgroup's phases take
#define LL_STEP2(K2, J2) do { a = ll_cmpx<K2, J2>(a, lane); b = ll_cmpx<K2, J2>(b, lane); c = ll_cmpx<K2, J2>(c, lane); d = ll_cmpx<K2, J2>(d, lane); } while (0)
__device__ __forceinline__ void ll_wave_sort4(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d, uint32_t lane) {
    LL_STEP2(2, 1);
    LL_STEP2(4, 2); LL_STEP2(4, 1);
    LL_STEP2(8, 4); LL_STEP2(8, 2); LL_STEP2(8, 1);
    LL_STEP2(16, 8); LL_STEP2(16, 4); LL_STEP2(16, 2); LL_STEP2(16, 1);
    LL_STEP2(32, 16); LL_STEP2(32, 8); LL_STEP2(32, 4); LL_STEP2(32, 2); LL_STEP2(32, 1);
    LL_STEP2(64, 32); LL_STEP2(64, 16); LL_STEP2(64, 8); LL_STEP2(64, 4); LL_STEP2(64, 2); LL_STEP2(64, 1);
}
#undef LL_STEP2
// inclusive maximum over the lanes at or below this one (the DPP ladder of wave_inclusive_scan_u32 with max for +)
__device__ __forceinline__ uint32_t ll_wave_max_scan(uint32_t v) {
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    const uint32_t a = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    const uint32_t b = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    const uint32_t c = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, false);
    v = mx(mx(v, a), mx(b, c));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xe, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xc, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return v;
}

// A segment of more than 64 labels, sorted and de-duplicated in place by one wavefront: bitonic network in the all-ascending form
// (the first step of every merge mirrors, the others shift), so that positions >= n simply count as +infinity and are never
// touched.  The labels live in global memory (a k-mer seen thousands of times); steps are separated by WORKGROUP-scope fences: the
// waves of a workgroup share their CU's write-through L1, so a wait for the outstanding stores is all the ordering needs -- a
// device-scope fence writes the XCD's L2 back and invalidates it (__threadfence() after the append sweep made that sweep 2.4x
// slower and the stores another 50 ms per 2*10^9 labels).  Returns the number of distinct labels.
__device__ __forceinline__ uint32_t ll_wave_sort_unique_global(uint32_t* __restrict__ a, uint32_t n, uint32_t lane) {
    uint32_t P = 128;
    while (P < n) P <<= 1;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    for (uint32_t k2 = 2; k2 <= P; k2 <<= 1) {
        for (uint32_t j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
            const bool mirror = j2 == (k2 >> 1);
            for (uint32_t t = lane; t < (P >> 1); t += 64) {
                const uint32_t i = ((t / j2) * 2u * j2) + (t % j2);
                const uint32_t q = mirror ? (i ^ (k2 - 1u)) : (i + j2);
                if (q < n) {                                                     // (i < q always)
                    const uint32_t x = a[i], y = a[q];
                    if (y < x) { a[i] = y; a[q] = x; }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
    }
    uint32_t out_n = 0, carry = 0;
    const uint64_t lt = lanemask_lt();
    for (uint32_t c0 = 0; c0 < n; c0 += 64) {
        const bool act = c0 + lane < n;
        const uint32_t v = act ? a[c0 + lane] : 0u;
        uint32_t pv = (uint32_t)__shfl_up((int)v, 1);
        if (lane == 0) pv = carry;
        const bool first = act && (c0 + lane == 0u || v != pv);
        const uint64_t fm = __ballot(first);
        carry = (uint32_t)__shfl((int)v, 63);
        if (first) a[out_n + (uint32_t)__popcll(fm & lt)] = v;                   // at or below its own position: never ahead of a read
        out_n += (uint32_t)__popcll(fm);
    }
    return out_n;
}

// DBG_LL_ABL = n (measurement builds, WRONG results): 1 stop after the counting sweep, 2 after the segment offsets, 3 no sort of the
// segments, 4 the append sweep without its global stores (and no sort), 5 one label per merged record only (and no sort), 6 the sort phase without its networks
#ifndef DBG_LL_ABL
#define DBG_LL_ABL 0
#endif
// -DDBG_LL_PHASE_TIMES (measurement builds): 100 MHz ticks of thread 0 per phase, summed over bins; printed by the host under DBG_DEBUG
#ifdef DBG_LL_PHASE_TIMES
__device__ unsigned long long g_ll_phase[8];
#define LLPH(i) do { if (tid == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_ll_phase[i], now_ - t_prev_); t_prev_ = now_; } } while (0)
#else
#define LLPH(i) do {} while (0)
#endif
template <int KW, int NBW, int NT, int T>
__device__ __forceinline__ void bin_labels_body(const uint32_t bin_idx, const uint64_t* __restrict__ recs, const uint64_t* __restrict__ recs_alt, uint32_t alt_from,
                                                       const uint64_t* __restrict__ seg_beg, const uint64_t* __restrict__ seg_end,
                                                       uint32_t n_src, uint64_t seg_stride,
                                                       int k, int stranded, uint64_t min_obs, FastOut out, uint64_t out_cap,
                                                       unsigned long long* __restrict__ out_cursor, ListOut lo, uint32_t* __restrict__ gflags) {
    constexpr int RW = NBW;                                 // words of a record's bases + meta bits (staged in LDS)
    constexpr int RWG = NBW + 1;                            // words of a record in memory: + the label word
    constexpr int NWV = NT / 64;
    constexpr uint32_t CH = 4;
    constexpr uint32_t CAPC = (NBW == 4 ? 3 : 4) * NT;       // chunk-map capacity per round (k >= 56: keeps two workgroups per CU)
    __shared__ __attribute__((aligned(16))) uint32_t s_tag[T];
    __shared__ __attribute__((aligned(16))) uint64_t s_key[KW * T];
    __shared__ uint32_t s_cnt[T];                           // observations; after phase D: distinct labels (0 = not valid)
    __shared__ uint32_t s_cur[T];                           // phase C: next free place of the slot's segment; after D: the segment's offset
    __shared__ uint32_t s_ex4[T / 4];                       // Exts, one byte per entry
    __shared__ uint64_t s_slab[RW * NT];                    // staged batch of records, word-major
    __shared__ uint32_t s_lab[NT];                          // labels of the batch's records, grouped by staged record (append sweep)
    __shared__ __attribute__((aligned(4))) uint16_t s_cmap[CAPC];   // chunk -> record slot | chunk index << 10
    // Reads that cover the same stretch of the genome produce identical records (fastpath.hip): a batch's identical records are
    // merged into one staged record that carries their number -- its k-mers are rolled and probed once -- and, in the append
    // sweep, the list of their labels.  The filter (bin_count_kernel's claim / barrier / compare scheme) borrows s_cmap, which is
    // only needed once the staged records are cut into chunks.
    constexpr uint32_t DD = 2 * NT;
    static_assert(CAPC * 2 >= DD * 4 || NBW == 4, "the duplicate filter borrows s_cmap");
    __shared__ uint32_t s_dd_own[NBW == 4 ? DD : 1];        // (k >= 56: the chunk map is smaller than the filter)
    uint32_t* const s_dd = NBW == 4 ? s_dd_own : reinterpret_cast<uint32_t*>(s_cmap);
    __shared__ uint32_t s_w[NT / 2];                        // copies per staged record (u16 halves)
    __shared__ uint32_t s_o[NT / 2];                        // append sweep: end of the staged record's label list in s_lab (u16 halves)
    __shared__ uint32_t s_m, s_cproc, s_nextq, s_nst;
    __shared__ uint32_t s_wsum[NWV];
    __shared__ uint32_t s_flag[2];                          // [0] table overflow, [1] claimed entries
    __shared__ unsigned long long s_base, s_base_all, s_lbase;
    __shared__ uint32_t s_st[24];
    __shared__ uint32_t s_segpre[66];
    __shared__ uint64_t s_segbeg[64];

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef DBG_LL_PHASE_TIMES
    unsigned long long t_prev_ = wall_clock64();
#endif
    if (tid < 64) {
        uint32_t len = 0;
        if (tid < n_src) {
            const uint64_t a = seg_beg[tid * seg_stride + (uint64_t)bin_idx], b = seg_end[tid * seg_stride + (uint64_t)bin_idx];
            len = (uint32_t)(b - a);
            s_segbeg[tid] = a;
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
    LLPH(0);
    if (total_recs == 0) return;
    const K128 kmask = k128_mask(k);
    const uint64_t lt_mask = lanemask_lt();

    int sp = 1;
    for (uint32_t guard = 0;; guard++) {
        if (sp == 0) break;
        if (guard > 20000u) { if (tid == 0) atomicOr(&gflags[3], 2u); break; }
        const uint32_t ent = guard == 0 ? 1u : s_st[sp - 1];
        const uint32_t P = ent & 0xffffu, pr = ent >> 16;
        sp--;
        if (tid == 0 && P > 1) { atomicMax(&gflags[1], P); atomicAdd(&gflags[2], 1u); }
        for (int i = tid; i < T; i += NT) { s_tag[i] = 0; s_cnt[i] = 0; s_cur[i] = 0; }
        for (int i = tid; i < T / 4; i += NT) s_ex4[i] = 0;
        if (tid == 0) { s_flag[0] = 0; s_flag[1] = 0; }
        __syncthreads();

        unsigned long long lbase = 0;                       // start of this pass's stretch of the label buffer (phase C)
        // ---- one sweep over the bin's records.  MODE 0: insert + count; MODE 1: look up + append the label ----
        auto stream = [&](const int MODE) -> bool {
            for (uint32_t r0 = 0; r0 < total_recs; r0 += NT) {
                const uint32_t nb = total_recs - r0 < (uint32_t)NT ? total_recs - r0 : (uint32_t)NT;
                constexpr uint64_t COLOUR_BITS = 63ull << 15;   // (the 6-bit colour of the record: meaningless here, differs between copies)
                uint64_t a[RW];
#pragma unroll
                for (int q = 0; q < RW; q++) a[q] = 0;
                uint32_t label = 0;
                bool pend = false;
                if (tid < nb) {
                    const uint32_t ridx = r0 + tid;
                    uint32_t sg = 0;
                    while (sg + 1 < n_src && ridx >= s_segpre[sg + 1]) sg++;
                    const uint64_t ri = s_segbeg[sg] + (ridx - s_segpre[sg]);
                    const uint64_t* g = (sg >= alt_from ? recs_alt : recs) + ri * RWG;
#pragma unroll
                    for (int q = 0; q < RW; q++) a[q] = g[q];
                    if (MODE == 1) label = (uint32_t)g[RW];
                    a[RW - 1] &= ~COLOUR_BITS;
                    const uint32_t rl = (uint32_t)(a[RW - 1] & 0x7f);
                    if (rl < (uint32_t)k || rl > (uint32_t)(32 * NBW - (META_BITS + 1) / 2)) atomicOr(&gflags[3], 4u);   // not a record of the scan
                    else pend = true;
                }
                if (tid < NT / 2) { s_w[tid] = 0; s_o[tid] = 0; }
                for (uint32_t i = tid; i < DD; i += NT) s_dd[i] = 0;
                if (tid == 0) s_nst = 0;
                __syncthreads();
                uint32_t myslot = 0;
                const bool have = pend;
                {
                    uint64_t ha = a[0], hb = a[1];
                    if constexpr (NBW == 3) ha += a[2] * 0x9E3779B97F4A7C15ull;
                    if constexpr (NBW == 4) { ha += a[2] * 0x9E3779B97F4A7C15ull; hb += a[3] * 0xC2B2AE3D27D4EB4Full; }
                    const uint64_t h = hash_key(ha, hb);
                    const uint32_t mytag = (uint32_t)(h >> 42) << 10;
                    uint32_t sl = (uint32_t)h & (DD - 1);
                    for (;;) {
                        if (pend) {
                            constexpr uint32_t CLAIMED = 1023u;
                            for (;;) {                                                   // to the first slot that is free or carries my hash bits
                                uint32_t v = s_dd[sl];
                                if (v == 0u) {
                                    v = atomicCAS(&s_dd[sl], 0u, CLAIMED | mytag);
                                    if (v == 0u) {                                       // first of its kind: stage it
                                        const uint32_t mine = atomicAdd(&s_nst, 1u);
#pragma unroll
                                        for (int q = 0; q < RW; q++) s_slab[q * NT + mine] = a[q];
                                        atomicAdd(&s_w[mine >> 1], 1u << (16 * (mine & 1u)));
                                        s_dd[sl] = (mine + 1u) | mytag;
                                        myslot = mine;
                                        pend = false;
                                        break;
                                    }
                                }
                                if ((v & ~1023u) == mytag) break;                        // (being) staged here: compared after the barrier
                                sl = (sl + 1u) & (DD - 1);
                            }
                        }
                        lds_barrier();
                        if (pend) {
                            const uint32_t r = (s_dd[sl] & 1023u) - 1u;
                            bool same = true;
#pragma unroll
                            for (int q = 0; q < RW; q++) same = same && s_slab[q * NT + r] == a[q];
                            if (same) {
                                atomicAdd(&s_w[r >> 1], 1u << (16 * (r & 1u)));
                                myslot = r;
                                pend = false;
                            } else sl = (sl + 1u) & (DD - 1);
                        }
                        if (!__syncthreads_or(pend ? 1 : 0)) break;
                    }
                }
                const uint32_t nstaged = s_nst;
                if (MODE == 1) {
                    // the labels of a staged record's copies, contiguous in s_lab: exclusive scan of the copies, then every incoming
                    // record takes the next place of its staged record's list (s_o ends up at the list's END)
                    const uint32_t wq = tid < nstaged ? (s_w[tid >> 1] >> (16 * (tid & 1u))) & 0xffffu : 0u;
                    uint32_t tot;
                    const uint32_t incl = block_inclusive_scan<NT>(wq, s_wsum, &tot);
                    if (tid < nstaged) atomicAdd(&s_o[tid >> 1], (incl - wq) << (16 * (tid & 1u)));
                    __syncthreads();
                    if (have) {
                        const uint32_t old = atomicAdd(&s_o[myslot >> 1], 1u << (16 * (myslot & 1u)));
                        s_lab[(old >> (16 * (myslot & 1u))) & 0xffffu] = label;
                    }
                }
                uint32_t nkr = 0;
                if (tid < nstaged) nkr = (uint32_t)(s_slab[(RW - 1) * NT + tid] & 0x7f) - (uint32_t)k + 1u;
                const uint32_t nch = (nkr + CH - 1) / CH;
                for (uint32_t base = 0; base < nstaged;) {
                    if (tid == 0) { s_m = nstaged; s_cproc = 0; s_nextq = 0; }
                    uint32_t tot;
                    const uint32_t mine = tid >= base ? nch : 0u;
                    const uint32_t incl = block_inclusive_scan<NT>(mine, s_wsum, &tot);      // barriers inside (also: the batch is staged)
                    if (incl <= CAPC) {
                        if (mine) atomicMax(&s_cproc, incl);
                        for (uint32_t c = 0; c < mine; c++) s_cmap[incl - mine + c] = (uint16_t)(tid | (c << 10));
                    } else if (tid >= base && tid < nstaged) atomicMin(&s_m, tid);
                    __syncthreads();
                    const uint32_t mend = s_m, cproc = s_cproc;
                    for (;;) {
                        uint32_t q0 = 0;
                        if (lane == 0) q0 = atomicAdd(&s_nextq, 64u);
                        q0 = __shfl(q0, 0);
                        if (q0 >= cproc) break;
                        const uint32_t q = q0 + lane;
                        const bool act = q < cproc;
                        const uint32_t e = act ? (uint32_t)s_cmap[q] : 0u;
                        const uint32_t r = e & 1023u, c = e >> 10;
                        uint64_t W0 = s_slab[r], W1 = s_slab[NT + r], W2 = 0, W3 = 0;
                        if (NBW > 2) W2 = s_slab[2 * NT + r];
                        if (NBW > 3) W3 = s_slab[3 * NT + r];
                        uint64_t meta;
                        {
                            uint64_t& WL = NBW == 2 ? W1 : (NBW == 3 ? W2 : W3);
                            meta = WL & ((1ull << META_BITS) - 1);
                            WL &= ~((1ull << META_BITS) - 1);
                        }
                        const uint32_t wgt = (s_w[r >> 1] >> (16 * (r & 1u))) & 0xffffu;
                        const uint32_t lstart = MODE == 1 ? ((s_o[r >> 1] >> (16 * (r & 1u))) & 0xffffu) - wgt : 0u;
                        const uint32_t rlen = (uint32_t)(meta & 0x7f), rexts = (uint32_t)(meta >> 7) & 0xffu;
                        const uint32_t rnk = rlen - (uint32_t)k + 1u, rnch = (rnk + 3u) >> 2;
                        const uint32_t cbase = rnk < 4u ? rnk : (rnk == 5u ? 2u : ((rnk & 3u) ? 3u : 4u)), crem = rnk - cbase * rnch;
                        uint32_t j = c * cbase + (c < crem ? c : crem);
                        const uint32_t jend = act ? j + cbase + (c < crem ? 1u : 0u) : j;
                        auto base_at = [&](uint32_t qq) -> uint32_t {
                            const uint64_t wd = qq < 32 ? W0 : (qq < 64 ? W1 : (NBW > 2 && qq < 96 ? W2 : (NBW > 3 ? W3 : (NBW > 2 ? W2 : W1))));
                            return (uint32_t)(wd >> (62 - 2 * (qq & 31))) & 3u;
                        };
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
                        uint32_t nx;
                        {
                            const uint32_t sft = 2 * (j + (uint32_t)k), ws = sft >> 6, bs = sft & 63;
                            const uint64_t A = ws == 0 ? W0 : (ws == 1 ? W1 : (NBW > 2 && ws == 2 ? W2 : (NBW > 3 && ws == 3 ? W3 : 0ull)));
                            const uint64_t B = ws == 0 ? W1 : (NBW > 2 && ws == 1 ? W2 : (NBW > 3 && ws == 2 ? W3 : 0ull));
                            const uint64_t v = bs ? (A << bs) | (B >> (64 - bs)) : A;
                            nx = (uint32_t)(v >> 32);
                        }
                        while (__any(j < jend)) {
                            if (j < jend) {
                                const uint32_t nbase = nx >> 30;
                                nx <<= 2;
                                uint32_t left = j == 0 ? (rexts & 0xfu) : (1u << lb);
                                uint32_t right = (j + (uint32_t)k == rlen) ? (rexts & 0xf0u) : (16u << nbase);
                                uint32_t ex = left | right;
                                K128 km = fw;
                                if (!stranded && !k128_lt(fw, rcw)) { km = rcw; ex = __brev(ex) >> 24; }   // ties flip (lib.rs:226-230)
                                const uint64_t h = hash_key(km.hi, km.lo);
                                if (P == 1 || ((uint32_t)(h >> 16) & (P - 1)) == pr) {
                                    const uint32_t mytag = ((uint32_t)(h >> 32) & 0x7fffffffu) | 1u;
                                    uint32_t bkt = (uint32_t)h & (T / 4 - 1);
                                    uint32_t slot = 0, tried = 0, nprobe = 0;
                                    const uint32_t rot = (uint32_t)(h >> 12) & 3u;
                                    bool hit = false;
                                    for (;;) {
                                        asm volatile("" ::: "memory");
                                        const uint4 t4 = *reinterpret_cast<const uint4*>(&s_tag[bkt * 4]);
                                        const uint32_t mm = ((t4.x == mytag ? 1u : 0u) | (t4.y == mytag ? 2u : 0u) | (t4.z == mytag ? 4u : 0u) |
                                                             (t4.w == mytag ? 8u : 0u)) & ~tried;
                                        const uint32_t em = (t4.x == 0u ? 1u : 0u) | (t4.y == 0u ? 2u : 0u) | (t4.z == 0u ? 4u : 0u) | (t4.w == 0u ? 8u : 0u);
                                        if (mm) {
                                            const uint32_t i = (uint32_t)__ffs((int)mm) - 1u, sl = bkt * 4 + i;
                                            bool same;
                                            if (KW == 2) {
                                                const ulonglong2 kk = *reinterpret_cast<const ulonglong2*>(&s_key[2 * sl]);
                                                same = kk.x == km.lo && kk.y == km.hi;
                                            } else same = s_key[sl] == km.lo;
                                            if (same) { hit = true; slot = sl; break; }
                                            tried |= 1u << i;
                                            continue;
                                        }
                                        if (MODE == 0) {
                                            const uint32_t bz = mytag | TAG_BUSY;
                                            if (t4.x == bz || t4.y == bz || t4.z == bz || t4.w == bz) continue;      // a claimer is writing its key
                                            if (em) {
                                                const uint32_t emr = ((em >> rot) | (em << (4u - rot))) & 15u;
                                                const uint32_t sl = bkt * 4 + (((uint32_t)__ffs((int)emr) - 1u + rot) & 3u);
                                                if (atomicCAS(&s_tag[sl], 0u, bz) == 0u) {
                                                    if (KW == 2) *reinterpret_cast<ulonglong2*>(&s_key[2 * sl]) = make_ulonglong2(km.lo, km.hi);
                                                    else s_key[sl] = km.lo;
                                                    asm volatile("" ::: "memory");
                                                    __hip_atomic_store(&s_tag[sl], mytag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                                    atomicAdd(&s_flag[1], 1u);
                                                    hit = true; slot = sl;
                                                    break;
                                                }
                                                continue;
                                            }
                                        } else if (em) break;                    // a key of the table sits before the first free slot of its chain
                                        bkt = (bkt + 1) & (T / 4 - 1);
                                        tried = 0;
                                        if (++nprobe >= (uint32_t)(T / 4)) break;
                                    }
                                    if (MODE == 0) {
                                        if (hit) {
                                            atomicAdd(&s_cnt[slot], wgt);
                                            atomicOr(&s_ex4[slot >> 2], ex << (8u * (slot & 3u)));
                                        } else __hip_atomic_store(&s_flag[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    } else if (hit && (uint64_t)s_cnt[slot] >= min_obs) {
                                        const uint32_t pos = atomicAdd(&s_cur[slot], wgt);
                                        for (uint32_t t = 0; t < (DBG_LL_ABL == 5 ? 1u : wgt); t++) {
                                            const uint32_t label = s_lab[lstart + t];
                                            if (DBG_LL_ABL != 4 || label == 0x12345u) lo.lab[lbase + pos + t] = label;
                                        }
                                    }
                                }
                                if (KW == 2) {
                                    const int sh = 2 * (k - 1) - 64;
                                    lb = (uint32_t)(fw.hi >> sh) & 3u;
                                    fw.hi = ((fw.hi << 2) | (fw.lo >> 62)) & kmask.hi;
                                    fw.lo = (fw.lo << 2) | nbase;
                                    rcw.lo = (rcw.lo >> 2) | (rcw.hi << 62);
                                    rcw.hi = (rcw.hi >> 2) | ((uint64_t)(3u - nbase) << sh);
                                } else {
                                    const int sh = 2 * (k - 1);
                                    lb = (uint32_t)(fw.lo >> sh) & 3u;
                                    fw.lo = ((fw.lo << 2) | nbase) & kmask.lo;
                                    rcw.lo = (rcw.lo >> 2) | ((uint64_t)(3u - nbase) << sh);
                                }
                                j++;
                            }
                        }
                    }
                    __syncthreads();
                    base = mend;
                }
                if (MODE == 0) {
                    // a table more than 7/8 full with records still to come: give up early and re-split the pass
                    if (r0 + NT < total_recs && tid == 0 && s_flag[1] > (uint32_t)(T - T / 8)) s_flag[0] = 1;
                    __syncthreads();
                    if (__hip_atomic_load(&s_flag[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) return true;
                }
            }
            return false;
        };

        // ---- A: count ----
        const bool ovf = stream(0);
        LLPH(1);
        if (DBG_LL_ABL == 1) break;
        bool room = true;
        if (!ovf) {
            // ---- B: segments of the valid k-mers (four consecutive slots per thread) ----
            constexpr int SPT = T / NT;                     // consecutive slots per thread
            static_assert(T == SPT * NT && SPT >= 1, "whole slots per thread");
            uint32_t v4[SPT], sum = 0;
#pragma unroll
            for (int q = 0; q < SPT; q++) {
                const uint32_t i = tid * SPT + q;
                const uint32_t c = s_cnt[i];
                v4[q] = (s_tag[i] != 0u && (uint64_t)c >= min_obs) ? c : 0u;
                sum += v4[q];
            }
            uint32_t tot;
            const uint32_t incl = block_inclusive_scan<NT>(sum, s_wsum, &tot);
            uint32_t o = incl - sum;
#pragma unroll
            for (int q = 0; q < SPT; q++) { s_cur[tid * SPT + q] = o; o += v4[q]; }
            if (tid == 0) s_lbase = tot ? atomicAdd(lo.cursor, (unsigned long long)tot) : 0ull;
            __syncthreads();
            lbase = s_lbase;
            room = lbase + tot <= lo.cap;
            if (!room && tid == 0) atomicOr(&gflags[0], 16u);
            LLPH(2);
            if (DBG_LL_ABL == 2) break;
            if (room && tot) {
                // ---- C: append ----
                (void)stream(1);
                __syncthreads();                            // (workgroup scope: the appended labels are visible to the waves that sort them)
                LLPH(3);
                // ---- D: sort + de-duplicate every segment; each wave takes its own T / NWV slots ----
                uint32_t* const L = lo.lab + lbase;
                for (uint32_t blk = 0; blk < (uint32_t)(T / NWV / 64) && DBG_LL_ABL < 3; blk++) {
                    const uint32_t i = wave * (T / NWV) + blk * 64 + lane;
                    const uint32_t c = s_cnt[i];
                    const bool valid = s_tag[i] != 0u && (uint64_t)c >= min_obs;
                    const uint32_t off = valid ? s_cur[i] - c : 0u;
                    const uint32_t v = (valid && c <= 64u) ? c : 0u;
                    const uint64_t bigm = __ballot(valid && c > 64u);
                    const uint32_t incl = wave_inclusive_scan_u32(v);
                    uint32_t nl = valid ? c : 0u;
                    // per-wave scratch in the (idle) record staging area: element -> slot lane marks, and the slot lanes' numbers
                    uint32_t* const WS = reinterpret_cast<uint32_t*>(s_slab) + wave * 256;
                    WS[64 + lane] = incl - v; WS[128 + lane] = off; WS[192 + lane] = v;
                    // Groups of slot lanes whose segments fill at most 64 lanes.  Four groups are set up -- their labels requested --
                    // before the first is sorted: with load -> sort -> store per group the kernel's 16 waves per CU cannot hide the
                    // load's latency (the sort is ~120 instructions).  The set-up is unconditional (a group past the end is empty),
                    // so that the four requests stay in flight while the first groups are sorted: a request behind a branch makes
                    // the compiler wait for all of them.
                    struct Grp { uint32_t g0, g1, sbase, n_el, j, excl_j, off_j, v_j, label; };
                    auto prepare = [&](uint32_t g0, uint32_t sbase) -> Grp {
                        Grp G;
                        G.g0 = g0; G.sbase = sbase;
                        const uint64_t fit = __ballot(lane >= g0 && incl - sbase <= 64u);    // incl is monotone: the lanes [g0, g1)
                        G.g1 = g0 + (uint32_t)__popcll(fit);
                        G.n_el = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)(G.g1 - 1u)) - sbase;
                        // element e of the batch belongs to the last slot lane whose segment starts at or before e: the slot
                        // lanes mark their first element, a maximum scan spreads the marks
                        WS[lane] = 0u;
                        if (lane >= g0 && lane < G.g1 && v) WS[incl - v - sbase] = lane + 1u;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        const bool act = lane < G.n_el;
                        const uint32_t ms = ll_wave_max_scan(WS[lane]);
                        G.j = act ? ms - 1u : g0;
                        G.excl_j = WS[64 + G.j] - sbase; G.off_j = WS[128 + G.j]; G.v_j = WS[192 + G.j];
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        G.label = act ? L[G.off_j + (lane - G.excl_j)] : 0u;
                        return G;
                    };
                    // (two groups at a time, no branches: an empty group has no active lane)
                    auto finish = [&](const Grp& G, uint32_t key) {
                        const bool act = lane < G.n_el;
                        // the segments keep their lane ranges: lane e still belongs to slot lane j after the sort
                        const uint32_t pk = (uint32_t)__shfl_up((int)key, 1);
                        const bool first = act && (lane == 0u || pk != key);
                        const uint64_t fm = __ballot(first);
                        if (first) L[G.off_j + (uint32_t)__popcll(fm & ll_range_mask(G.excl_j, G.v_j) & lt_mask)] = key & 0xffffffu;
                        if (lane >= G.g0 && lane < G.g1 && v) nl = (uint32_t)__popcll(fm & ll_range_mask(incl - v - G.sbase, v));
                    };
                    auto process4 = [&](const Grp& GA, const Grp& GB, const Grp& GC, const Grp& GD) {
                        // (segment in the batch, label): labels are < 2^24 (checked by the host side), at most 64 segments
                        uint32_t ka = lane < GA.n_el ? (((GA.j - GA.g0) << 24) | GA.label) : 0xffffffffu;
                        uint32_t kb = lane < GB.n_el ? (((GB.j - GB.g0) << 24) | GB.label) : 0xffffffffu;
                        uint32_t kc = lane < GC.n_el ? (((GC.j - GC.g0) << 24) | GC.label) : 0xffffffffu;
                        uint32_t kd = lane < GD.n_el ? (((GD.j - GD.g0) << 24) | GD.label) : 0xffffffffu;
                        if (DBG_LL_ABL != 6) ll_wave_sort4(ka, kb, kc, kd, lane);
                        finish(GA, ka); finish(GB, kb); finish(GC, kc); finish(GD, kd);
                    };
                    uint32_t g0 = 0, sbase = 0;
                    while (g0 < 64u) {
                        Grp G[4];
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            G[q] = prepare(g0, sbase);
                            g0 = G[q].g1;
                            sbase = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)(g0 - 1u));
                        }
                        process4(G[0], G[1], G[2], G[3]);
                    }
                    uint64_t bm = bigm;
                    while (bm) {
                        const uint32_t jj = (uint32_t)__ffsll((long long)bm) - 1u;
                        bm &= bm - 1;
                        const uint32_t c_j = (uint32_t)__shfl((int)c, (int)jj), off_j = (uint32_t)__shfl((int)off, (int)jj);
                        const uint32_t n = ll_wave_sort_unique_global(L + off_j, c_j, lane);
                        if (lane == jj) nl = n;
                    }
                    s_cnt[i] = nl;
                    s_cur[i] = off;
                }
            } else if (room) {
                // no valid k-mer in this pass: nothing to append; counts of invalid slots must read as "no labels"
                for (int i = tid; i < T; i += NT) s_cnt[i] = 0;
            }
            __syncthreads();
            LLPH(4);
        }
        if (!ovf && room) {
            // ---- E: emit (bin_count_kernel's scheme; valid = has labels) ----
            constexpr int EMIT_IT = T / NWV / 64;
            const uint32_t slot0 = wave * (T / NWV) + lane;
            uint64_t vb[EMIT_IT], ab[EMIT_IT];
            uint32_t nv = 0, na = 0;
#pragma unroll
            for (int it = 0; it < EMIT_IT; it++) {
                const uint32_t i = slot0 + it * 64;
                const bool occ = s_tag[i] != 0u;
                vb[it] = __ballot(occ && s_cnt[i] != 0u); ab[it] = __ballot(occ);
                nv += (uint32_t)__popcll(vb[it]); na += (uint32_t)__popcll(ab[it]);
            }
            if (lane == 0) s_wsum[wave] = nv | (na << 16);
            __syncthreads();
            uint32_t before = 0, total = 0;
#pragma unroll
            for (int w = 0; w < NWV; w++) { const uint32_t x = s_wsum[w]; before += (uint32_t)w < wave ? x : 0u; total += x; }
            const uint32_t tot_valid = total & 0xffffu, tot_all = total >> 16;
            if (tid == 0) {
                s_base = tot_valid ? atomicAdd(out_cursor, (unsigned long long)tot_valid) : 0ull;
                s_base_all = (out.all_lo && tot_all) ? atomicAdd(out.all_cursor, (unsigned long long)tot_all) : 0ull;
            }
            __syncthreads();
            const unsigned long long base = s_base + (before & 0xffffu), base_all = s_base_all + (before >> 16);
            const bool fit = !(tot_valid && s_base + tot_valid > out_cap);
            const bool fit_all = !(out.all_lo && tot_all && s_base_all + tot_all > out.all_cap);
            if (!fit && tid == 0) atomicOr(&gflags[0], 1u);
            if (!fit_all && tid == 0) atomicOr(&gflags[0], 8u);
            if (fit && fit_all) {
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
                        const unsigned long long goff = lbase + s_cur[i];
                        out.w_rec[q] = make_uint4((uint32_t)goff, (uint32_t)(goff >> 32), s_cnt[i], (s_ex4[i >> 2] >> (8u * (i & 3u))) & 0xffu);
                        const uint32_t pay = (uint32_t)q;
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
        } else if (ovf) {
            if (P >= 4096u || sp + 2 >= 24) { if (tid == 0) atomicOr(&gflags[0], 2u); break; }
            if (tid == 0) { s_st[sp] = (2 * P) | (pr << 16); s_st[sp + 1] = (2 * P) | ((pr + P) << 16); }
            sp += 2;
        }
        if (sp != 0) __syncthreads();
        LLPH(5);
    }
}
// resident workgroups take bins from gflags[BIN_TICKET] (bin_count_kernel's scheme, fastpath.hip); n_bins == 0: one workgroup per bin
template <int KW, int NBW, int NT, int T>
__global__ void __launch_bounds__(NT, NT == 512 ? 4 : 1) bin_labels_kernel(const uint64_t* __restrict__ recs, const uint64_t* __restrict__ recs_alt, uint32_t alt_from,
                                                       const uint64_t* __restrict__ seg_beg, const uint64_t* __restrict__ seg_end,
                                                       uint32_t n_src, uint64_t seg_stride,
                                                       int k, int stranded, uint64_t min_obs, FastOut out, uint64_t out_cap,
                                                       unsigned long long* __restrict__ out_cursor, ListOut lo, uint32_t* __restrict__ gflags, uint32_t n_bins) {
    if (n_bins == 0) {
        bin_labels_body<KW, NBW, NT, T>(blockIdx.x, recs, recs_alt, alt_from, seg_beg, seg_end, n_src, seg_stride, k, stranded, min_obs, out, out_cap, out_cursor, lo, gflags);
        return;
    }
    __shared__ uint32_t s_ticket;
    if (threadIdx.x == 0) s_ticket = atomicAdd(&gflags[BIN_TICKET], 1u);
    __syncthreads();
    for (;;) {
        const uint32_t bin = s_ticket;
        __syncthreads();
        if (bin >= n_bins) break;
        if (threadIdx.x == 0) s_ticket = atomicAdd(&gflags[BIN_TICKET], 1u);
        bin_labels_body<KW, NBW, NT, T>(bin, recs, recs_alt, alt_from, seg_beg, seg_end, n_src, seg_stride, k, stranded, min_obs, out, out_cap, out_cursor, lo, gflags);
        __syncthreads();
    }
}

// after the order-restoring sort: the payload columns spell the record's position in the unsorted output (Exts column = low byte,
// mask column = the other 24 bits); Exts, the number of labels and the segment's place are fetched from there
__global__ void __launch_bounds__(256) ll_meta_kernel(uint32_t n, uint8_t* __restrict__ exts_io, const uint32_t* __restrict__ q_hi24,
                                                      const uint4* __restrict__ w_rec, uint32_t* __restrict__ setn, uint64_t* __restrict__ seg_off) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t q = (uint32_t)exts_io[i] | (q_hi24[i] << 8);
    const uint4 r = w_rec[q];
    exts_io[i] = (uint8_t)r.w;
    setn[i] = r.z;
    seg_off[i] = (uint64_t)r.x | ((uint64_t)r.y << 32);
}

// label lists in table order: a wavefront takes 64 consecutive keys, whose lists form one contiguous stretch of set_val; lane e of a
// round writes element e of the stretch (consecutive lanes, consecutive words).  Which key an element belongs to: the keys whose
// lists start inside the round mark their first element in the wave's LDS row, a maximum scan (DPP) spreads the marks, elements before
// the first mark belong to the key the previous round ended in.  (First form: a binary search over the 64 offsets with shuffles, nine
// ds_bpermute per round: 31.5 ms for 9.9e9 labels.)  The reads run along the segments (a list is ~80 contiguous bytes at 30x):
// 84-byte runs at random alignment touch 2-3 sectors of 64 bytes, so the 29.6 ms are ~120 GB of traffic for 79 GB of labels moved --
// bandwidth, not latency (four rounds' loads in flight together: 29.9 ms).
__global__ void __launch_bounds__(256) ll_csr_kernel(uint32_t n, const uint64_t* __restrict__ set_off, const uint64_t* __restrict__ seg_off,
                                                     const uint32_t* __restrict__ lab, uint32_t* __restrict__ set_val) {
    __shared__ uint32_t s_ws[4][256];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t k0 = (blockIdx.x * 4 + wave) * 64;
    if (k0 >= n) return;
    uint32_t* const WS = s_ws[wave];
    const bool on = k0 + lane < n;
    const uint32_t i = on ? k0 + lane : n - 1;
    const uint64_t base = set_off[k0];
    const uint64_t end = set_off[k0 + 64 < n ? k0 + 64 : n];
    const uint32_t mine = (uint32_t)(set_off[i] - base);                 // my list's offset inside the wave's stretch (< 2^32: 64 lists)
    const uint64_t so = seg_off[i];
    WS[64 + lane] = mine; WS[128 + lane] = (uint32_t)so; WS[192 + lane] = (uint32_t)(so >> 32);
    const uint64_t len = end - base;
    uint32_t carry = 0;                                                  // key of the element before this round's first (uniform)
    for (uint64_t e0 = 0; e0 < len; e0 += 64) {
        WS[lane] = 0u;
        // (keys with an empty list cannot occur -- every valid k-mer has a label -- but a later key's mark would simply win)
        if (on && (uint64_t)mine >= e0 && (uint64_t)mine < e0 + 64) WS[mine - (uint32_t)e0] = lane + 1u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t ms = ll_wave_max_scan(WS[lane]);
        const uint32_t key = ms ? ms - 1u : carry;
        const uint32_t mo = WS[64 + key];
        const uint64_t src = ((uint64_t)WS[192 + key] << 32) | WS[128 + key];
        const uint64_t e = e0 + lane;
        if (e < len) set_val[base + e] = lab[src + (e - mo)];
        carry = (uint32_t)__builtin_amdgcn_readlane((int)key, 63);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ---- host side: counting state of the label-list route (the analogue of FastCountState's begin / bins / finish, so that the sharded
//      flow can count its exchange rounds one after the other and sort once) ----
struct ListCountState {
    FastCountState cs;                      // unsorted (key, output position) records, side records {segment, labels, Exts}, all keys
    DBuf<uint32_t> lab;                     // label segments
    uint64_t lab_cap = 0, lab_used = 0;
    DBuf<unsigned long long> lab_cursor;
    uint64_t min_obs = 0;
};

static void lists_plan(FastPlan* pl) {
    pl->lists = true; pl->rw = pl->nbw + 1;             // the label rides in an extra record word
    pl->wide = true;                                    // (payload = output position, side records: the WIDE plumbing of the count state)
    pl->lmap = nullptr; pl->linv.on = 0; pl->weighted = false;
    // 4-word records (k >= 56): the staging area and the duplicate filter leave room for two workgroups per CU only next to 1024-entry
    // tables, so the bins are half the size (as for the WIDE colour layout); with 2048 entries and one workgroup per CU the kernel took
    // 349 ms against 200 at k = 47 (C2 shape, 5000 labels)
    if (pl->nbw == 4) pl->nbins = (uint32_t)std::min<uint64_t>((uint64_t)pl->nbins * 2, (1ull << 23) - 1);
}

// labels of a label-list run must be < 2^24: the segment sort packs (segment, label) into 32 bits, and the C ABI demands it (dbg_mi355x.h)
static int lists_check_labels(dbg_ctx* c, const SeqDev& s) {
    uint32_t mx = 0;
    DBG_TRY(seq_max_label(c, s, &mx));
    if (mx >= (1u << 24)) return c->fail(17, "D1 values must be < 2^24");
    return 0;
}

static int lists_count_begin(dbg_ctx* c, const FastPlan& pl, uint64_t min_obs, uint64_t n_kmers_hint, uint64_t lab_cap, bool report_all, ListCountState* st) {
    st->min_obs = min_obs;
    st->lab_cap = std::max<uint64_t>(lab_cap, 1); st->lab_used = 0;
    if (!st->lab.p || st->lab.n < st->lab_cap) ALLOC_OR_FAIL(c, st->lab, st->lab_cap);       // (the caller may have reserved it already)
    ALLOC_OR_FAIL(c, st->lab_cursor, 1);
    return fast_count_begin(c, pl, min_obs, n_kmers_hint, &st->cs, report_all);
}

// bin_labels_kernel over `nbins_local` bins whose records arrive as n_src segments; valid k-mers and their label segments are appended
// to the state.  An output or label buffer that turns out too small is grown (what earlier calls produced is kept) and the launch redone.
static int lists_count_bins(dbg_ctx* c, ListCountState* st, const uint64_t* recs, const uint64_t* recs_alt, uint32_t alt_from,
                            const uint64_t* seg_beg, const uint64_t* seg_end, uint32_t n_src, uint64_t seg_stride, uint32_t nbins_local,
                            uint64_t n_kmers_units, uint64_t n_recs_hint) {
    FastCountState& cs = st->cs;
    const FastPlan& pl = cs.pl;
    const int k = pl.k;
    for (int attempt = 0;; attempt++) {
        unsigned long long start = cs.n_out, start_all = cs.n_all, start_lab = st->lab_used;
        HIP_TRY(c, hipMemcpyAsync(cs.out_cursor.p, &start, 8, hipMemcpyHostToDevice, c->stream));
        if (cs.report_all) HIP_TRY(c, hipMemcpyAsync(cs.all_cursor.p, &start_all, 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(st->lab_cursor.p, &start_lab, 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemsetAsync(cs.gflags.p, 0, 128, c->stream));
        FastOut fo{cs.u_hi.p, cs.u_lo.p, cs.u_pay.p, cs.use16 ? cs.u16.p : nullptr, cs.w_rec.p,
                   cs.report_all ? cs.a_hi.p : nullptr, cs.report_all ? cs.a_lo.p : nullptr, cs.report_all ? cs.all_cursor.p : nullptr, cs.all_cap};
        ListOut lo{st->lab.p, st->lab_cap, st->lab_cursor.p};
        if (nbins_local) {
            if ((uint64_t)nbins_local * 512 >= (1ull << 32)) return c->fail(135, "fast path: too many bins for the workgroup counting kernel (a grid holds < 2^32 threads)");
            c->t_begin("bin_labels", n_kmers_units);
const bool resident = !(c->opt("DBG_FAST_PERSIST") && atoi(c->opt("DBG_FAST_PERSIST")) == 0);
            const uint32_t n_bins_arg = resident ? nbins_local : 0u;
#define ARGS_ recs, recs_alt, alt_from, seg_beg, seg_end, n_src, seg_stride, k, pl.stranded ? 1 : 0, st->min_obs, fo, cs.cap, cs.out_cursor.p, lo, cs.gflags.p, n_bins_arg
#define LK(...) do { auto kern = bin_labels_kernel<__VA_ARGS__>; \
            const uint32_t grid_wg = resident ? std::min(nbins_local, resident_grid(c, (const void*)kern, 512, 0, c->opt("DBG_FAST_PERSIST"))) : nbins_local; \
            kern<<<grid_wg, 512, 0, c->stream>>>(ARGS_); } while (0)
            if (!pl.has_hi) LK(1, 2, 512, 2048);
            else if (pl.nbw == 2) LK(2, 2, 512, 2048);
            else if (pl.nbw == 3) LK(2, 3, 512, 2048);
            else LK(2, 4, 512, 1024);       // (k >= 56: see lists_plan)
#undef LK
#undef ARGS_
            c->t_end();
            LAUNCH_CHECK(c, "bin_labels");
        }
        unsigned long long cur = 0, cur_all = 0, cur_lab = 0;
        uint32_t flv[16] = {0};
        HIP_TRY(c, hipMemcpyAsync(&cur, cs.out_cursor.p, 8, hipMemcpyDeviceToHost, c->stream));
        if (cs.report_all) HIP_TRY(c, hipMemcpyAsync(&cur_all, cs.all_cursor.p, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(&cur_lab, st->lab_cursor.p, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(flv, cs.gflags.p, 64, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
#ifdef DBG_LL_PHASE_TIMES
        if (c->opt("DBG_DEBUG")) {
            unsigned long long ph[8];
            (void)hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_ll_phase), sizeof(ph));
            fprintf(stderr, "[label-lists phases] (100 MHz ticks of thread 0, summed over bins) prologue=%llu count=%llu segments=%llu append=%llu sort=%llu emit=%llu\n",
                    ph[0], ph[1], ph[2], ph[3], ph[4], ph[5]);
            unsigned long long z[8] = {0};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ll_phase), z, sizeof(z));
        }
#endif
        if (c->opt("DBG_DEBUG")) fprintf(stderr, "[fastpath] label lists: bins=%u srcs=%u recs=%llu valid=%llu labels appended=%llu (room for %llu) flags=%u maxP=%u split_passes=%u wd=%u\n",
                                         nbins_local, n_src, (unsigned long long)n_recs_hint, cur, cur_lab, (unsigned long long)st->lab_cap, flv[0], flv[1], flv[2], flv[3]);
        if (flv[3] & 4u) return c->fail(134, "fast path: corrupt super-k-mer record (record buffer or segment table of the counting stage is wrong)");
        if (flv[3]) return c->fail(132, "fast path: internal watchdog fired");
        if (flv[0] & 6u) return c->fail(130, "fast path: a bin exceeded the multi-pass limit");
        if (flv[0] & 25u) {
            if (cur >= (1ull << 32) || cur_all >= (1ull << 32)) return c->fail(131, "fast path: more than 2^32-1 k-mers in one table");
            if (attempt >= 4) return c->fail(138, "fast path: the output buffers of the label-list run kept turning out too small");
            if (flv[0] & 1u) DBG_TRY(fast_count_alloc(c, &cs, cur + cur / 16 + 1024));
            if (flv[0] & 8u) DBG_TRY(fast_count_alloc_all(c, &cs, cur_all + cur_all / 16 + 1024));
            if (flv[0] & 16u) {                                  // (the cursor went on counting: it holds what this launch wants)
                const uint64_t want = cur_lab + cur_lab / 16 + 4096;
                DBuf<uint32_t> nl;
                ALLOC_OR_FAIL(c, nl, want);
                if (st->lab_used) HIP_TRY(c, hipMemcpyAsync(nl.p, st->lab.p, st->lab_used * 4, hipMemcpyDeviceToDevice, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                std::swap(st->lab, nl);
                st->lab_cap = want;
            }
            continue;
        }
        cs.n_out = cur; cs.n_all = cur_all; st->lab_used = cur_lab;
        break;
    }
    return 0;
}

// order-restoring sort of everything counted so far, then the lists in table order
static int lists_count_finish(dbg_ctx* c, ListCountState* st, dbg_kmer_table* out) {
    FastCountState& cs = st->cs;
    const FastPlan& pl = cs.pl;
    const int k = pl.k;
    const bool has_hi = pl.has_hi, report_all = cs.report_all;
    const uint64_t n_out = cs.n_out;
    const size_t na = std::max<uint64_t>(n_out, 1);
    DBuf<uint64_t> seg_off;
    DBuf<uint32_t> setn;
    ALLOC_OR_FAIL(c, seg_off, na); ALLOC_OR_FAIL(c, setn, na);
    {
        DBuf<uint32_t> t_pay, msk_sorted;
        DBuf<uint64_t> t_hi, t_lo, o_hi, o_lo, o_set_off;
        DBuf<uint8_t> o_exts;
        DBuf<uint32_t> o_set_val;
        if (!cs.use16) {
            ALLOC_OR_FAIL(c, t_pay, na); ALLOC_OR_FAIL(c, t_lo, na);
            if (has_hi) ALLOC_OR_FAIL(c, t_hi, na);
        }
        ALLOC_OR_FAIL(c, o_hi, na); ALLOC_OR_FAIL(c, o_lo, na); ALLOC_OR_FAIL(c, o_exts, na);
        ALLOC_OR_FAIL(c, msk_sorted, na); ALLOC_OR_FAIL(c, o_set_off, na + 1);
        if (cs.use16) {
            DBuf<uint4> t16;
            ALLOC_OR_FAIL(c, t16, na);
            DBG_TRY(sort_table_hybrid16(c, n_out, cs.u16.p, t16.p, 2 * k, true, !c->opt("DBG_NO_HYBRID_SORT"), o_hi.p, o_lo.p, o_exts.p, nullptr, nullptr, msk_sorted.p));
        } else {
            RecArrays A{has_hi ? cs.u_hi.p : nullptr, cs.u_lo.p, cs.u_pay.p}, B{has_hi ? t_hi.p : nullptr, t_lo.p, t_pay.p};
            DBG_TRY(sort_table_hybrid(c, n_out, A, B, 2 * k, true, !c->opt("DBG_NO_HYBRID_SORT"), o_hi.p, o_lo.p, o_exts.p, nullptr, nullptr, msk_sorted.p));
        }
        uint64_t n_setval = 0;
        if (n_out) {
            c->t_begin("list_meta", n_out);
            ll_meta_kernel<<<cdiv(n_out, 256), 256, 0, c->stream>>>((uint32_t)n_out, o_exts.p, msk_sorted.p, cs.w_rec.p, setn.p, seg_off.p);
            c->t_end();
            LAUNCH_CHECK(c, "ll_meta");
            DBG_TRY(scan_exclusive_u32_u64(c, setn.p, o_set_off.p, n_out));
            HIP_TRY(c, hipMemcpyAsync(&n_setval, o_set_off.p + n_out, 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        } else HIP_TRY(c, hipMemsetAsync(o_set_off.p, 0, 8, c->stream));
        ALLOC_OR_FAIL(c, o_set_val, std::max<uint64_t>(n_setval, 1));
        if (n_out) {
            c->t_begin("set_csr", n_out);
            ll_csr_kernel<<<cdiv(n_out, 256), 256, 0, c->stream>>>((uint32_t)n_out, o_set_off.p, seg_off.p, st->lab.p, o_set_val.p);
            c->t_end();
            LAUNCH_CHECK(c, "ll_csr");
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        memset(out, 0, sizeof(*out));
        out->n = n_out;
        out->key_hi = o_hi.take(); out->key_lo = o_lo.take(); out->exts = o_exts.take();
        out->set_off = o_set_off.take(); out->set_val = o_set_val.take(); out->n_set_val = n_setval;
        out->n_kmer_instances = cs.n_kmers_hint; out->n_passes = 1; out->on_device = 1;
    }
    if (report_all) {
        const uint64_t n_all = cs.n_all;
        const size_t naa = std::max<uint64_t>(n_all, 1);
        DBuf<uint64_t> b_hi, b_lo, f_hi, f_lo;
        DBuf<uint32_t> a_pay, b_pay;
        DBuf<uint8_t> x_exts;
        DBuf<uint16_t> x_count;
        ALLOC_OR_FAIL(c, a_pay, naa); ALLOC_OR_FAIL(c, b_pay, naa); ALLOC_OR_FAIL(c, b_lo, naa);
        ALLOC_OR_FAIL(c, f_hi, naa); ALLOC_OR_FAIL(c, f_lo, naa); ALLOC_OR_FAIL(c, x_exts, naa); ALLOC_OR_FAIL(c, x_count, naa);
        if (has_hi) ALLOC_OR_FAIL(c, b_hi, naa);
        HIP_TRY(c, hipMemsetAsync(a_pay.p, 0, naa * 4, c->stream));
        RecArrays A2{has_hi ? cs.a_hi.p : nullptr, cs.a_lo.p, a_pay.p}, B2{has_hi ? b_hi.p : nullptr, b_lo.p, b_pay.p};
        DBG_TRY(sort_table_hybrid(c, n_all, A2, B2, 2 * k, false, !c->opt("DBG_NO_HYBRID_SORT"), f_hi.p, f_lo.p, x_exts.p, x_count.p, nullptr, nullptr));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        out->n_all = n_all;
        out->all_hi = f_hi.take(); out->all_lo = f_lo.take();
    }
    return 0;
}

// *used = false (and nothing written): not enough device memory for the label buffer -- the caller takes label groups or the generic path
static int filter_kmers_fast_lists(dbg_ctx* c, const SeqDev& s, const dbg_filter_params* prm, uint64_t n_kmers, dbg_kmer_table* out, bool* used) {
    *used = false;
    const int k = (int)prm->k;
    FastPlan pl;
    if (!fast_make_plan(c, k, prm->stranded != 0, true, n_kmers, 0, &pl)) return 0;
    lists_plan(&pl);
    DBG_TRY(lists_check_labels(c, s));
    ListCountState st;
    // every instance of a valid k-mer appends one label: n_kmers places always suffice
    if (!st.lab.alloc(c, std::max<uint64_t>(n_kmers, 1))) return 0;
    if (c->opt("DBG_DEBUG")) fprintf(stderr, "[fastpath] label lists: %llu k-mer instances, label buffer %.2f GB\n", (unsigned long long)n_kmers, n_kmers * 4e-9);

    FastScan sc;
    DBG_TRY(fast_scan(c, s, pl, n_kmers, &sc, true));
    const uint32_t nb = pl.nbins * NCLS;
    DBuf<uint64_t> ovf_off, ovf_recs, seg;
    DBuf<unsigned long long> total;
    ALLOC_OR_FAIL(c, ovf_off, (size_t)nb + 1);
    DBG_TRY(fast_bin_offsets(c, &sc, ovf_off.p, false));
    ALLOC_OR_FAIL(c, ovf_recs, std::max<uint64_t>(sc.n_recs * pl.rw, 1));
    DBuf<uint64_t> slab;
    DBuf<uint32_t> cursor;
    std::swap(slab, sc.slab); std::swap(cursor, sc.cursor);
    const uint32_t slab_cap = sc.slab_cap;
    DBG_TRY(fast_scatter(c, &sc, ovf_off.p, ovf_recs.p));
    ALLOC_OR_FAIL(c, seg, (size_t)nb * 4);
    ALLOC_OR_FAIL(c, total, 1);
    HIP_TRY(c, hipMemsetAsync(total.p, 0, 8, c->stream));
    slab_bounds_kernel<<<cdiv(nb, 256), 256, 0, c->stream>>>(cursor.p, slab_cap, ovf_off.p, nb, seg.p, seg.p + 2 * (size_t)nb, total.p);
    LAUNCH_CHECK(c, "slab_bounds");
    unsigned long long n_recs_total = 0;
    HIP_TRY(c, hipMemcpyAsync(&n_recs_total, total.p, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->t_begin("sk_records", n_recs_total);
    c->t_end();
    DBG_TRY(lists_count_begin(c, pl, prm->min_kmer_obs, n_kmers, n_kmers, prm->report_all_kmers != 0, &st));
    DBG_TRY(lists_count_bins(c, &st, slab.p, ovf_recs.p, 1, seg.p, seg.p + 2 * (size_t)nb, sc.n_recs ? 2u : 1u, (uint64_t)nb, pl.nbins, n_kmers, n_recs_total));
    slab.release(); ovf_recs.release();
    DBG_TRY(lists_count_finish(c, &st, out));
    *used = true;
    return 0;
}
