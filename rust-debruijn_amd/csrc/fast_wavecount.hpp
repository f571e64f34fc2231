// Wave-autonomous counting kernel of the fast path (included by fastpath.hip inside its anonymous namespace).
//
// One WAVEFRONT owns one (small) bin and a private LDS hash table of T entries; a workgroup IS one wavefront, so the kernel
// has no workgroup barrier at all.  The workgroup form (bin_count_kernel: 512 threads, one 2048-entry table per bin) spends a
// bin's time in ~17 barrier-separated phases whose dependent LDS round trips a second resident workgroup only partly hides
// (66 % VALU busy, 50 % of the wave cycles waiting).  Here every wave walks through its own bins at its
// own pace; LDS operations of one wave are performed in order, which replaces every barrier and every "claimed / busy"
// hand-shake of the workgroup form, and the order of insertion is under control:
//   * records (super-k-mers) arrive 64 at a time, one per lane; identical records (the ~18 reads across one stretch of the
//     genome yield the same piece) are merged into one staged record with a weight and a colour set -- a 128-entry hash filter,
//     first of its kind claims (LDS CAS), the others compare themselves with the staged record in the same iteration;
//   * the staged records are cut into chunks of <= 4 k-mers; chunks of records that stand for several reads (whole pieces)
//     are dealt first, single-copy records (pieces cut short by a read end -- sub-strings of a whole piece -- and pieces with
//     a sequencing error) afterwards: their k-mers then mostly FIND their table entry instead of racing each other for it
//     (the workgroup form loses 46 % of its insertions to a lane inserting the same k-mer at the same moment);
//   * valid entries leave through a per-wave output allocator: a wave reserves WV_CHUNK output records with ONE global atomic
//     (same-address atomics complete at ~10^8 per second: one per bin would serialise the launch) and fills them bin after bin;
//     the unused tail of every wave's last chunk is a hole that fill_holes() closes afterwards by moving the last records of the
//     output into the holes (the output is unordered at this point: the order-restoring sort follows).
// Reference semantics: CountFilter / CountFilterSet::summarize (src/filter.rs:53-62, :85-100) over the k-mers + Exts of
// iter_kmer_exts (src/lib.rs:812-841), canonicalised by min_rc_flip (lib.rs:224-231) -- exactly as bin_count_kernel.
// WV_ABL = n (measurement builds, tools/abl_wave.sh): the kernel stops after phase n of a bin -- 1 segment bounds + table clear +
// first records, 2 records staged, 3 chunk map built, 4 chunks set up (no k-mer is rolled or inserted), 5 everything but emission
#ifndef WV_ABL
#define WV_ABL 0
#endif
constexpr uint32_t WV_CHUNK = 512;          // output records per reservation
constexpr uint32_t WV_DD = 128;             // duplicate-filter slots (2 per staged record)
constexpr uint32_t WV_CAPC = 384;           // chunk-map entries
constexpr uint32_t WV_MIN_ROOM = 16;        // keep filling the staging area while at least this many slots are free

struct WaveHole { unsigned long long start; unsigned long long len; };

__device__ __forceinline__ uint32_t rl32(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ uint64_t rl64(uint64_t v, uint32_t l) { return (uint64_t)rl32((uint32_t)v, l) | ((uint64_t)rl32((uint32_t)(v >> 32), l) << 32); }
__device__ __forceinline__ uint64_t wave_inclusive_scan_4x16(uint64_t v) {
    return (uint64_t)wave_inclusive_scan_u32((uint32_t)v) | ((uint64_t)wave_inclusive_scan_u32((uint32_t)(v >> 32)) << 32);
}
__device__ __forceinline__ uint32_t fsum4x16(uint64_t x) {
    return (uint32_t)(x & 0xffff) + (uint32_t)((x >> 16) & 0xffff) + (uint32_t)((x >> 32) & 0xffff) + (uint32_t)(x >> 48);
}

template <int KW, int NBW, bool IS_SET, int T>
__global__ void __launch_bounds__(64) bin_count_wave_kernel(const uint64_t* __restrict__ recs, const uint64_t* __restrict__ recs_alt, uint32_t alt_from,
                                                            const uint64_t* __restrict__ seg_beg, const uint64_t* __restrict__ seg_end,
                                                            uint32_t n_src, uint64_t seg_stride, uint32_t nbins,
                                                            int k, int stranded, uint64_t min_obs, FastOut out, uint64_t out_cap,
                                                            unsigned long long* __restrict__ out_cursor, uint32_t* __restrict__ gflags,
                                                            WaveHole* __restrict__ holes) {
    constexpr int RW = NBW;
    constexpr uint32_t CH = 4;
    static_assert(T % 256 == 0 && T >= 256, "the table is cleared and emitted in steps of 256 slots");
    __shared__ __attribute__((aligned(16))) uint32_t s_tag[T];
    __shared__ __attribute__((aligned(16))) uint64_t s_key[KW * T];
    __shared__ __attribute__((aligned(16))) uint32_t s_cnt[T];
    __shared__ __attribute__((aligned(16))) uint32_t s_aux[T];          // Exts | colour mask << 8
    __shared__ uint64_t s_slab[RW * 64];                                // staged (distinct) records, word-major
    __shared__ __attribute__((aligned(8))) uint16_t s_cmap[WV_CAPC];    // chunk -> staged slot | chunk index << 6
    uint32_t* const s_dd = reinterpret_cast<uint32_t*>(s_cmap);         // duplicate filter (dead before the chunk map is built)
    static_assert(WV_CAPC * 2 >= WV_DD * 4, "the duplicate filter borrows the chunk map");
    typedef typename std::conditional<(T > 256), uint16_t, uint8_t>::type slot_t;
    __shared__ slot_t s_list[T];                                        // occupied table entries in the order they were claimed
    __shared__ uint32_t s_w[64];                                        // copies each staged record stands for
    __shared__ uint32_t s_cmk[IS_SET ? 64 : 1];                         // union of their colours

    const uint32_t lane = threadIdx.x;
    const uint64_t lt = lanemask_lt();
    const K128 kmask = k128_mask(k);
    constexpr uint64_t COLOUR_BITS = 63ull << 15;
    unsigned long long o_pos = 0, o_end = 0;                            // this wave's output chunk (uniform)
    const uint64_t alt_delta = (uint64_t)((uintptr_t)recs_alt - (uintptr_t)recs);       // bytes from the first record buffer to the second
    uint32_t fl_bad = 0;

    for (uint32_t bin = blockIdx.x; bin < nbins; bin += gridDim.x) {
        // ---- the bin's records: n_src segments (its slab + its overflow; one per source rank after an exchange); lane s
        //      keeps segment s: first record, and how many records of the bin come before it ----
        uint64_t sbeg = 0;
        uint32_t slen = 0;
        if (lane < n_src) {
            sbeg = seg_beg[lane * seg_stride + (uint64_t)bin * NCLS];
            slen = (uint32_t)(seg_end[lane * seg_stride + (uint64_t)(bin + 1) * NCLS - 1] - sbeg);
        }
        const uint32_t sincl = wave_inclusive_scan_u32(slen), spre = sincl - slen;
        const uint32_t total_recs = rl32(sincl, 63);
        if (total_recs == 0) continue;

        uint64_t P0 = 0, P1 = 0, P2 = 0, P3 = 0;
        // record ridx of the bin -> P0..P3 (zero past the end).  One base pointer + an offset for the second buffer: a select
        // between two pointer arguments made the compiler build a pointer table in scratch memory and use flat loads.
#define WV_LOAD_REC(RIDX) do { \
            const uint32_t ridx_ = (RIDX); \
            uint64_t base_ = 0, boff_ = 0; uint32_t pre_ = 0; \
            for (uint32_t s_ = 0; s_ < n_src; s_++) {               /* uniform: segment bounds come from lane s_ */ \
                const uint32_t ps_ = rl32(spre, s_); \
                const uint64_t pb_ = rl64(sbeg, s_); \
                if (ridx_ >= ps_) { pre_ = ps_; base_ = pb_; boff_ = s_ >= alt_from ? alt_delta : 0ull; } \
            } \
            P0 = P1 = P2 = P3 = 0; \
            if (ridx_ < total_recs) { \
                const uint64_t* g_ = (const uint64_t*)((uintptr_t)recs + boff_) + (base_ + (ridx_ - pre_)) * RW; \
                P0 = g_[0]; P1 = g_[1]; \
                if (NBW > 2) P2 = g_[2]; \
                if (NBW > 3) P3 = g_[3]; \
            } } while (0)

        // Hash-selected passes (P, r): the pass handles the keys with (hash >> 16) % P == r.  A pass whose distinct keys overflow
        // the table emits nothing and is replaced by its children (2P, r), (2P, r + P).  The walk over the tree of passes needs
        // no stack: after (P, r) comes its sibling when it is a first child (r < P/2), else the successor of its parent.
        uint32_t P = 1, pr = 0;
        for (uint32_t guard = 0;; guard++) {
            if (guard > 20000u) { fl_bad |= 2u; break; }
            WV_LOAD_REC(lane);                                          // the first records are on their way while the table is cleared
#pragma unroll
            for (int i = 0; i < T; i += 256) {
                *reinterpret_cast<uint4*>(&s_tag[i + lane * 4]) = make_uint4(0, 0, 0, 0);
                *reinterpret_cast<uint4*>(&s_cnt[i + lane * 4]) = make_uint4(0, 0, 0, 0);
                *reinterpret_cast<uint4*>(&s_aux[i + lane * 4]) = make_uint4(0, 0, 0, 0);
            }
            uint32_t n_ins = 0;                                         // occupied table entries (uniform)
            bool pass_ovf = false, bad_bin = false;
            if (WV_ABL == 1) { if (P0 + P1 + P2 + P3 == 0x1234567ull) fl_bad |= 16u; break; }
            uint32_t rnext = 0;
            while (rnext < total_recs && !pass_ovf) {
                // ---- stage distinct records ----
                *reinterpret_cast<uint2*>(&s_dd[lane * 2]) = make_uint2(0, 0);
                uint32_t nst = 0;
                for (;;) {
                    const uint32_t room = 64u - nst;
                    const uint32_t take = total_recs - rnext < room ? total_recs - rnext : room;
                    bool pend = lane < take;
                    const uint64_t PL0 = NBW == 2 ? P1 : (NBW == 3 ? P2 : P3);          // word holding the meta bits
                    uint32_t mytag = 0, sl = 0, colour = 0;
                    if (pend) {
                        const uint32_t rl = (uint32_t)(PL0 & 0x7f);
                        if (rl < (uint32_t)k || rl > (uint32_t)(32 * NBW - (META_BITS + 1) / 2)) { fl_bad |= 4u; pend = false; bad_bin = true; }
                        const uint64_t lastw = PL0 & ~COLOUR_BITS;
                        uint64_t ha = P0, hb = NBW == 2 ? lastw : P1;
                        if (NBW == 3) ha += lastw * 0x9E3779B97F4A7C15ull;
                        if (NBW == 4) { ha += P2 * 0x9E3779B97F4A7C15ull; hb += lastw * 0xC2B2AE3D27D4EB4Full; }
                        const uint64_t h = hash_key(ha, hb);
                        mytag = (uint32_t)(h >> 42) << 10;
                        colour = 1u << ((uint32_t)(PL0 >> 15) & 31u);
                        sl = (uint32_t)h & (WV_DD - 1);
                    }
                    if (__any(bad_bin)) { bad_bin = true; break; }
                    while (__any(pend)) {
                        uint32_t v = 1u;
                        bool won = false;
                        if (pend) {
                            v = s_dd[sl];
                            if (v == 0u) { v = atomicCAS(&s_dd[sl], 0u, 1023u | mytag); won = v == 0u; }
                        }
                        const uint64_t wm = __ballot(won);
                        if (won) {                                      // first of its kind: stage it
                            const uint32_t mine = nst + (uint32_t)__popcll(wm & lt);
                            s_slab[mine] = P0; s_slab[64 + mine] = P1;
                            if (NBW > 2) s_slab[128 + mine] = P2;
                            if (NBW > 3) s_slab[192 + mine] = P3;
                            s_w[mine] = 1u;
                            if (IS_SET) s_cmk[mine] = colour;
                            s_dd[sl] = (mine + 1u) | mytag;
                            pend = false;
                        }
                        nst += (uint32_t)__popcll(wm);
                        if (pend) {
                            // the entry carries a record's hash bits; one that was claimed in this very iteration has been published by
                            // now (the winner's stores above precede this load in program order, and a wave's LDS operations are in order)
                            bool same = false;
                            uint32_t r = 0;
                            if ((v & ~1023u) == mytag) {
                                r = (s_dd[sl] & 1023u) - 1u;
                                same = s_slab[r] == P0;
                                if (NBW > 2) same = same && s_slab[64 + r] == P1;
                                if (NBW > 3) same = same && s_slab[128 + r] == P2;
                                same = same && ((s_slab[(NBW - 1) * 64 + r] ^ PL0) & ~COLOUR_BITS) == 0;
                            }
                            if (same) {
                                atomicAdd(&s_w[r], 1u);
                                if (IS_SET) atomicOr(&s_cmk[r], colour);
                                pend = false;
                            } else sl = (sl + 1u) & (WV_DD - 1);
                        }
                    }
                    rnext += take;
                    if (rnext >= total_recs) break;
                    WV_LOAD_REC(rnext + lane);                          // the next incoming records are on their way
                    if (64u - nst < WV_MIN_ROOM) break;
                }
                if (bad_bin) break;
                if (WV_ABL == 2) continue;

                // ---- cut the staged records into chunks of <= CH k-mers; map order = (several copies first, longer chunks first) ----
                uint32_t ngroups = 1;
                for (uint32_t g = 0; g < ngroups; g++) {
                    const uint32_t gsz = 64u / ngroups;
                    const bool have = lane < nst && lane / gsz == g;
                    uint32_t nkr = 0;
                    bool heavy = false;
                    if (have) {
                        nkr = (uint32_t)(s_slab[(NBW - 1) * 64 + lane] & 0x7f) - (uint32_t)k + 1u;
                        heavy = s_w[lane] >= 2u;
                    }
                    const uint32_t nch = (nkr + CH - 1) / CH;
                    const uint32_t cb = nch ? nkr / nch : 0u, cr = nkr - cb * nch;   // chunk c: cb + (c < cr) k-mers
                    const uint32_t cl_hi = CH - cb - 1, cl_lo = CH - cb;              // class (4 - length) of the cr longer / nch - cr shorter chunks
                    uint64_t contrib = 0;
                    if (nch) contrib = (cr ? (uint64_t)cr << (16 * cl_hi) : 0ull) + ((uint64_t)(nch - cr) << (16 * cl_lo));
                    const uint64_t cH = heavy ? contrib : 0ull, cL = heavy ? 0ull : contrib;
                    const uint64_t iH = wave_inclusive_scan_4x16(cH), iL = wave_inclusive_scan_4x16(cL);
                    const uint64_t tH = rl64(iH, 63), tL = rl64(iL, 63);
                    const uint32_t nH = fsum4x16(tH), totc = nH + fsum4x16(tL);
                    if (totc > WV_CAPC) { ngroups *= 2; g = ~0u; continue; }         // (long records, all different: halves, then quarters)
                    if (nch) {
                        const uint64_t excl = (heavy ? iH : iL) - contrib, tot = heavy ? tH : tL;
                        const uint32_t off = heavy ? 0u : nH;
                        const uint32_t pos_lo = off + fsum4x16(tot & ((1ull << (16 * cl_lo)) - 1)) + (uint32_t)((excl >> (16 * cl_lo)) & 0xffff);
                        for (uint32_t c = cr; c < nch; c++) s_cmap[pos_lo + c - cr] = (uint16_t)(lane | (c << 6));
                        if (cr) {
                            const uint32_t pos_hi = off + fsum4x16(tot & ((1ull << (16 * cl_hi)) - 1)) + (uint32_t)((excl >> (16 * cl_hi)) & 0xffff);
                            for (uint32_t c = 0; c < cr; c++) s_cmap[pos_hi + c] = (uint16_t)(lane | (c << 6));
                        }
                    }
                    if (WV_ABL == 3) continue;
                    // ---- the chunks, 64 at a time ----
                    bool full = false;
                    for (uint32_t q0 = 0; q0 < totc; q0 += 64) {
                        const uint32_t q = q0 + lane;
                        const bool act = q < totc;
                        const uint32_t e = act ? (uint32_t)s_cmap[q] : 0u;
                        const uint32_t r = e & 63u, c = e >> 6;
                        uint64_t W0 = s_slab[r], W1 = s_slab[64 + r], W2 = 0, W3 = 0;
                        if (NBW > 2) W2 = s_slab[128 + r];
                        if (NBW > 3) W3 = s_slab[192 + r];
                        uint64_t meta;
                        {
                            uint64_t& WL = NBW == 2 ? W1 : (NBW == 3 ? W2 : W3);
                            meta = WL & ((1ull << META_BITS) - 1);
                            WL &= ~((1ull << META_BITS) - 1);
                        }
                        const uint32_t rlen = (uint32_t)(meta & 0x7f), rexts = (uint32_t)(meta >> 7) & 0xffu;
                        const uint32_t wgt = s_w[r], cset = IS_SET ? s_cmk[r] << 8 : 0u;
                        const uint32_t rnk = rlen - (uint32_t)k + 1u, rnch = (rnk + 3u) >> 2;
                        const uint32_t cbase = rnk < 4u ? rnk : (rnk == 5u ? 2u : ((rnk & 3u) ? 3u : 4u)), crem = rnk - cbase * rnch;
                        uint32_t j = c * cbase + (c < crem ? c : crem);
                        const uint32_t jend = act ? j + cbase + (c < crem ? 1u : 0u) : j;
                        K128 fw;
                        {
                            const uint32_t sft = 2 * j, ws = sft >> 6, bs = sft & 63;
                            const uint64_t A = ws == 0 ? W0 : W1, B = ws == 0 ? W1 : (NBW > 2 ? W2 : 0ull),
                                           C = ws == 0 ? (NBW > 2 ? W2 : 0ull) : (NBW > 3 ? W3 : 0ull);
                            const uint64_t h = bs ? (A << bs) | (B >> (64 - bs)) : A, l = bs ? (B << bs) | (C >> (64 - bs)) : B;
                            fw = k128_shr(K128{h, l}, 128 - 2 * k);
                        }
                        K128 rcw = kmer_rc(fw, k);
                        uint32_t lb = 0;
                        if (j) {
                            const uint32_t qb = j - 1;
                            const uint64_t wd = qb < 32 ? W0 : (qb < 64 ? W1 : (NBW > 2 && qb < 96 ? W2 : (NBW > 3 ? W3 : (NBW > 2 ? W2 : W1))));
                            lb = (uint32_t)(wd >> (62 - 2 * (qb & 31))) & 3u;
                        }
                        uint32_t nx;
                        {
                            const uint32_t sft = 2 * (j + (uint32_t)k), ws = sft >> 6, bs = sft & 63;
                            const uint64_t A = ws == 0 ? W0 : (ws == 1 ? W1 : (NBW > 2 && ws == 2 ? W2 : (NBW > 3 && ws == 3 ? W3 : 0ull)));
                            const uint64_t B = ws == 0 ? W1 : (NBW > 2 && ws == 1 ? W2 : (NBW > 3 && ws == 2 ? W3 : 0ull));
                            const uint64_t v = bs ? (A << bs) | (B >> (64 - bs)) : A;
                            nx = (uint32_t)(v >> 32);
                        }
                        if (WV_ABL == 4) { if (fw.lo + rcw.hi + lb + nx + wgt + cset + rexts == 0x1234567u) fl_bad |= 16u; continue; }
                        while (__any(j < jend)) {
                            bool ins = false;
                            uint32_t ins_slot = 0;
                            if (j < jend) {
                                const uint32_t nbase = nx >> 30;
                                nx <<= 2;
                                {
                                    const uint32_t left = j == 0 ? (rexts & 0xfu) : (1u << lb);
                                    const uint32_t right = (j + (uint32_t)k == rlen) ? (rexts & 0xf0u) : (16u << nbase);
                                    uint32_t ex = left | right;
                                    K128 km = fw;
                                    if (!stranded && !k128_lt(fw, rcw)) { km = rcw; ex = __brev(ex) >> 24; }
                                    const uint64_t h = hash_key(km.hi, km.lo);
                                    if (P == 1 || ((uint32_t)(h >> 16) & (P - 1)) == pr) {
                                        // bucketised linear probing, 4 tags per 16-byte bucket (as bin_count_kernel); no BUSY state: the table
                                        // is private to this wave, whose LDS operations are performed in order -- the key store of a claimer
                                        // precedes every later look at its tag
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
                                                const uint32_t i = (uint32_t)__ffs((int)mm) - 1u, s2 = bkt * 4 + i;
                                                bool same;
                                                if (KW == 2) {
                                                    const ulonglong2 kk = *reinterpret_cast<const ulonglong2*>(&s_key[2 * s2]);
                                                    same = kk.x == km.lo && kk.y == km.hi;
                                                } else same = s_key[s2] == km.lo;
                                                if (same) { hit = true; slot = s2; break; }
                                                tried |= 1u << i;
                                                continue;
                                            }
                                            if (em) {
                                                const uint32_t emr = ((em >> rot) | (em << (4u - rot))) & 15u;
                                                const uint32_t s2 = bkt * 4 + (((uint32_t)__ffs((int)emr) - 1u + rot) & 3u);
                                                if (atomicCAS(&s_tag[s2], 0u, mytag) == 0u) {
                                                    if (KW == 2) *reinterpret_cast<ulonglong2*>(&s_key[2 * s2]) = make_ulonglong2(km.lo, km.hi);
                                                    else s_key[s2] = km.lo;
                                                    asm volatile("" ::: "memory");
                                                    hit = true; slot = s2; ins = true; ins_slot = s2;
                                                    break;
                                                }
                                                continue;                                    // lost the slot to another lane: look at the bucket again
                                            }
                                            bkt = (bkt + 1) & (T / 4 - 1);
                                            tried = 0;
                                            if (++nprobe >= (uint32_t)(T / 4)) break;        // table full
                                        }
                                        if (hit) {
                                            atomicAdd(&s_cnt[slot], wgt);
                                            atomicOr(&s_aux[slot], ex | cset);
                                        } else full = true;
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
                            const uint64_t im = __ballot(ins);                   // new entries join the list the emission walks
                            if (ins) s_list[n_ins + (uint32_t)__popcll(im & lt)] = (slot_t)ins_slot;
                            n_ins += (uint32_t)__popcll(im);
                        }
                        if (__any(full)) { pass_ovf = true; break; }
                    }
                    if (pass_ovf) break;
                }
                // a table more than 7/8 full makes the remaining records probe long chains: re-split the pass early
                if (!pass_ovf && rnext < total_recs && n_ins > (uint32_t)(T - T / 8)) pass_ovf = true;
            }
            if (bad_bin) break;                                         // corrupt input: the launch fails (gflags)
            if (WV_ABL >= 2) break;

            if (!pass_ovf) {
                // ---- emit the valid entries of this pass through the wave's output allocator: the list of claimed entries is
                //      walked 64 at a time (a 256-entry table holds ~100 k-mers: two steps instead of a scan of all the slots) ----
                for (uint32_t eb = 0; eb < n_ins; eb += 64) {
                    const uint32_t idx = eb + lane;
                    const uint32_t i = idx < n_ins ? (uint32_t)s_list[idx] : 0u;
                    const uint32_t c = s_cnt[i];
                    const bool valid = idx < n_ins && (IS_SET ? (uint64_t)c >= min_obs : (uint64_t)(c > 65535u ? 65535u : c) >= min_obs);
                    const uint64_t vm = __ballot(valid);
                    if (!vm) continue;
                    const uint32_t nv = (uint32_t)__popcll(vm);
                    const uint32_t room = (uint32_t)(o_end - o_pos);
                    unsigned long long nbase = 0;
                    if (nv > room) {
                        if (lane == 0) nbase = atomicAdd(out_cursor, (unsigned long long)WV_CHUNK);
                        nbase = rl64(nbase, 0);
                        if (nbase + WV_CHUNK > out_cap) fl_bad |= 1u;                 // output buffer too small: the host grows it and repeats the launch
                    }
                    if (valid) {
                        const uint32_t rk = (uint32_t)__popcll(vm & lt);
                        const unsigned long long q = rk < room ? o_pos + rk : nbase + (rk - room);
                        if (q < out_cap) {
                            const uint32_t c16 = c > 65535u ? 65535u : c;
                            const uint32_t pay = IS_SET ? s_aux[i] : ((s_aux[i] & 0xffu) | (c16 << 8));
                            if (out.rec16) {
                                const uint64_t klo = KW == 2 ? s_key[2 * i] : s_key[i], khi = KW == 2 ? s_key[2 * i + 1] : 0ull;
                                out.rec16[q] = make_uint4((uint32_t)klo, (uint32_t)(klo >> 32), (uint32_t)khi, pay);
                            } else {
                                if (KW == 2) { out.hi[q] = s_key[2 * i + 1]; out.lo[q] = s_key[2 * i]; }
                                else out.lo[q] = s_key[i];
                                out.pay[q] = pay;
                            }
                        }
                    }
                    if (nv > room) { o_pos = nbase + (nv - room); o_end = nbase + WV_CHUNK; }
                    else o_pos += nv;
                }
                // next pass: the sibling of a first child, else the successor of the parent
                while (P > 1 && pr >= P / 2) { P >>= 1; pr -= P; }
                if (P == 1) break;
                pr += P / 2;
            } else {
                if (P >= 32768u) { fl_bad |= 8u; break; }         // (hash bits 16..30 select the pass)
                P <<= 1;                                                // first child (2P, r); its sibling (2P, r + P) follows it
                if (lane == 0) { atomicMax(&gflags[1], P); atomicAdd(&gflags[2], 1u); }
            }
        }
    }
#undef WV_LOAD_REC
    if (lane == 0) {
        holes[blockIdx.x] = WaveHole{o_pos, o_end - o_pos};
        if (fl_bad & 1u) atomicOr(&gflags[0], 1u);
        if (fl_bad & 8u) atomicOr(&gflags[0], 2u);
        if (fl_bad & 6u) atomicOr(&gflags[3], fl_bad & 6u);
    }
}

// ---- closing the holes -----------------------------------------------------------------------------------------------
// The launch reserved records [start, R) of the output in chunks; every wave left the unused tail of its last chunk: H hole
// slots in all, V = R - H valid records.  The valid records at positions >= V move into the hole slots below V (the output
// is unordered here, so any bijection will do).  marks covers the tail region [V, R): 1 = hole slot.
__global__ void __launch_bounds__(256) holes_mark_kernel(const WaveHole* __restrict__ holes, uint32_t nh, unsigned long long V, uint32_t* __restrict__ marks) {
    const uint32_t h = blockIdx.x;
    if (h >= nh) return;
    const unsigned long long a = holes[h].start, n = holes[h].len;
    for (unsigned long long i = threadIdx.x; i < n; i += blockDim.x) if (a + i >= V) marks[a + i - V] = 1u;
}
// dpre[h] = hole slots below V in holes 0..h-1 (exclusive scan done by one workgroup: a few thousand holes)
__global__ void __launch_bounds__(1024) holes_dest_scan_kernel(const WaveHole* __restrict__ holes, uint32_t nh, unsigned long long V, uint32_t* __restrict__ dpre) {
    __shared__ uint32_t s_wsum[16];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nh; base += 1024) {
        const uint32_t h = base + threadIdx.x;
        uint32_t d = 0;
        if (h < nh) {
            const unsigned long long a = holes[h].start, n = holes[h].len;
            d = a >= V ? 0u : (uint32_t)(a + n <= V ? n : V - a);
        }
        uint32_t tot;
        const uint32_t incl = block_inclusive_scan<1024>(d, s_wsum, &tot);
        if (h < nh) dpre[h] = carry + incl - d;
        carry += tot;
    }
    if (threadIdx.x == 0) dpre[nh] = carry;
}
template <bool REC16>
__global__ void __launch_bounds__(256) holes_move_kernel(const WaveHole* __restrict__ holes, uint32_t nh, unsigned long long V, uint32_t tail_len,
                                                         const uint32_t* __restrict__ marks, const uint32_t* __restrict__ mpre,
                                                         const uint32_t* __restrict__ dpre, FastOut out, int has_hi) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tail_len || marks[t]) return;
    const uint32_t j = t - mpre[t];                                     // rank among the valid records of the tail region
    uint32_t lo = 0, hi = nh;                                           // last hole h with dpre[h] <= j and a slot below V for j
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (dpre[mid] <= j) lo = mid; else hi = mid; }
    const unsigned long long dst = holes[lo].start + (j - dpre[lo]), src = V + t;
    if (REC16) out.rec16[dst] = out.rec16[src];
    else {
        if (has_hi) out.hi[dst] = out.hi[src];
        out.lo[dst] = out.lo[src];
        out.pay[dst] = out.pay[src];
    }
}
