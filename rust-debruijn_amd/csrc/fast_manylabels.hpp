// CountFilterSet with more than 64 distinct labels on the fast path (round 3; included by fastpath.hip).
//
// The counting kernel's colour sets hold 64 colours (two mask words per LDS table entry).  Larger alphabets used to take the
// generic path (sort every k-mer instance: ~4 Gkmer/s against ~60-90).  Here they are served by label GROUPS of 64:
//   1. the valid k-mers -- keys, Exts, validity by total count (filter.rs:85-100 counts observations over all labels) -- come
//      from one CountFilter run of the fast path over all reads: table T, ascending;
//   2. the reads are partitioned by label group (one counting pass + one scatter over the per-read metadata; the packed bases
//      stay where they are), and every group runs the fast path over ITS reads with min_kmer_obs = 1 and the WIDE colour layout:
//      table T_g (ascending keys of every k-mer seen in the group, 64-bit colour mask each);
//   3. T_g is joined into T (both ascending: a tile of T finds its window of T_g with two binary searches, then every key
//      searches that window), filling word g of the key's row of masks;
//   4. the label lists (CSR, ascending labels: filter.rs:96-98 sorts and de-duplicates) are written from the mask rows.
// No sort ever sees duplicate keys, and every step reuses the kernels of the 64-colour path.  Up to ML_MAX_GROUPS * 64 labels.
#pragma once

constexpr uint32_t ML_MAX_GROUPS = 16;
constexpr uint32_t ML_BLOCK_READS = 1024;

// per block of ML_BLOCK_READS reads: reads per label group -> blk_counts[g * nblk + block]
__global__ void __launch_bounds__(256) ml_group_count_kernel(const void* __restrict__ data, uint32_t width, uint64_t n, const uint8_t* __restrict__ lgroup,
                                                             uint32_t G, uint32_t* __restrict__ blk_counts, uint32_t nblk) {
    __shared__ uint32_t s_c[ML_MAX_GROUPS];
    if (threadIdx.x < ML_MAX_GROUPS) s_c[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * ML_BLOCK_READS;
    for (uint32_t j = threadIdx.x; j < ML_BLOCK_READS; j += 256) {
        const uint64_t i = base + j;
        if (i < n) {
            const uint32_t v = width == 1 ? ((const uint8_t*)data)[i] : (width == 2 ? ((const uint16_t*)data)[i] : ((const uint32_t*)data)[i]);
            atomicAdd(&s_c[lgroup[v]], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < G) blk_counts[(size_t)threadIdx.x * nblk + blockIdx.x] = s_c[threadIdx.x];
}

// the reads' metadata grouped by label group (any order inside a group); the label becomes the colour index inside its group
__global__ void __launch_bounds__(256) ml_group_scatter_kernel(const void* __restrict__ data, uint32_t width, uint64_t n, const uint8_t* __restrict__ lgroup,
                                                               const uint8_t* __restrict__ lcolour, const uint64_t* __restrict__ blk_off, uint32_t nblk,
                                                               const uint64_t* __restrict__ start, const uint32_t* __restrict__ length,
                                                               const uint8_t* __restrict__ exts, uint64_t* __restrict__ o_start,
                                                               uint32_t* __restrict__ o_length, uint8_t* __restrict__ o_exts, uint8_t* __restrict__ o_colour) {
    __shared__ uint32_t s_c[ML_MAX_GROUPS];
    if (threadIdx.x < ML_MAX_GROUPS) s_c[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * ML_BLOCK_READS;
    for (uint32_t j = threadIdx.x; j < ML_BLOCK_READS; j += 256) {
        const uint64_t i = base + j;
        if (i < n) {
            const uint32_t v = width == 1 ? ((const uint8_t*)data)[i] : (width == 2 ? ((const uint16_t*)data)[i] : ((const uint32_t*)data)[i]);
            const uint32_t g = lgroup[v];
            const uint64_t d = blk_off[(size_t)g * nblk + blockIdx.x] + atomicAdd(&s_c[g], 1u);
            o_start[d] = start[i];
            o_length[d] = length[i];
            if (exts) o_exts[d] = exts[i];
            o_colour[d] = lcolour[v];
        }
    }
}

__device__ __forceinline__ bool ml_less(uint64_t ah, uint64_t al, uint64_t bh, uint64_t bl) { return ah < bh || (ah == bh && al < bl); }

// T (a_*, ascending) x T_g (b_*, ascending): masks[i * G + g] = mask of T_g's entry with T's key i, if there is one
constexpr uint32_t ML_JOIN_TILE = 1024;
__global__ void __launch_bounds__(256) ml_join_kernel(const uint64_t* __restrict__ a_hi, const uint64_t* __restrict__ a_lo, uint32_t na,
                                                      const uint64_t* __restrict__ b_hi, const uint64_t* __restrict__ b_lo,
                                                      const unsigned long long* __restrict__ b_msk, uint32_t nb_,
                                                      unsigned long long* __restrict__ masks, uint32_t G, uint32_t g) {
    __shared__ uint32_t s_win[2];
    const uint32_t t0 = blockIdx.x * ML_JOIN_TILE;
    const uint32_t t1 = t0 + ML_JOIN_TILE < na ? t0 + ML_JOIN_TILE : na;
    if (threadIdx.x < 2) {
        // lower bound of the tile's first key (thread 0) / upper bound of its last key (thread 1) in T_g
        const uint32_t i = threadIdx.x == 0 ? t0 : t1 - 1;
        const uint64_t kh = a_hi[i], kl = a_lo[i];
        uint32_t lo = 0, hi = nb_;
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            const bool go_right = threadIdx.x == 0 ? ml_less(b_hi[mid], b_lo[mid], kh, kl) : !ml_less(kh, kl, b_hi[mid], b_lo[mid]);
            if (go_right) lo = mid + 1; else hi = mid;
        }
        s_win[threadIdx.x] = lo;
    }
    __syncthreads();
    const uint32_t w0 = s_win[0], w1 = s_win[1];
    for (uint32_t i = t0 + threadIdx.x; i < t1; i += 256) {
        const uint64_t kh = a_hi[i], kl = a_lo[i];
        uint32_t lo = w0, hi = w1;
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (ml_less(b_hi[mid], b_lo[mid], kh, kl)) lo = mid + 1; else hi = mid;
        }
        if (lo < w1 && b_hi[lo] == kh && b_lo[lo] == kl) masks[(size_t)i * G + g] = b_msk[lo];
    }
}

__global__ void __launch_bounds__(256) ml_setn_kernel(const unsigned long long* __restrict__ masks, uint32_t n, uint32_t G, uint32_t* __restrict__ setn) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t c = 0;
    for (uint32_t g = 0; g < G; g++) c += (uint32_t)__popcll(masks[(size_t)i * G + g]);
    setn[i] = c;
}
// labels of colour (g, bit) = labels[g * 64 + bit]: ascending in g and bit, so every list comes out sorted.  A wavefront takes 64
// keys whose lists are one contiguous stretch of set_val: every lane expands its key's masks into the wave's LDS staging area at
// the list's offset inside that stretch, then the wave copies the stretch out with consecutive lanes on consecutive words (a lane
// writing its own list straight to memory moved 9*10^9 labels at 0.9 TB/s: 41 ms at 100 labels).  Stretches longer than the staging
// area go out in pieces; the lanes then re-expand and keep only the entries of the current piece.
constexpr uint32_t ML_CSR_STAGE = 2048;
__global__ void __launch_bounds__(256) ml_csr_kernel(const unsigned long long* __restrict__ masks, uint32_t n, uint32_t G, const uint64_t* __restrict__ set_off,
                                                     const uint32_t* __restrict__ labels, uint32_t* __restrict__ set_val) {
    __shared__ uint32_t s_stage[4][ML_CSR_STAGE];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t k0 = (blockIdx.x * 4 + wave) * 64;
    if (k0 >= n) return;
    const uint32_t i = k0 + lane;
    const bool on = i < n;
    const uint64_t base = set_off[k0];
    const uint64_t end = set_off[k0 + 64 < n ? k0 + 64 : n];
    const uint64_t mine = on ? set_off[i] - base : 0;                 // my list's offset inside the wave's stretch
    uint32_t* st = s_stage[wave];
    for (uint64_t p0 = 0; p0 < end - base; p0 += ML_CSR_STAGE) {      // wave-uniform
        const uint64_t p1 = p0 + ML_CSR_STAGE < end - base ? p0 + ML_CSR_STAGE : end - base;
        if (on) {
            uint64_t o = mine;
            for (uint32_t g = 0; g < G && o < p1; g++) {
                unsigned long long m = masks[(size_t)i * G + g];
                const uint32_t c = (uint32_t)__popcll(m);
                if (o + c <= p0) { o += c; continue; }                // wholly before this piece
                while (m && o < p1) {
                    const uint32_t b = (uint32_t)__ffsll((long long)m) - 1u;
                    m &= m - 1;
                    if (o >= p0) st[o - p0] = labels[g * 64 + b];
                    o++;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (uint64_t q = p0 + lane; q < p1; q += 64) set_val[base + q] = st[q - p0];
        __builtin_amdgcn_wave_barrier();
    }
}

// *used = false (and nothing written): the rows of masks do not fit the device -- the caller takes the generic path
static int filter_kmers_fast_many(dbg_ctx* c, const SeqDev& s, const dbg_filter_params* prm, uint64_t n_kmers, const std::vector<uint32_t>& labels,
                                  dbg_kmer_table* out, bool* used) {
    *used = false;
    const int k = (int)prm->k;
    const bool stranded = prm->stranded != 0;
    const uint32_t nd = (uint32_t)labels.size(), G = (nd + 63) / 64;
    if (G > ML_MAX_GROUPS) return c->fail(137, "fast path: too many labels for the label-group passes");
    // Step 1 takes the valid k-mers from a CountFilter run, whose validity test reads the u16-saturated count: right for
    // min_kmer_obs <= 65535 only.  CountFilterSet::summarize compares the unsaturated nobs (filter.rs:85-100), so larger
    // thresholds take the generic path (which counts in 64 bits).
    if (prm->min_kmer_obs > 65535) return 0;
    if (c->opt("DBG_DEBUG")) fprintf(stderr, "[fastpath] %u distinct labels: %u label groups\n", nd, G);

    // 1. the valid k-mers: CountFilter over all reads
    dbg_kmer_table T;
    memset(&T, 0, sizeof(T));
    {
        FastPlan pl;
        if (!fast_make_plan(c, k, stranded, false, n_kmers, 0, &pl)) return c->fail(137, "fast path: unsupported k for the label-group passes");
        DBG_TRY(fast_run(c, s, pl, prm->min_kmer_obs, prm->report_all_kmers != 0, n_kmers, &T));
    }
    struct TableGuard { dbg_ctx* c; dbg_kmer_table* t; ~TableGuard() { dbg_free_table(c, t); } } guard_T{c, &T};
    const uint32_t nv = (uint32_t)T.n;

    // 2. reads by label group
    const uint32_t top = labels.back();
    std::vector<uint8_t> h_group((size_t)top + 1, 0), h_colour((size_t)top + 1, 0);
    for (uint32_t i = 0; i < nd; i++) { h_group[labels[i]] = (uint8_t)(i / 64); h_colour[labels[i]] = (uint8_t)(i % 64); }
    DBuf<uint8_t> d_group, d_colour, p_exts, p_colour;
    DBuf<uint32_t> d_labels, blk_counts, p_length;
    DBuf<uint64_t> blk_off, p_start;
    const uint32_t nblk = cdiv(std::max<uint64_t>(s.n, 1), ML_BLOCK_READS);
    ALLOC_OR_FAIL(c, d_group, (size_t)top + 1); ALLOC_OR_FAIL(c, d_colour, (size_t)top + 1); ALLOC_OR_FAIL(c, d_labels, (size_t)G * 64);
    ALLOC_OR_FAIL(c, blk_counts, (size_t)G * nblk); ALLOC_OR_FAIL(c, blk_off, (size_t)G * nblk + 1);
    ALLOC_OR_FAIL(c, p_start, s.n); ALLOC_OR_FAIL(c, p_length, s.n); ALLOC_OR_FAIL(c, p_colour, s.n);
    if (s.exts) ALLOC_OR_FAIL(c, p_exts, s.n);
    std::vector<uint32_t> h_labels((size_t)G * 64, 0);
    std::copy(labels.begin(), labels.end(), h_labels.begin());
    HIP_TRY(c, hipMemcpyAsync(d_group.p, h_group.data(), h_group.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d_colour.p, h_colour.data(), h_colour.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d_labels.p, h_labels.data(), h_labels.size() * 4, hipMemcpyHostToDevice, c->stream));
    c->t_begin("label_groups", s.n);
    ml_group_count_kernel<<<nblk, 256, 0, c->stream>>>(s.data, s.data_width, s.n, d_group.p, G, blk_counts.p, nblk);
    c->t_end();
    LAUNCH_CHECK(c, "ml_group_count");
    DBG_TRY(scan_exclusive_u32_u64(c, blk_counts.p, blk_off.p, (uint64_t)G * nblk));
    c->t_begin("label_groups", 0);
    ml_group_scatter_kernel<<<nblk, 256, 0, c->stream>>>(s.data, s.data_width, s.n, d_group.p, d_colour.p, blk_off.p, nblk, s.start, s.length, s.exts,
                                                        p_start.p, p_length.p, p_exts.p, p_colour.p);
    c->t_end();
    LAUNCH_CHECK(c, "ml_group_scatter");
    std::vector<uint64_t> g_off(G + 1, 0);
    for (uint32_t g = 0; g < G; g++)
        HIP_TRY(c, hipMemcpyAsync(&g_off[g], blk_off.p + (size_t)g * nblk, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));      // (also: the host tables above leave scope only after their copies)
    g_off[G] = s.n;

    // 3. one 64-colour run per group, joined into the rows of masks
    DBuf<unsigned long long> masks;
    if (!masks.alloc(c, std::max<size_t>((size_t)nv * G, 1))) return 0;       // (T is released by its guard)
    HIP_TRY(c, hipMemsetAsync(masks.p, 0, std::max<size_t>((size_t)nv * G, 1) * 8, c->stream));
    for (uint32_t g = 0; g < G && nv; g++) {
        const uint64_t a = g_off[g], n_g = g_off[g + 1] - a;
        if (!n_g) continue;
        SeqDev sg{s.words, p_start.p + a, p_length.p + a, s.exts ? p_exts.p + a : nullptr, p_colour.p + a, 1u, n_g, s.n_words};
        uint64_t nk_g = 0;
        DBG_TRY(kmer_total(c, sg, k, &nk_g));
        if (!nk_g) continue;
        FastPlan pl;
        if (!fast_make_plan(c, k, stranded, true, nk_g, 0, &pl)) return c->fail(137, "fast path: unsupported k for the label-group passes");
        pl.wide = true;                               // (colours are their own indices: no label map inside a group)
        // Bins are sized by k-mer INSTANCES for 30x reads, where a bin's 8 000 instances hold ~320 distinct k-mers.  A group sees
        // 1/G of the coverage and keeps its singletons (min_kmer_obs = 1), so its instances hold far more distinct k-mers: size
        // the bins for ~350 distinct ones in the 1024-entry tables of the WIDE layout (with 4 000 instances per bin the tables
        // of a 7x group overflowed and bins were re-streamed in passes: 2.2 s of bin_count at 250 labels instead of 0.3).
        if (!c->opt("DBG_FAST_TARGET")) {
            const double distinct = std::min<double>((double)nk_g, (double)nv + 0.06 * (double)nk_g);
            const double per_bin = std::max(512.0, std::min(4000.0, 350.0 * (double)nk_g / std::max(distinct, 1.0)));
            pl.nbins = (uint32_t)std::min<double>(std::max(1.0, (double)nk_g / per_bin), (double)((1u << 23) - 1));
        }
        dbg_kmer_table Tg;
        memset(&Tg, 0, sizeof(Tg));
        DBuf<unsigned long long> msk;
        const int rc = fast_run(c, sg, pl, 1, false, nk_g, &Tg, &msk);
        TableGuard guard_g{c, &Tg};
        DBG_TRY(rc);
        if (Tg.n) {
            c->t_begin("label_join", nv);
            ml_join_kernel<<<cdiv(nv, ML_JOIN_TILE), 256, 0, c->stream>>>(T.key_hi, T.key_lo, nv, Tg.key_hi, Tg.key_lo, msk.p, (uint32_t)Tg.n, masks.p, G, g);
            c->t_end();
            LAUNCH_CHECK(c, "ml_join");
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream));  // T_g and its masks are released at the end of this iteration
    }

    // 4. label lists
    DBuf<uint32_t> setn, set_val;
    DBuf<uint64_t> set_off;
    ALLOC_OR_FAIL(c, setn, std::max<uint32_t>(nv, 1)); ALLOC_OR_FAIL(c, set_off, (size_t)nv + 1);
    uint64_t n_setval = 0;
    if (nv) {
        ml_setn_kernel<<<cdiv(nv, 256), 256, 0, c->stream>>>(masks.p, nv, G, setn.p);
        LAUNCH_CHECK(c, "ml_setn");
        DBG_TRY(scan_exclusive_u32_u64(c, setn.p, set_off.p, nv));
        HIP_TRY(c, hipMemcpyAsync(&n_setval, set_off.p + nv, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    } else HIP_TRY(c, hipMemsetAsync(set_off.p, 0, 8, c->stream));
    ALLOC_OR_FAIL(c, set_val, std::max<uint64_t>(n_setval, 1));
    if (nv) {
        c->t_begin("set_csr", nv);
        ml_csr_kernel<<<cdiv(nv, 256), 256, 0, c->stream>>>(masks.p, nv, G, set_off.p, d_labels.p, set_val.p);   // a wave per 64 keys
        c->t_end();
        LAUNCH_CHECK(c, "ml_csr");
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *out = T;
    memset(&T, 0, sizeof(T));                         // ownership of T's arrays moves to *out
    if (out->count) { c->dfree(out->count); out->count = nullptr; }
    out->set_off = set_off.take(); out->set_val = set_val.take(); out->n_set_val = n_setval;
    out->n_passes = 1 + G;
    *used = true;
    return 0;
}
