// The rank-spanning counting flow behind the C ABI (round 4): dbg_shard_filter_kmers_dev.  Included at the end of fastpath.hip.
//
// filter_kmers over the reads of all ranks (src/filter.rs:139-231 on the union; the reference scales out by MSP shards + one
// filter_kmers per shard, src/test.rs:433-456, src/msp.rs:279-324).  A shard here is the contiguous range of minimizer bins a
// rank owns.  What the Python orchestration of rounds 2-3 did with torch.distributed (all-reduces, send_layout,
// exchange_geometry, the < 1 GiB rule, the round pipeline) is done here, in the library, over a dbg_transport:
//
//   scan (+ sender-side merge)            ctx stream
//   record histogram all-reduced          transport, ctx stream            -> ownership bounds (greedy, host)
//   offsets in (round, destination, bin)  layout kernels, ctx stream
//   slab compaction in that order         ctx stream                       -> ONE send buffer, every message a contiguous range
//   per-bin counts to the owners          all_to_allv, ctx stream
//   round c + 1 of the records            all_to_allv, communication stream  | overlapped with
//   counting of round c                   bin_count, ctx stream              | (ordered by events, no host wait per round)
//   sort                                  ctx stream
//
// A rank's own records never travel: the counting kernel reads them from the send buffer (segment 0) and the received ones
// from the round's receive buffer (segments 1 .. W-1).
#pragma once
#include <chrono>
#include "shard_comm.hpp"

namespace {

// layout position of every bin: ranges (round c, destination d) in that order, bins ascending inside a range
struct LayoutTab {
    uint32_t world, n_rounds;
    uint32_t bounds[65];            // [world + 1]
};

__global__ void __launch_bounds__(256) layout_pos_kernel(LayoutTab lt, const uint32_t* __restrict__ cuts /* [world][n_rounds + 1] */,
                                                          const uint32_t* __restrict__ range_base /* [n_rounds * world] bins before the range */,
                                                          const uint32_t* __restrict__ count, uint32_t nb, uint32_t* __restrict__ pos_of,
                                                          uint32_t* __restrict__ perm_count) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    uint32_t d = 0;
    while (d + 1 < lt.world && b >= lt.bounds[d + 1]) d++;
    const uint32_t rel = b - lt.bounds[d];
    const uint32_t* cd = cuts + (size_t)d * (lt.n_rounds + 1);
    uint32_t lo = 0, hi = lt.n_rounds;                       // largest c with cd[c] <= rel (cd[n_rounds] > rel)
    while (hi - lo > 1) { const uint32_t m = (lo + hi) >> 1; if (cd[m] <= rel) lo = m; else hi = m; }
    // (empty ranges repeat a cut value: take the last range that starts at or before rel)
    const uint32_t pos = range_base[lo * lt.world + d] + (rel - cd[lo]);
    pos_of[b] = pos;
    perm_count[pos] = count[b];
}

__global__ void __launch_bounds__(256) layout_off_kernel(const uint32_t* __restrict__ pos_of, const uint64_t* __restrict__ csum, uint32_t nb,
                                                          uint64_t* __restrict__ off) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nb) off[b] = csum[pos_of[b]];
}

__global__ void __launch_bounds__(256) gather_u64_kernel(const uint64_t* __restrict__ src, const uint64_t* __restrict__ idx, uint32_t n,
                                                          uint64_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}

// where a bin's overflow records (those that did not fit its slab) go: behind the slab's records in the bin's place of the send buffer
__global__ void __launch_bounds__(256) ovf_base_kernel(const uint32_t* __restrict__ cursor, uint32_t slab_cap, const uint64_t* __restrict__ off, uint32_t nb,
                                                        uint64_t* __restrict__ ovf_base) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nb) { const uint32_t cnt = cursor[b]; ovf_base[b] = off[b] + (cnt < slab_cap ? cnt : slab_cap); }
}

// records per coarse group of f consecutive bins (the last group may be shorter)
__global__ void __launch_bounds__(256) coarse_hist_kernel(const uint32_t* __restrict__ count, uint32_t nb, uint32_t f, uint32_t ng, uint64_t* __restrict__ out) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ng) return;
    uint64_t s = 0;
    const uint32_t lo = g * f, hi = lo + f < nb ? lo + f : nb;
    for (uint32_t b = lo; b < hi; b++) s += count[b];
    out[g] = s;
}

// Segment tables of all rounds from the scanned per-source counts.  G[s * nbl + j] = records of source s (in source order: self
// first) before owned bin j, over the whole owned range (G has W * nbl + 1 entries).  Round c covers owned bins [cut[c], cut[c+1]);
// its table is seg_c[s][0 .. nbc] with seg_c[s][j - cut[c]] = base[c * W + s] + G[s * nbl + j] - G[s * nbl + cut[c]].
__global__ void __launch_bounds__(256) seg_tables_kernel(const uint64_t* __restrict__ G, uint32_t nbl, uint32_t W, uint32_t n_rounds,
                                                          const uint32_t* __restrict__ cut /* [n_rounds + 1] */,
                                                          const uint64_t* __restrict__ base /* [n_rounds * W] */,
                                                          const uint64_t* __restrict__ tab_off /* [n_rounds] start of round c's table in seg */,
                                                          uint64_t* __restrict__ seg) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)W * (nbl + 1)) return;
    const uint32_t s = (uint32_t)(t / (nbl + 1)), j = (uint32_t)(t % (nbl + 1));
    const uint64_t g = G[(uint64_t)s * nbl + j] ;
    uint32_t lo = 0, hi = n_rounds;                          // largest c with cut[c] <= j
    while (hi - lo > 1) { const uint32_t m = (lo + hi) >> 1; if (cut[m] <= j) lo = m; else hi = m; }
    // j closes round c - 1 (as its last column) for every round that ends at j and opens / lies inside every round with cut[c] <= j < cut[c + 1]
    for (int32_t c = (int32_t)lo; c >= 0; c--) {
        if (j > cut[c + 1]) break;
        if (j < cut[c]) continue;
        const uint32_t nbc = cut[c + 1] - cut[c];
        seg[tab_off[c] + (uint64_t)s * (nbc + 1) + (j - cut[c])] = base[(uint64_t)c * W + s] + g - G[(uint64_t)s * nbl + cut[c]];
    }
}

struct XTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

// events of one call, returned to the ctx's pool however the call ends
struct EventSet {
    dbg_ctx* c;
    std::vector<hipEvent_t> ev;
    explicit EventSet(dbg_ctx* c_) : c(c_) {}
    hipEvent_t get() { ev.push_back(c->get_event()); return ev.back(); }
    ~EventSet() { for (hipEvent_t e : ev) c->event_pool.push_back(e); }
};

}  // namespace

namespace {
// what one exchange + count run needs beyond the plan
struct XRun {
    uint32_t n_rounds = 0;
    bool balance = true;
    int merge = 0;
    uint64_t n_local = 0, n_max = 0, total = 0;
    std::vector<uint32_t>* bounds_io = nullptr;         // empty: ownership is derived here (and returned); filled: used as it is
    DBuf<unsigned long long>* keep_masks = nullptr;     // label-group runs
    int lrc_in = 0;                                     // status of the rank-local work done before the run
    bool main_run = true;                               // the call's own table (timing records cleared, merge statistics kept)
    bool lists = false;                                 // CountFilterSet over a large label alphabet: label lists (fast_labellists.hpp) -- the records
                                                        // carry their read's label in an extra word, the owners run bin_labels_kernel round by round
};

// scan -> ownership -> layout -> pipelined exchange rounds -> count -> sort, over the plan sp (see the header of this file)
int shard_exchange_run(dbg_ctx* c, ShardComm& X, const dbg_transport* tr, const dbg_seqset* ds, dbg_shard_plan sp, const XRun& o,
                       dbg_kmer_table* out, dbg_shard_stats* S, const XTimer& t_setup) {
    const uint32_t W = X.W, me = X.me;
    const uint64_t n_local = o.n_local, n_max = o.n_max;
    int lrc = 0;
    sp.merge_dups = o.merge ? 1u : 0u;
    // ---- phase "scan": plan, scan (+ merge), record histogram of the coarse bin groups ----
    FastPlan pl;
    FastScan sc;
    uint32_t nb = 0, rw = 0, f = 1, ng = 0;
    const bool given = o.bounds_io && !o.bounds_io->empty();       // ownership handed in (label-group runs: the bounds of the first run)
    const bool balance = o.balance && W > 1 && !given;
    DBuf<uint64_t> gh;
    if (o.main_run) c->t_clear();
    lrc = [&]() -> int {
        if (o.lrc_in) return o.lrc_in;                              // (rank-local work before this run failed: c->err holds its message)
        if (o.lists) {                                              // (no colour layout to agree on: labels travel as they are)
            dbg_shard_plan spp = sp;
            spp.max_label = 0; spp.n_labels = 0;
            DBG_TRY(plan_from(c, &spp, &pl));
            lists_plan(&pl);
        } else DBG_TRY(plan_from(c, &sp, &pl));
        nb = pl.nbins * NCLS; rw = (uint32_t)pl.rw;
        if (X.inject("scan")) return X.injected("scan");
        DBG_TRY(shard_scan_core(c, ds, &sp, pl, &sc));
        if (balance) {
            // Records per COARSE group of f consecutive bins (f a power of two that leaves every rank >= 4096 groups to be cut from): the
            // all-reduce and the host's greedy cut then handle 10^4..10^5 values instead of one per bin -- at eight ranks of 10^8 reads
            // the per-bin form was 67 MB through pageable memory and two host loops over 8*10^6 entries, every step.  Ownership boundaries
            // become multiples of f bins: a granularity of f / (bins per rank) <= 1/4096 of a rank's share.
            while ((uint64_t)f * 2 * 4096 * W <= nb) f *= 2;
            ng = (nb + f - 1) / f;
            if (X.inject("hist")) return X.injected("hist");
            ALLOC_OR_FAIL(c, gh, ng);
            coarse_hist_kernel<<<cdiv(ng, 256), 256, 0, c->stream>>>(sc.cursor.p, nb, f, ng, gh.p);
            LAUNCH_CHECK(c, "coarse_hist");
        }
        return 0;
    }();
    DBG_TRY(X.agree(lrc, "scan"));
    sp.n_bins = nb; sp.rec_words = rw; sp.bin_group = NCLS;
    S->n_bins = nb; S->merge_dups = o.merge;
    hipStream_t xs = c->get_comm_stream();

    // ---- ownership ----
    std::vector<uint32_t> bounds(W + 1);
    if (balance) {
        if (tr->all_reduce_u64(tr->self, gh.p, ng, 0, c->stream)) return X.op_failed("all_reduce_u64 (record histogram)");
        std::vector<uint64_t> hh(ng);
        if (hipMemcpyAsync(hh.data(), gh.p, (size_t)ng * 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { (void)hipGetLastError(); return X.op_failed("copy of the record histogram"); }
        DBG_TRY(X.wait_stream(c->stream, "the all-reduced record histogram"));
        static_assert(NCLS == 1, "a coarse group is f whole bin groups");
        // (pure host arithmetic on values every rank holds alike: the ranks succeed or fail together)
        if (dbg_shard_owner_bounds(hh.data(), ng, 1, W, bounds.data())) return c->fail(162, "sharded flow: ownership bounds");
        for (uint32_t r = 0; r <= W; r++) bounds[r] = (uint32_t)std::min<uint64_t>((uint64_t)bounds[r] * f, nb);
        bounds[W] = nb;
        gh.release();
    } else if (given) {
        if (o.bounds_io->size() != W + 1 || (*o.bounds_io)[W] != nb) return c->fail(162, "sharded flow: ownership bounds of another bin space");
        bounds = *o.bounds_io;
    } else if (dbg_shard_owner_bounds(nullptr, nb, NCLS, W, bounds.data())) return c->fail(162, "sharded flow: ownership bounds");
    if (o.bounds_io) *o.bounds_io = bounds;
    S->balanced = balance ? 1 : 0;
    S->owned_lo = bounds[me]; S->owned_hi = bounds[me + 1];

    // ---- rounds ----
    uint32_t n_rounds = o.n_rounds;
    if (!n_rounds) {
        // Rounds are the unit of overlap (round c + 1 travels while round c is counted) and of receive-buffer memory (two buffers of
        // one round's incoming records); the transport itself keeps every message under 1 GiB.  Four rounds leave a quarter of the
        // exchange exposed, eight an eighth -- more only when a round's receive buffer would pass 8 GiB.  Records are at most ~2.5
        // bytes per k-mer instance (24-byte records of >= 10 k-mers at k = 47; denser for small k); every rank must arrive at the
        // same number, hence the estimate from the largest rank.
        const uint64_t est_in = (uint64_t)((double)std::max<uint64_t>(n_max, 1) * 2.5 * (double)(W - 1) / (double)W);
        n_rounds = (uint32_t)std::max<uint64_t>(W < 4 ? 4 : 8, (est_in + (8ull << 30) - 1) >> 33);
    }
    n_rounds = std::min<uint32_t>(n_rounds, 64);
    const uint32_t stride = n_rounds + 1;
    std::vector<uint32_t> cuts((size_t)W * stride);
    if (dbg_shard_round_cuts(bounds.data(), W, NCLS, &n_rounds, cuts.data())) return c->fail(162, "sharded flow: round cuts");
    const uint32_t R = n_rounds;
    S->n_rounds = R;
    // compact [W][R + 1]
    std::vector<uint32_t> cutc((size_t)W * (R + 1));
    for (uint32_t d = 0; d < W; d++) for (uint32_t r = 0; r <= R; r++) cutc[(size_t)d * (R + 1) + r] = cuts[(size_t)d * stride + r];
    const uint32_t* mycut = &cutc[(size_t)me * (R + 1)];
    const uint32_t nbl = bounds[me + 1] - bounds[me];

    // ---- phase "layout": offsets of every bin in (round, destination, bin) order; room for the per-bin counts of my bins ----
    std::vector<uint32_t> range_base((size_t)R * W);
    {
        uint32_t acc = 0;
        for (uint32_t r = 0; r < R; r++) for (uint32_t d = 0; d < W; d++) { range_base[(size_t)r * W + d] = acc; acc += cutc[(size_t)d * (R + 1) + r + 1] - cutc[(size_t)d * (R + 1) + r]; }
    }
    DBuf<uint32_t> d_cuts, d_rbase, pos_of, perm_count, rhist;
    DBuf<uint64_t> csum, off, d_idx, d_edge;
    const uint32_t n_edges = R * W + 1;
    std::vector<uint64_t> eidx(n_edges), edge(n_edges, 0);
    // source order of the counting side: self first, then the other ranks ascending
    std::vector<uint32_t> src_rank(W), row_of(W);
    src_rank[0] = me;
    for (uint32_t r = 0, i = 1; r < W; r++) if (r != me) src_rank[i++] = r;
    for (uint32_t i = 0; i < W; i++) row_of[src_rank[i]] = i;
    lrc = [&]() -> int {
        if (!xs) return c->fail(100, "sharded flow: no communication stream");
        if (X.inject("layout")) return X.injected("layout");
        ALLOC_OR_FAIL(c, d_cuts, cutc.size()); ALLOC_OR_FAIL(c, d_rbase, range_base.size());
        ALLOC_OR_FAIL(c, pos_of, std::max(nb, 1u)); ALLOC_OR_FAIL(c, perm_count, std::max(nb, 1u));
        ALLOC_OR_FAIL(c, csum, (size_t)nb + 1); ALLOC_OR_FAIL(c, off, (size_t)nb + 1);
        HIP_TRY(c, hipMemcpyAsync(d_cuts.p, cutc.data(), cutc.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(d_rbase.p, range_base.data(), range_base.size() * 4, hipMemcpyHostToDevice, c->stream));
        LayoutTab lt;
        lt.world = W; lt.n_rounds = R;
        for (uint32_t d = 0; d <= W; d++) lt.bounds[d] = bounds[d];
        layout_pos_kernel<<<cdiv(nb, 256), 256, 0, c->stream>>>(lt, d_cuts.p, d_rbase.p, sc.cursor.p, nb, pos_of.p, perm_count.p);
        LAUNCH_CHECK(c, "layout_pos");
        DBG_TRY(scan_exclusive_u32_u64(c, perm_count.p, csum.p, nb));
        layout_off_kernel<<<cdiv(nb, 256), 256, 0, c->stream>>>(pos_of.p, csum.p, nb, off.p);
        LAUNCH_CHECK(c, "layout_off");
        // edges of the (round, destination) blocks
        for (uint32_t i = 0; i + 1 < n_edges; i++) eidx[i] = range_base[i];
        eidx[n_edges - 1] = nb;
        ALLOC_OR_FAIL(c, d_idx, n_edges); ALLOC_OR_FAIL(c, d_edge, n_edges);
        HIP_TRY(c, hipMemcpyAsync(d_idx.p, eidx.data(), (size_t)n_edges * 8, hipMemcpyHostToDevice, c->stream));
        gather_u64_kernel<<<cdiv(n_edges, 256), 256, 0, c->stream>>>(csum.p, d_idx.p, n_edges, d_edge.p);
        LAUNCH_CHECK(c, "gather_u64");
        HIP_TRY(c, hipMemcpyAsync(edge.data(), d_edge.p, (size_t)n_edges * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (X.inject("counts")) return X.injected("counts");
        ALLOC_OR_FAIL(c, rhist, std::max<size_t>((size_t)W * nbl, 1));
        if (nbl) HIP_TRY(c, hipMemcpyAsync(rhist.p, sc.cursor.p + bounds[me], (size_t)nbl * 4, hipMemcpyDeviceToDevice, c->stream));
        return 0;
    }();
    DBG_TRY(X.agree(lrc, "layout"));
    const uint64_t n_recs = edge[n_edges - 1];
    S->records_scanned = n_recs;
    c->t_begin("sk_records", n_recs);    // bookkeeping entry: units = super-k-mer records (no kernel)
    c->t_end();

    // ---- per-bin counts of my bins from every source ----
    if (X.live) {
        std::vector<uint64_t> soff(W), sby(W), roff(W), rby(W);
        for (uint32_t d = 0; d < W; d++) {
            soff[d] = (uint64_t)bounds[d] * 4; sby[d] = d == me ? 0 : (uint64_t)(bounds[d + 1] - bounds[d]) * 4;
            roff[d] = (uint64_t)row_of[d] * nbl * 4; rby[d] = d == me ? 0 : (uint64_t)nbl * 4;
        }
        if (tr->all_to_allv(tr->self, sc.cursor.p, soff.data(), sby.data(), rhist.p, roff.data(), rby.data(), c->stream))
            return X.op_failed("all_to_allv (per-bin counts)");
        DBG_TRY(X.wait_stream(c->stream, "the per-bin record counts of the other ranks"));
    }

    // ---- phase "tables": segment tables of all rounds, the send buffer with round 0 compacted, receive buffers, counting state.
    //      Everything the rounds need is reserved HERE, so that no rank can run out of memory between two rounds' messages. ----
    std::vector<uint64_t> cnt((size_t)R * W), base((size_t)R * W), tab_off(R), recv_recs(R);
    uint64_t max_recv = 0, owned = 0;
    DBuf<uint64_t> seg, recs, ovf_base, rbuf[2];
    EventSet events(c);
    std::vector<hipEvent_t> ev_done(R), ev_a(R), ev_b(R), ev_w0(R), ev_w1(R), ev_counted(R);
    std::unique_ptr<FastCountState> cs(new FastCountState());
    std::unique_ptr<ListCountState> ls(new ListCountState());
    auto compact_round = [&](uint32_t r, hipStream_t stream) -> int {
        for (uint32_t d = 0; d < W; d++) {
            const uint32_t b0 = bounds[d] + cutc[(size_t)d * (R + 1) + r], b1 = bounds[d] + cutc[(size_t)d * (R + 1) + r + 1];
            if (b1 <= b0) continue;
            const uint32_t nbr = b1 - b0, blocks = cdiv((uint64_t)nbr * 64, 256);
            const uint64_t* slab_r = sc.slab.p + (uint64_t)b0 * sc.slab_cap * rw;
#define COMPACT(RW_) slab_compact_kernel<RW_><<<blocks, 256, 0, stream>>>(slab_r, sc.slab_cap, sc.cursor.p + b0, off.p + b0, nbr, recs.p, ovf_base.p + b0)
            if (rw == 2) COMPACT(2); else if (rw == 3) COMPACT(3); else if (rw == 4) COMPACT(4); else COMPACT(5);
#undef COMPACT
            LAUNCH_CHECK(c, "slab_compact");
        }
        return 0;
    };
    lrc = [&]() -> int {
        if (X.inject("tables")) return X.injected("tables");
        DBuf<uint64_t> G;
        ALLOC_OR_FAIL(c, G, (size_t)W * nbl + 1);
        DBG_TRY(scan_exclusive_u32_u64(c, rhist.p, G.p, (uint64_t)W * nbl));
        // G at the round cuts of every source -> per-round record counts (host)
        std::vector<uint64_t> gidx((size_t)W * (R + 1)), gval((size_t)W * (R + 1));
        for (uint32_t s = 0; s < W; s++) for (uint32_t r = 0; r <= R; r++) gidx[(size_t)s * (R + 1) + r] = (uint64_t)s * nbl + mycut[r];
        DBuf<uint64_t> d_gidx, d_gval;
        ALLOC_OR_FAIL(c, d_gidx, gidx.size()); ALLOC_OR_FAIL(c, d_gval, gidx.size());
        HIP_TRY(c, hipMemcpyAsync(d_gidx.p, gidx.data(), gidx.size() * 8, hipMemcpyHostToDevice, c->stream));
        gather_u64_kernel<<<cdiv(gidx.size(), 256), 256, 0, c->stream>>>(G.p, d_gidx.p, (uint32_t)gidx.size(), d_gval.p);
        LAUNCH_CHECK(c, "gather_u64");
        HIP_TRY(c, hipMemcpyAsync(gval.data(), d_gval.p, gidx.size() * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        // cnt[r][s] = records of source row s in round r; base: row 0 (self) = absolute offset of block (r, me) in the send buffer,
        // rows >= 1 = offsets in the round's receive buffer
        uint64_t seg_words = 0;
        for (uint32_t r = 0; r < R; r++) {
            uint64_t acc = 0;
            for (uint32_t s = 0; s < W; s++) {
                cnt[(size_t)r * W + s] = gval[(size_t)s * (R + 1) + r + 1] - gval[(size_t)s * (R + 1) + r];
                owned += cnt[(size_t)r * W + s];
                if (s == 0) base[(size_t)r * W] = edge[(size_t)r * W + me];
                else { base[(size_t)r * W + s] = acc; acc += cnt[(size_t)r * W + s]; }
            }
            if (cnt[(size_t)r * W] != edge[(size_t)r * W + me + 1] - edge[(size_t)r * W + me])
                return c->fail(163, "sharded flow: layout and histogram disagree about this rank's own records");
            recv_recs[r] = acc;
            max_recv = std::max(max_recv, acc);
            tab_off[r] = seg_words;
            seg_words += (uint64_t)W * (mycut[r + 1] - mycut[r] + 1);
        }
        DBuf<uint64_t> d_base, d_taboff;
        DBuf<uint32_t> d_mycut;
        ALLOC_OR_FAIL(c, seg, std::max<uint64_t>(seg_words, 1)); ALLOC_OR_FAIL(c, d_base, base.size()); ALLOC_OR_FAIL(c, d_taboff, R);
        ALLOC_OR_FAIL(c, d_mycut, R + 1);
        HIP_TRY(c, hipMemcpyAsync(d_base.p, base.data(), base.size() * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(d_taboff.p, tab_off.data(), (size_t)R * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(d_mycut.p, mycut, (size_t)(R + 1) * 4, hipMemcpyHostToDevice, c->stream));
        seg_tables_kernel<<<cdiv((uint64_t)W * (nbl + 1), 256), 256, 0, c->stream>>>(G.p, nbl, W, R, d_mycut.p, d_base.p, d_taboff.p, seg.p);
        LAUNCH_CHECK(c, "seg_tables");
        // ---- send buffer: slabs compacted in layout order, ROUND BY ROUND.  Only round 0 has to be in place before the first message
        //      leaves (the first round is the one whose exchange nothing hides: it starts 5 ms earlier per 10^8 reads this way); the
        //      compaction of round r >= 1 is queued on the communication stream right in front of that round's all-to-all and runs next
        //      to the counting of round r - 1.  Measured on one GPU: the overlap itself buys nothing -- the counting kernel loses the
        //      1.65 ms the copy takes (13.3 -> 14.9 ms per round; 119.1 against 119.6 ms per step) -- it only moves the work off the
        //      path in front of the first message.  The few records that outgrew their slab are placed behind their bins first
        //      (their positions follow from the counts alone). ----
        if (X.inject("recs")) return X.injected("recs");
        ALLOC_OR_FAIL(c, recs, std::max<uint64_t>(n_recs * rw, 1));
        ALLOC_OR_FAIL(c, ovf_base, (size_t)nb + 1);
        ovf_base_kernel<<<cdiv(nb, 256), 256, 0, c->stream>>>(sc.cursor.p, sc.slab_cap, off.p, nb, ovf_base.p);
        LAUNCH_CHECK(c, "ovf_base");
        c->t_begin("slab_compact", n_recs / R);
        DBG_TRY(compact_round(0, c->stream));
        c->t_end();
        DBG_TRY(fast_scatter(c, &sc, ovf_base.p, recs.p));                // (releases the scan's read-order buffers)
        HIP_TRY(c, hipStreamSynchronize(c->stream));                     // everything queued so far is complete: layout, tables, round 0
        pos_of.release(); perm_count.release(); csum.release(); G.release(); rhist.release();
        if (X.inject("rbuf")) return X.injected("rbuf");
        ALLOC_OR_FAIL(c, rbuf[0], std::max<uint64_t>(max_recv * rw, 1));
        if (R > 1) ALLOC_OR_FAIL(c, rbuf[1], std::max<uint64_t>(max_recv * rw, 1));
        for (uint32_t r = 0; r < R; r++) { ev_done[r] = events.get(); ev_a[r] = events.get(); ev_b[r] = events.get(); ev_w0[r] = events.get(); ev_w1[r] = events.get(); ev_counted[r] = events.get(); }
        if (X.inject("count_begin")) return X.injected("count_begin");
        if (pl.lists) {
            // room for one label per k-mer instance this rank will count: its records' share of the instances it scanned, with a margin
            // (lists_count_bins grows the buffer if a round wants more)
            const double per_rec = n_recs ? (double)std::max<uint64_t>(n_local, 1) / (double)n_recs : 16.0;
            const uint64_t lab_cap = (uint64_t)((double)owned * per_rec * 1.15) + 65536;
            DBG_TRY(lists_count_begin(c, pl, sp.min_kmer_obs, std::max<uint64_t>(n_local, 1), lab_cap, false, ls.get()));
            return 0;
        }
        DBG_TRY(fast_count_begin(c, pl, sp.min_kmer_obs, std::max<uint64_t>(n_local, 1), cs.get()));
        cs->keep_masks = o.keep_masks;                              // (label-group runs: the sorted 64-bit masks instead of a CSR)
        return 0;
    }();
    DBG_TRY(X.agree(lrc, "tables"));
    S->records_owned = owned;

    // ---- pipelined rounds ----
    std::vector<double> host_ms(R, 0.0), wait_ms(R, 0.0);
    int tr_rc = 0;                     // a transport failure: nothing left to agree on (the communicator is aborted)
    auto launch = [&](uint32_t r) -> int {
        std::vector<uint64_t> soff(W), sby(W), roff(W), rby(W);
        for (uint32_t d = 0; d < W; d++) {
            soff[d] = edge[(size_t)r * W + d] * rw * 8;
            sby[d] = d == me ? 0 : (edge[(size_t)r * W + d + 1] - edge[(size_t)r * W + d]) * rw * 8;
            roff[d] = d == me ? 0 : base[(size_t)r * W + row_of[d]] * rw * 8;
            rby[d] = d == me ? 0 : cnt[(size_t)r * W + row_of[d]] * rw * 8;
            S->bytes_sent += sby[d];
        }
        // this round's part of the send buffer, in stream order before its messages.  (A launch failure here is local: the messages
        // still go out -- with whatever the buffer holds -- so that no peer waits for them; the status is agreed after the rounds.)
        if (r) { const int e = compact_round(r, xs); if (e && !lrc) lrc = e; }
        // receive buffer r & 1 was last read by the counting of round r - 2: its messages wait for that kernel IN STREAM ORDER (an
        // event recorded behind it on the ctx stream) -- not merely because the host happens to have waited for its result
        if (r >= 2 && hipStreamWaitEvent(xs, ev_counted[r - 2], 0) != hipSuccess) { (void)hipGetLastError(); return tr_rc = X.op_failed("hipStreamWaitEvent in front of all_to_allv"); }
        (void)hipEventRecord(ev_w0[r], xs);
        XTimer th;
        const int e = X.live ? tr->all_to_allv(tr->self, recs.p, soff.data(), sby.data(), rbuf[r & 1].p, roff.data(), rby.data(), xs) : 0;
        host_ms[r] = th.ms();
        if (e) return tr_rc = X.op_failed("all_to_allv (records)");
        (void)hipEventRecord(ev_w1[r], xs);
        if (hipEventRecord(ev_done[r], xs) != hipSuccess) { (void)hipGetLastError(); return tr_rc = X.op_failed("hipEventRecord behind all_to_allv"); }
        return 0;
    };
    S->setup_ms = t_setup.ms();
    // A rank whose counting fails (lrc != 0) keeps taking part in the remaining rounds' exchanges -- its peers' messages must find
    // their receives -- and only stops counting; the status is agreed after the last round.
    if (launch(0)) return tr_rc;
    for (uint32_t r = 0; r < R; r++) {
        // buffer (r + 1) & 1 was last read by the counting of round r - 1: launch() orders round r + 1's messages behind that kernel
        // with ev_counted[r - 1]; round r + 1 goes on the wire now and travels while round r is counted
        if (r + 1 < R && launch(r + 1)) break;
        // the host waits for round r's records itself, with a deadline and an eye on the communicator: the ctx stream is idle here, so
        // this wait IS the exchange time the counting could not hide
        XTimer tw;
        if ((tr_rc = X.wait_event(ev_done[r], "an exchange round's records"))) break;
        wait_ms[r] = tw.ms();
        (void)hipEventRecord(ev_a[r], c->stream);
        if (hipStreamWaitEvent(c->stream, ev_done[r], 0) != hipSuccess) { (void)hipGetLastError(); if (!lrc) lrc = c->fail(100, "hipStreamWaitEvent failed"); }
        (void)hipEventRecord(ev_b[r], c->stream);
        const uint32_t nbc = mycut[r + 1] - mycut[r];
        if (!lrc && X.inject("round") && r == std::min<uint32_t>(1, R - 1)) lrc = X.injected("round");
        if (nbc && !lrc && pl.lists)
            lrc = lists_count_bins(c, ls.get(), recs.p, rbuf[r & 1].p, 1, seg.p + tab_off[r], seg.p + tab_off[r] + 1, W, (uint64_t)nbc + 1, nbc / NCLS,
                                   std::max<uint64_t>(n_local, 1) / R, 0);
        else if (nbc && !lrc)
            lrc = fast_count_bins(c, cs.get(), recs.p, rbuf[r & 1].p, 1, seg.p + tab_off[r], seg.p + tab_off[r] + 1, W, (uint64_t)nbc + 1, nbc / NCLS,
                                  std::max<uint64_t>(n_local, 1) / R, 0);
        else (void)hipStreamSynchronize(c->stream);
        (void)hipEventRecord(ev_counted[r], c->stream);           // (also behind a re-launch after an output buffer grew: the last one counts)
    }
    if (tr_rc) { (void)hipStreamSynchronize(c->stream); return tr_rc; }
    DBG_TRY(X.wait_stream(xs, "the last exchange round"));
    double wire_ms = 0.0;
    for (uint32_t r = 0; r < R; r++) {
        float w = 0.f, wire = 0.f;
        if (hipEventElapsedTime(&w, ev_a[r], ev_b[r]) != hipSuccess) { (void)hipGetLastError(); w = 0.f; }
        if (hipEventElapsedTime(&wire, ev_w0[r], ev_w1[r]) != hipSuccess) { (void)hipGetLastError(); wire = 0.f; }
        // a synchronous transport spends the exchange inside the call: that is exposed time as well (an asynchronous one returns
        // in well under a millisecond)
        const double ex = (double)w + (wait_ms[r] > 0.05 ? wait_ms[r] : 0.0) + (host_ms[r] > 1.0 ? host_ms[r] : 0.0);
        S->exposed_ms_round[r] = ex;
        S->exposed_ms += ex;
        if (r) wire_ms += std::max<double>(wire, host_ms[r] > 1.0 ? host_ms[r] : 0.0);
    }
    sc.slab.release(); sc.cursor.release(); off.release(); ovf_base.release();
    recs.release(); rbuf[0].release(); rbuf[1].release(); seg.release();

    // ---- phase "finish": one order-restoring sort of everything counted ----
    if (!lrc && X.inject("finish")) lrc = X.injected("finish");
    if (!lrc) lrc = pl.lists ? lists_count_finish(c, ls.get(), out) : fast_count_finish(c, cs.get(), out);
    const int arc = X.agree(lrc, "finish");
    if (arc) {
        if (!lrc) { dbg_free_table(c, out); memset(out, 0, sizeof(*out)); }    // this rank's table is of no use without the others'
        return arc;
    }
    out->n_kmer_instances = o.total;
    if (!o.main_run) return 0;
    // what the next call's merge vote looks at: exposed and wire time beyond the first round (always exposed), the merge's cost and
    // what it removed
    c->shard_last_valid = true; c->shard_last_merge = o.merge != 0;
    c->shard_last_exposed_ms = S->exposed_ms - S->exposed_ms_round[0];
    c->shard_last_wire_ms = wire_ms;
    c->shard_last_merge_ratio = n_recs ? (double)(n_recs + sc.n_merged_away) / (double)n_recs : 1.0;
    c->shard_last_merge_cost_ms = 10.0 * (double)n_local / 1.04e10;         // slab_merge: ~9-10 ms per 10^8 reads of 150 bases
    return 0;
}
}  // namespace

namespace {
// CountFilterSet with 65 .. 1024 distinct labels over all ranks (round 5): the label groups of fast_manylabels.hpp across ranks.
// One CountFilter run gives every owner the valid k-mers of its bins (validity counts observations over all labels,
// src/filter.rs:85-100); then, per group of 64 labels, one more run over the reads of that group alone -- min_kmer_obs = 1, the
// WIDE colour layout, the sorted 64-bit masks kept -- in the SAME bin space with the SAME ownership bounds, so that a k-mer's group
// masks arrive at the rank that holds it in T; the masks are joined into T's rows by key and the label lists written from the rows.
// The groups' reads are selected by permuting the reads' metadata (start, length, Exts, colour inside the group); the packed bases
// stay where they are.  Every run agrees on its own status; the rank-local steps between the runs hand theirs to the next run.
int shard_filter_label_groups(dbg_ctx* c, ShardComm& X, const dbg_transport* tr, const dbg_seqset* ds, const dbg_shard_params* p,
                              dbg_shard_plan sp, XRun o, const std::vector<uint32_t>& labels, dbg_kmer_table* out, dbg_shard_stats* S,
                              const XTimer& t_setup) {
    const uint32_t nd = (uint32_t)labels.size(), G = (nd + 63) / 64;
    SeqDev s{ds->words, ds->start, ds->length, ds->exts, ds->data, ds->data ? ds->data_width : 0u, ds->n_seqs, ds->n_words};
    // 1. the valid k-mers of the bins this rank owns.  Bins of half the usual size: the groups' tables (WIDE layout, 1024 entries)
    //    see the same bins, each with up to all of a bin's distinct k-mers
    dbg_shard_plan spA = sp;
    spA.summarizer = DBG_COUNT_FILTER; spA.n_labels = 0; spA.max_label = 0; spA.n_bins = 0;
    {
        FastPlan pl;
        DBG_TRY(plan_from(c, &spA, &pl));                            // (the same arguments on every rank: the same outcome)
        spA.n_bins = (uint32_t)std::min<uint64_t>((uint64_t)pl.nbins * 2, (1ull << 23) - 1) * NCLS;
    }
    std::vector<uint32_t> bounds;
    XRun oa = o;
    oa.bounds_io = &bounds;
    dbg_kmer_table T;
    memset(&T, 0, sizeof(T));
    DBG_TRY(shard_exchange_run(c, X, tr, ds, spA, oa, &T, S, t_setup));
    struct TableGuard { dbg_ctx* c; dbg_kmer_table* t; ~TableGuard() { dbg_free_table(c, t); } } guard_T{c, &T};
    const uint32_t nv = (uint32_t)T.n;

    // 2. this rank's reads by label group (rank-local; its status travels with the first group's run)
    DBuf<uint8_t> d_group, d_colour, p_exts, p_colour;
    DBuf<uint32_t> d_labels, blk_counts, p_length;
    DBuf<uint64_t> blk_off, p_start;
    DBuf<unsigned long long> masks;
    std::vector<uint64_t> g_off(G + 1, 0);
    int lrc = [&]() -> int {
        if (X.inject("groups")) return X.injected("groups");
        const uint32_t top = labels.back();
        std::vector<uint8_t> h_group((size_t)top + 1, 0), h_colour((size_t)top + 1, 0);
        for (uint32_t i = 0; i < nd; i++) { h_group[labels[i]] = (uint8_t)(i / 64); h_colour[labels[i]] = (uint8_t)(i % 64); }
        const uint32_t nblk = cdiv(std::max<uint64_t>(s.n, 1), ML_BLOCK_READS);
        ALLOC_OR_FAIL(c, d_group, (size_t)top + 1); ALLOC_OR_FAIL(c, d_colour, (size_t)top + 1); ALLOC_OR_FAIL(c, d_labels, (size_t)G * 64);
        ALLOC_OR_FAIL(c, blk_counts, (size_t)G * nblk); ALLOC_OR_FAIL(c, blk_off, (size_t)G * nblk + 1);
        ALLOC_OR_FAIL(c, p_start, std::max<uint64_t>(s.n, 1)); ALLOC_OR_FAIL(c, p_length, std::max<uint64_t>(s.n, 1)); ALLOC_OR_FAIL(c, p_colour, std::max<uint64_t>(s.n, 1));
        if (s.exts) ALLOC_OR_FAIL(c, p_exts, std::max<uint64_t>(s.n, 1));
        std::vector<uint32_t> h_labels((size_t)G * 64, 0);
        std::copy(labels.begin(), labels.end(), h_labels.begin());
        HIP_TRY(c, hipMemcpyAsync(d_group.p, h_group.data(), h_group.size(), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(d_colour.p, h_colour.data(), h_colour.size(), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(d_labels.p, h_labels.data(), h_labels.size() * 4, hipMemcpyHostToDevice, c->stream));
        if (s.n) {
            c->t_begin("label_groups", s.n);
            ml_group_count_kernel<<<nblk, 256, 0, c->stream>>>(s.data, s.data_width, s.n, d_group.p, G, blk_counts.p, nblk);
            c->t_end();
            LAUNCH_CHECK(c, "ml_group_count");
            DBG_TRY(scan_exclusive_u32_u64(c, blk_counts.p, blk_off.p, (uint64_t)G * nblk));
            ml_group_scatter_kernel<<<nblk, 256, 0, c->stream>>>(s.data, s.data_width, s.n, d_group.p, d_colour.p, blk_off.p, nblk, s.start, s.length, s.exts,
                                                                p_start.p, p_length.p, p_exts.p, p_colour.p);
            LAUNCH_CHECK(c, "ml_group_scatter");
            for (uint32_t g = 0; g < G; g++) HIP_TRY(c, hipMemcpyAsync(&g_off[g], blk_off.p + (size_t)g * nblk, 8, hipMemcpyDeviceToHost, c->stream));
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream));              // (also: the host tables above leave scope only after their copies)
        g_off[G] = s.n;
        if (!s.n) for (uint32_t g = 0; g < G; g++) g_off[g] = 0;
        ALLOC_OR_FAIL(c, masks, std::max<size_t>((size_t)nv * G, 1));
        HIP_TRY(c, hipMemsetAsync(masks.p, 0, std::max<size_t>((size_t)nv * G, 1) * 8, c->stream));
        return 0;
    }();

    // 3. one 64-colour run per group over all ranks, joined into the rows of masks
    for (uint32_t g = 0; g < G; g++) {
        const uint64_t a = lrc ? 0 : g_off[g], n_g = lrc ? 0 : g_off[g + 1] - a;
        dbg_seqset dg = *ds;
        dg.start = p_start.p ? p_start.p + a : ds->start; dg.length = p_length.p ? p_length.p + a : ds->length;
        dg.exts = s.exts && p_exts.p ? p_exts.p + a : nullptr;
        dg.data = p_colour.p ? (const void*)(p_colour.p + a) : nullptr; dg.data_width = 1; dg.n_seqs = n_g;
        uint64_t nk_g = 0;
        if (!lrc && n_g) lrc = dbg_count_kmer_instances_dev(c, &dg, p->k, &nk_g);
        dbg_shard_plan spg = sp;
        spg.summarizer = DBG_COUNT_FILTER_SET; spg.max_label = 63; spg.n_labels = 0; spg.n_bins = spA.n_bins; spg.min_kmer_obs = 1;
        XRun og = o;
        og.bounds_io = &bounds; og.lrc_in = lrc; og.main_run = false; og.n_local = nk_g;
        DBuf<unsigned long long> msk;
        og.keep_masks = &msk;
        dbg_kmer_table Tg;
        memset(&Tg, 0, sizeof(Tg));
        dbg_shard_stats Sg;
        memset(&Sg, 0, sizeof(Sg));
        const int rc = shard_exchange_run(c, X, tr, &dg, spg, og, &Tg, &Sg, t_setup);
        TableGuard guard_g{c, &Tg};
        if (rc) return rc;                                         // (agreed inside the run: every rank is here)
        S->bytes_sent += Sg.bytes_sent; S->exposed_ms += Sg.exposed_ms; S->records_scanned += Sg.records_scanned; S->records_owned += Sg.records_owned;
        lrc = [&]() -> int {
            if (X.inject("join") && g == G - 1) return X.injected("join");
            if (Tg.n && nv) {
                c->t_begin("label_join", nv);
                ml_join_kernel<<<cdiv(nv, ML_JOIN_TILE), 256, 0, c->stream>>>(T.key_hi, T.key_lo, nv, Tg.key_hi, Tg.key_lo, msk.p, (uint32_t)Tg.n, masks.p, G, g);
                c->t_end();
                LAUNCH_CHECK(c, "ml_join");
            }
            HIP_TRY(c, hipStreamSynchronize(c->stream));          // T_g and its masks are released at the end of this iteration
            return 0;
        }();
    }

    // 4. label lists (rank-local), then the call's last agreement
    DBuf<uint32_t> setn, set_val;
    DBuf<uint64_t> set_off;
    uint64_t n_setval = 0;
    if (!lrc) lrc = [&]() -> int {
        ALLOC_OR_FAIL(c, setn, std::max<uint32_t>(nv, 1)); ALLOC_OR_FAIL(c, set_off, (size_t)nv + 1);
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
        return 0;
    }();
    DBG_TRY(X.agree(lrc, "label lists"));
    *out = T;
    memset(&T, 0, sizeof(T));                                      // ownership of T's arrays moves to *out
    if (out->count) { c->dfree(out->count); out->count = nullptr; }
    out->set_off = set_off.take(); out->set_val = set_val.take(); out->n_set_val = n_setval;
    out->n_passes = 1 + G;
    return 0;
}
}  // namespace
extern "C" int dbg_shard_filter_kmers_dev(dbg_ctx* c, const dbg_transport* tr, const dbg_seqset* ds, const dbg_shard_params* p,
                                          dbg_kmer_table* out, dbg_shard_stats* stats) {
    if (!c) return 1;
    if (!ds || !p || !out) return c->fail(10, "null argument");
    HIP_TRY(c, hipSetDevice(c->device));
    XTimer t_setup;
    const uint32_t W = tr ? (uint32_t)tr->world : 1u, me = tr ? (uint32_t)tr->rank : 0u;
    if (W == 0 || me >= W) return c->fail(161, "sharded flow: bad transport (rank / world)");
    if (W > 64) return c->fail(161, "sharded flow: at most 64 ranks (one record segment per source rank in the counting kernel)");
    if (tr && (W > 1 || p->force_exchange) && (!tr->all_reduce_u64 || !tr->all_to_allv)) return c->fail(161, "sharded flow: the transport lacks all_reduce_u64 / all_to_allv");
    dbg_shard_stats st_local;
    dbg_shard_stats* S = stats ? stats : &st_local;
    memset(S, 0, sizeof(*S));
    memset(out, 0, sizeof(*out));
    const bool is_set = p->summarizer == DBG_COUNT_FILTER_SET;
    const bool collective = W > 1 || (p->force_exchange && tr);
    // Failure agreement (shard_comm.hpp): every phase below runs its rank-local work to the end with the status kept in `lrc`, and
    // ends in X.agree() before the phase's data moves; all ranks leave with the same verdict.
    ShardComm X(c, tr, collective);
    DBG_TRY(X.prepare());
    int lrc = 0;

    // ---- phase "count": global quantities every rank must agree on ----
    uint64_t n_local = 0;
    uint32_t my_max_label = 0;
    lrc = [&]() -> int {
        if (X.inject("count")) return X.injected("count");
        DBG_TRY(dbg_count_kmer_instances_dev(c, ds, p->k, &n_local));
        if (is_set) DBG_TRY(dbg_seqset_max_label_dev(c, ds, &my_max_label));
        return 0;
    }();
    if (lrc) { n_local = 0; my_max_label = 0; }
    // The sender-side merge, when the library is to decide (merge_dups = -1), is decided by ALL ranks together -- a vote from what
    // each ctx measured in its previous call, max-reduced with the other global quantities -- and it is re-evaluated every call:
    //   last call without the merge: on, if the exchange time the counting could not hide (beyond the first round, which is always
    //     exposed) exceeded what the merge costs;
    //   last call with the merge: it stays on while the wire time it saved -- wire time x (records before / after the merge - 1), all
    //     of which would have been exposed on top of what already was -- still exceeds its cost.
    uint64_t vote = 0;
    if (p->merge_dups < 0 && !c->opt("DBG_SHARD_MERGE") && W != 2 && c->shard_last_valid) {
        double cost = c->shard_last_merge_cost_ms;
        if (const char* e = c->opt("DBG_SHARD_MERGE_COST_MS")) cost = atof(e);      // (tests: what the merge is taken to cost)
        if (!c->shard_last_merge) vote = c->shard_last_exposed_ms > cost ? 1 : 0;
        else {
            const double ratio = std::max(1.0, c->shard_last_merge_ratio);
            vote = c->shard_last_exposed_ms + c->shard_last_wire_ms * (ratio - 1.0) > cost ? 1 : 0;
        }
    }
    uint64_t mx3[3] = {n_local, my_max_label, vote};
    DBG_TRY(X.agree(lrc, "count", mx3, 3));
    uint64_t sum1[1] = {n_local};
    DBG_TRY(X.reduce(sum1, 1, 0, "all_reduce_u64 (k-mer instances)"));
    const uint64_t total = sum1[0], n_max = mx3[0];
    const uint32_t max_label = (uint32_t)mx3[1];
    vote = mx3[2];
    S->total_kmers = total; S->local_kmers = n_local;

    if (collective && is_set && max_label >= (1u << 24)) {
        // ---- full-width D1 (round 6; CountFilterSet<D: Ord> takes any u32, filter.rs:68-101) ----
        // The exchange and the counting hold a label in 24 bits.  Every rank ranks its distinct labels (labeldict.hip), the ranks' lists
        // are gathered, every rank builds the same job-wide dictionary from them, the call runs once more on the ranks (their maximum is
        // below 2^24, so this branch is not taken again) and the owner translates its table's set_val back.  Decided from max_label, which
        // all ranks hold alike.
        if (!tr->all_gather) return c->fail(161, "sharded flow: labels of 2^24 and more need the transport's all_gather");
        LabelDict local, glob;
        DBuf<uint32_t> send, recv, compact, rank;
        DBuf<uint64_t> cnt_mine, cnt_all;
        uint64_t nd[1] = {0};
        lrc = [&]() -> int {
            if (X.inject("labeldict")) return X.injected("labeldict");
            if (ds->n_seqs && (!ds->data || ds->data_width != 4)) return c->fail(14, "sharded flow: a rank holds labels of 2^24 and more, so every rank must pass u32 labels (data_width 4)");
            DBG_TRY(label_dict_make(c, (const uint32_t*)ds->data, ds->n_seqs, &local));
            nd[0] = local.n;
            return 0;
        }();
        DBG_TRY(X.agree(lrc, "label dictionary (local)", nd, 1));
        const uint64_t nmax = std::max<uint64_t>(nd[0], 1);
        lrc = [&]() -> int {
            ALLOC_OR_FAIL(c, send, nmax); ALLOC_OR_FAIL(c, recv, nmax * W); ALLOC_OR_FAIL(c, compact, nmax * W);
            ALLOC_OR_FAIL(c, cnt_mine, 1); ALLOC_OR_FAIL(c, cnt_all, W);
            HIP_TRY(c, hipMemsetAsync(send.p, 0, nmax * 4, c->stream));
            if (local.n) HIP_TRY(c, hipMemcpyAsync(send.p, local.dict.p, (size_t)local.n * 4, hipMemcpyDeviceToDevice, c->stream));
            const uint64_t mine = local.n;
            HIP_TRY(c, hipMemcpyAsync(cnt_mine.p, &mine, 8, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            return 0;
        }();
        DBG_TRY(X.agree(lrc, "label dictionary (buffers)"));
        if (tr->all_gather(tr->self, cnt_mine.p, cnt_all.p, 8, c->stream)) return X.op_failed("all_gather (distinct-label counts)");
        if (tr->all_gather(tr->self, send.p, recv.p, nmax * 4, c->stream)) return X.op_failed("all_gather (distinct labels)");
        DBG_TRY(X.wait_stream(c->stream, "the gathered label lists"));
        lrc = [&]() -> int {
            std::vector<uint64_t> cnt(W);
            HIP_TRY(c, hipMemcpyAsync(cnt.data(), cnt_all.p, (size_t)W * 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            uint64_t tot = 0;
            for (uint32_t r = 0; r < W; r++) {
                if (cnt[r] > nmax) return c->fail(166, "label dictionary: a rank reports more distinct labels than the agreed maximum");
                if (cnt[r]) HIP_TRY(c, hipMemcpyAsync(compact.p + tot, recv.p + (uint64_t)r * nmax, (size_t)cnt[r] * 4, hipMemcpyDeviceToDevice, c->stream));
                tot += cnt[r];
            }
            DBG_TRY(label_dict_make(c, compact.p, tot, &glob));              // the same on every rank: the union of all lists
            DBG_TRY(label_dict_rank(c, glob, (const uint32_t*)ds->data, ds->n_seqs, &rank));
            return 0;
        }();
        DBG_TRY(X.agree(lrc, "label dictionary (job-wide)"));
        local.bm.release(); local.base.release(); local.dict.release(); glob.bm.release(); glob.base.release();
        send.release(); recv.release(); compact.release();
        dbg_seqset ranked = *ds;
        ranked.data = rank.p;
        DBG_TRY(dbg_shard_filter_kmers_dev(c, tr, &ranked, p, out, stats));
        lrc = label_dict_translate(c, out->set_val, out->n_set_val, glob.dict.p, glob.n);
        if (int r = X.agree(lrc, "label dictionary (translation)")) { dbg_free_table(c, out); return r; }
        return 0;
    }

    if (!collective) {
        // one rank: the sharded table is the whole table
        dbg_filter_params fp{p->k, p->stranded, p->summarizer, p->min_kmer_obs, 0, 4};
        DBG_TRY(dbg_filter_kmers_dev(c, ds, &fp, out));
        S->n_rounds = 0; S->setup_ms = 0.0;
        return 0;
    }
    // (argument checks on values every rank holds alike need no agreement)
    if (p->k < 4 || p->k > 64) return c->fail(11, "k must be in 4..=64 (filter.rs:18-23 reads the first 4 bases)");
    // Shapes the super-k-mer exchange does not take -- k < 16, more than 64 distinct labels over all ranks, labels >= 65536 -- go
    // the key-range route (shard_generic.hip): decided from values every rank holds alike.  DBG_PATH=generic insists on it.
    bool key_range_route = p->k < 16 || (c->opt("DBG_PATH") && !strcmp(c->opt("DBG_PATH"), "generic"));
    bool label_groups = false, label_lists = false;
    std::vector<uint32_t> label_list;                            // the distinct labels of all ranks, ascending (labels >= 64 only)

    dbg_shard_plan sp;
    memset(&sp, 0, sizeof(sp));
    sp.k = p->k; sp.stranded = p->stranded; sp.summarizer = p->summarizer; sp.min_kmer_obs = p->min_kmer_obs;
    sp.total_kmers = std::max<uint64_t>(total, 1); sp.max_label = max_label;
    if (is_set && max_label >= 64 && !key_range_route) {
        // ---- phase "labels" ----
        // labels beyond the 64 colours of the counting kernel: a sparse alphabet is mapped to colour indices, the same way on every
        // rank -- the union of the ranks' label sets (a max-reduction of presence flags: the transport has sum and max, no OR)
        std::vector<uint32_t> bm(2049, 0u);
        lrc = X.inject("labels") ? X.injected("labels") : dbg_seqset_label_bitmap_dev(c, ds, bm.data());
        DBG_TRY(X.agree(lrc, "labels"));
        std::vector<uint64_t> pres(65537);
        for (uint32_t v = 0; v < 65536; v++) pres[v] = (bm[v >> 5] >> (v & 31)) & 1u;
        pres[65536] = bm[2048] ? 1 : 0;
        DBG_TRY(X.reduce(pres.data(), 65537, 1, "all_reduce_u64 (label presence)"));
        uint32_t nl = 0;
        for (uint32_t v = 0; v < 65536; v++) if (pres[v]) { if (nl < 64) sp.labels[nl] = v; nl++; label_list.push_back(v); }
        // 65 .. 1024 distinct labels (all < 65536): label groups of 64, as on one GPU (fast_manylabels.hpp); beyond that, or labels
        // >= 65536, or a threshold the u16 counts cannot decide: the key-range route
        // More than 64 distinct labels over all ranks, or labels >= 65536: label lists (round 5; every value is one all ranks hold alike).
        // With DBG_LABEL_LISTS=0: label groups for 65 .. 1024 labels < 65536, the key-range route beyond.
        label_lists = (pres[65536] || nl > 64) && max_label < (1u << 24) && !(c->opt("DBG_LABEL_LISTS") && !strcmp(c->opt("DBG_LABEL_LISTS"), "0"));
        label_groups = !label_lists && !pres[65536] && nl > 64 && nl <= 64u * ML_MAX_GROUPS && p->min_kmer_obs <= 65535 && !c->opt("DBG_NO_LABEL_GROUPS");
        if (label_lists) {}
        else if (pres[65536] || (nl > 64 && !label_groups)) key_range_route = true;
        else if (nl <= 64) sp.n_labels = nl;
    }
    if (key_range_route) {
        const int r = shard_filter_generic(c, &X, tr, ds, p, total, out, S);
        S->total_kmers = total; S->local_kmers = n_local;
        return r;
    }
    // sender-side merge: asked for, or decided above: on at 2 ranks, otherwise the ranks' vote
    int merge = p->merge_dups;
    if (merge < 0) {
        if (const char* e = c->opt("DBG_SHARD_MERGE")) merge = atoi(e) != 0;
        else if (W == 2) merge = 1;
        else merge = vote ? 1 : 0;
    }
    sp.merge_dups = merge ? 1u : 0u;


    std::vector<uint32_t> labels_all;
    for (uint32_t v = 0; v < (uint32_t)label_list.size(); v++) labels_all.push_back(label_list[v]);
    XRun o;
    o.n_rounds = p->n_rounds; o.balance = p->balance != 0; o.merge = merge; o.n_local = n_local; o.n_max = n_max; o.total = total;
    if (label_lists) { o.lists = true; o.merge = 0; }                    // (records with different labels must stay apart: no sender-side merge)
    if (label_groups) return shard_filter_label_groups(c, X, tr, ds, p, sp, o, labels_all, out, S, t_setup);
    return shard_exchange_run(c, X, tr, ds, sp, o, out, S, t_setup);
}
