// Rank-spanning form of the generic path (round 5): filter_kmers over the reads of all ranks for the shapes the super-k-mer exchange
// of fast_exchange.hpp does not take -- k < 16 (its internal minimizers are 13 to 15 bases), more than 64 distinct labels over all
// ranks, labels >= 65536 -- so that dbg_shard_filter_kmers_dev covers what dbg_filter_kmers_dev covers (src/filter.rs:139-231 with
// any KmerSummarizer data the C ABI carries, filter.rs:27-35,68-101; the reference's own scale-out has no such limit,
// src/test.rs:433-456).
//
// The partition is the "non-MSP variant" of SURVEY.md section 8(e): ownership by KEY RANGE.  Every rank histograms the top byte of
// its canonical k-mers; the all-reduced histogram is cut greedily into W contiguous byte ranges of nearly equal record count (the
// reference's bucket ranges, filter.rs:156-168, spread over ranks instead of passes); an owner's range is cut further into rounds so
// that a round's records fit its memory.  Per round every rank extracts the k-mer instances of each owner's sub-range (the generic
// path's own range kernels: iter_kmer_exts + min_rc_flip + Exts::rc fused, lib.rs:812-841), the three record columns travel with
// one variable all-to-all each, and the owner sorts and reduces what it received (stable LSD radix sort by (key, D1), group_by +
// summarize).  Sub-ranges ascend, so an owner's round tables simply follow one another, and the owners' tables are the consecutive
// pieces of the table dbg_filter_kmers_dev returns for the concatenated reads.
// Every message size follows from the ranks' byte histograms (all-gathered once): no count exchange per round.
// Failure agreement as in the super-k-mer flow (shard_comm.hpp): local work runs to the end of its phase, the status is agreed
// before the phase's data moves.
#include "dbg_internal.hpp"
#include "shard_comm.hpp"
#include <algorithm>

int shard_filter_generic(dbg_ctx* c, void* shard_comm, const dbg_transport* tr, const dbg_seqset* ds, const dbg_shard_params* p,
                         uint64_t total_kmers, dbg_kmer_table* out, dbg_shard_stats* S) {
    ShardComm& X = *static_cast<ShardComm*>(shard_comm);
    const uint32_t W = X.W, me = X.me;
    const int k = (int)p->k;
    const bool has_hi = k > 32, is_set = p->summarizer == DBG_COUNT_FILTER_SET, stranded = p->stranded != 0;
    if (k < 4 || k > 64) return c->fail(11, "k must be in 4..=64 (filter.rs:18-23 reads the first 4 bases)");
    if (W > 1 && !tr->all_gather) return c->fail(161, "sharded flow (key-range route): the transport lacks all_gather");
    SeqDev s{ds->words, ds->start, ds->length, ds->exts, ds->data, ds->data ? ds->data_width : 0u, ds->n_seqs, ds->n_words};
    int lrc = 0;

    // ---- phase "histogram": top bytes of this rank's canonical k-mers; what a pass may hold here ----
    DBuf<unsigned long long> d_hist;
    DBuf<uint64_t> d_all;
    DBuf<uint32_t> kcount;
    DBuf<uint64_t> koff;
    uint64_t pass_max = (1ull << 32) - 1;
    lrc = [&]() -> int {
        if (X.inject("histogram")) return X.injected("histogram");
        if (s.data && s.data_width == 4) {
            // (the payload word carries Exts | D1 << 8: D1 < 2^24, as in the single-GPU generic path)
            uint32_t mx = 0;
            DBG_TRY(dbg_seqset_max_label_dev(c, ds, &mx));
            if (is_set && mx >= (1u << 24)) return c->fail(17, "D1 values must be < 2^24");
        }
        ALLOC_OR_FAIL(c, d_hist, 256);
        ALLOC_OR_FAIL(c, d_all, (size_t)256 * W);
        ALLOC_OR_FAIL(c, kcount, std::max<uint64_t>(s.n, 1));
        ALLOC_OR_FAIL(c, koff, s.n + 1);
        DBG_TRY(kmer_top_byte_hist(c, s, k, stranded, d_hist.p));
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const uint64_t budget = c->scratch_budget ? c->scratch_budget : (uint64_t)((free_b + c->pooled_bytes) * 0.5);
            const uint64_t per_rec = 3ull * ((has_hi ? 8 : 0) + 8 + 4) + 24;       // send columns + two record buffers + sort/reduce scratch
            pass_max = std::min<uint64_t>(pass_max, std::max<uint64_t>(budget / per_rec, 1u << 20));
        } else (void)hipGetLastError();
        if (const char* e = c->opt("DBG_GENERIC_PASS_MAX")) pass_max = std::max<uint64_t>(1, strtoull(e, nullptr, 10));   // tests: force several rounds
        return 0;
    }();
    // (the smallest rank decides how much a round may hold: a MAX-reduction of the complement, next to the status)
    uint64_t inv_pm[1] = {~pass_max};
    DBG_TRY(X.agree(lrc, "histogram", inv_pm, 1));
    pass_max = ~inv_pm[0];
    std::vector<uint64_t> hall((size_t)256 * W, 0);
    if (X.live && tr->all_gather) {
        if (tr->all_gather(tr->self, d_hist.p, d_all.p, 256 * 8, c->stream)) return X.op_failed("all_gather (top-byte histograms)");
        if (hipMemcpyAsync(hall.data(), d_all.p, hall.size() * 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { (void)hipGetLastError(); return X.op_failed("copy of the histograms"); }
    } else if (hipMemcpyAsync(hall.data(), d_hist.p, 256 * 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { (void)hipGetLastError(); return X.op_failed("copy of the histogram"); }
    DBG_TRY(X.wait_stream(c->stream, "the top-byte histograms of all ranks"));
    auto H = [&](uint32_t r, uint32_t b) -> uint64_t { return hall[(size_t)r * 256 + b]; };
    uint64_t glob[256], total = 0;
    for (uint32_t b = 0; b < 256; b++) { glob[b] = 0; for (uint32_t r = 0; r < W; r++) glob[b] += H(r, b); total += glob[b]; }

    // ---- ownership (byte boundaries) and rounds: host arithmetic on values every rank holds alike ----
    std::vector<uint32_t> bb(W + 1, 256);
    bb[0] = 0;
    {
        uint64_t run = 0;
        uint32_t b = 0;
        for (uint32_t r = 1; r < W; r++) {
            const uint64_t target = (uint64_t)((unsigned __int128)total * r / W);
            while (b < 256 && run + glob[b] <= target) run += glob[b++];
            if (b < 256 && target - run > run + glob[b] - target) run += glob[b++];      // the nearer of the two boundaries around the target
            bb[r] = std::max(b, bb[r - 1]);
        }
    }
    // rounds: owner d's range is cut greedily into pieces of at most pass_max records; every rank runs as many rounds as the owner
    // with the most pieces (an owner that has run out of pieces takes part with empty messages)
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> piece(W);
    uint32_t R = 1;
    for (uint32_t d = 0; d < W; d++) {
        uint32_t lo = bb[d];
        uint64_t acc = 0;
        for (uint32_t b = bb[d]; b < bb[d + 1]; b++) {
            if (glob[b] > pass_max) return c->fail(20, "sharded flow (key-range route): one top-byte range of k-mers alone exceeds what a round can hold");
            if (acc + glob[b] > pass_max) { piece[d].push_back({lo, b}); lo = b; acc = 0; }
            acc += glob[b];
        }
        piece[d].push_back({lo, bb[d + 1]});
        R = std::max<uint32_t>(R, (uint32_t)piece[d].size());
    }
    S->n_rounds = R; S->balanced = 1; S->merge_dups = 0; S->n_bins = 256; S->owned_lo = bb[me]; S->owned_hi = bb[me + 1];
    auto rng = [&](uint32_t d, uint32_t j) -> std::pair<uint32_t, uint32_t> { return j < piece[d].size() ? piece[d][j] : std::make_pair(bb[d + 1], bb[d + 1]); };
    auto cnt_of = [&](uint32_t src, uint32_t d, uint32_t j) -> uint64_t { uint64_t n = 0; auto q = rng(d, j); for (uint32_t b = q.first; b < q.second; b++) n += H(src, b); return n; };

    // ---- rounds ----
    std::vector<ReduceOut> parts;
    auto free_parts = [&]() {
        for (auto& r : parts) for (void* q : {(void*)r.key_hi, (void*)r.key_lo, (void*)r.exts, (void*)r.count, (void*)r.set_off, (void*)r.set_val, (void*)r.all_hi, (void*)r.all_lo}) c->dfree(q);
        parts.clear();
    };
    const int pay_bits = is_set && s.data ? (s.data_width == 1 ? 8 : (s.data_width == 2 ? 16 : 24)) : 0;
    for (uint32_t j = 0; j < R; j++) {
        // phase "extract": this rank's instances of every owner's sub-range, destination by destination; room for what arrives
        std::vector<uint64_t> scnt(W), rcnt(W), sbase(W), rbase(W);
        uint64_t n_send = 0, n_recv = 0;
        for (uint32_t d = 0; d < W; d++) { scnt[d] = cnt_of(me, d, j); sbase[d] = n_send; n_send += scnt[d]; }
        for (uint32_t src = 0; src < W; src++) { rcnt[src] = cnt_of(src, me, j); rbase[src] = n_recv; n_recv += rcnt[src]; }
        DBuf<uint64_t> s_hi, s_lo, a_hi, a_lo, b_hi, b_lo;
        DBuf<uint32_t> s_pay, a_pay, b_pay;
        if (!lrc) lrc = [&]() -> int {
            if (X.inject("extract") && j == R - 1) return X.injected("extract");
            const size_t ns = std::max<uint64_t>(n_send, 1), nr = std::max<uint64_t>(n_recv, 1);
            if (n_recv >= (1ull << 32)) return c->fail(20, "sharded flow (key-range route): a round of more than 2^32-1 records");
            if (has_hi) { ALLOC_OR_FAIL(c, s_hi, ns); ALLOC_OR_FAIL(c, a_hi, nr); ALLOC_OR_FAIL(c, b_hi, nr); }
            ALLOC_OR_FAIL(c, s_lo, ns); ALLOC_OR_FAIL(c, a_lo, nr); ALLOC_OR_FAIL(c, b_lo, nr);
            ALLOC_OR_FAIL(c, s_pay, ns); ALLOC_OR_FAIL(c, a_pay, nr); ALLOC_OR_FAIL(c, b_pay, nr);
            for (uint32_t d = 0; d < W; d++) {
                if (!scnt[d]) continue;
                const auto q = rng(d, j);
                // a rank's own records go straight into its receive columns (they never travel)
                RecArrays dst = d == me ? RecArrays{has_hi ? a_hi.p + rbase[me] : nullptr, a_lo.p + rbase[me], a_pay.p + rbase[me]}
                                        : RecArrays{has_hi ? s_hi.p + sbase[d] : nullptr, s_lo.p + sbase[d], s_pay.p + sbase[d]};
                DBG_TRY(kmer_counts_range(c, s, k, stranded, q.first, q.second, kcount.p));
                DBG_TRY(scan_exclusive_u32_u64(c, kcount.p, koff.p, s.n));
                DBG_TRY(extract_kmers_range(c, s, koff.p, scnt[d], k, stranded, q.first, q.second, dst));
            }
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            return 0;
        }();
        if (int r = X.agree(lrc, "extract")) { free_parts(); return r; }
        if (X.live) {
            std::vector<uint64_t> so(W), sb(W), ro(W), rb(W);
            auto xchg = [&](const void* sp, void* rp, uint32_t width, const char* what) -> int {
                for (uint32_t d = 0; d < W; d++) {
                    so[d] = sbase[d] * width; sb[d] = d == me ? 0 : scnt[d] * width;
                    ro[d] = rbase[d] * width; rb[d] = d == me ? 0 : rcnt[d] * width;
                    S->bytes_sent += sb[d];
                }
                if (tr->all_to_allv(tr->self, sp, so.data(), sb.data(), rp, ro.data(), rb.data(), c->stream)) return X.op_failed(what);
                return 0;
            };
            if (has_hi && xchg(s_hi.p, a_hi.p, 8, "all_to_allv (key_hi)")) { free_parts(); return 160; }
            if (xchg(s_lo.p, a_lo.p, 8, "all_to_allv (key_lo)")) { free_parts(); return 160; }
            if (xchg(s_pay.p, a_pay.p, 4, "all_to_allv (payload)")) { free_parts(); return 160; }
            if (int r = X.wait_stream(c->stream, "a round's k-mer records")) { free_parts(); return r; }
        }
        S->records_owned += n_recv; S->records_scanned += n_send;
        s_hi.release(); s_lo.release(); s_pay.release();
        // phase "reduce" (local; its status is agreed with the next round's, or at the end)
        lrc = [&]() -> int {
            if (X.inject("reduce") && j == 0) return X.injected("reduce");
            RecArrays A{has_hi ? a_hi.p : nullptr, a_lo.p, a_pay.p}, B{has_hi ? b_hi.p : nullptr, b_lo.p, b_pay.p};
            bool in_b = false;
            ReduceOut ro;
            DBG_TRY(radix_sort_records(c, n_recv, A, B, 2 * k, 8, pay_bits, &in_b));
            DBG_TRY(reduce_sorted_records(c, n_recv, in_b ? B : A, has_hi, p->summarizer, p->min_kmer_obs, false, &ro));
            parts.push_back(ro);
            return 0;
        }();
    }
    ReduceOut r;
    if (!lrc) {
        if (parts.empty()) parts.push_back(ReduceOut());
        lrc = join_reduce_parts(c, parts, is_set, false, &r);
        if (!lrc && !has_hi && r.n_valid && hipMemsetAsync(r.key_hi, 0, r.n_valid * 8, c->stream) != hipSuccess) { (void)hipGetLastError(); lrc = c->fail(100, "hipMemsetAsync failed"); }
        if (!lrc && hipStreamSynchronize(c->stream) != hipSuccess) { (void)hipGetLastError(); lrc = c->fail(100, "hipStreamSynchronize failed"); }
    } else free_parts();
    if (int a = X.agree(lrc, "finish")) {
        if (!lrc) for (void* q : {(void*)r.key_hi, (void*)r.key_lo, (void*)r.exts, (void*)r.count, (void*)r.set_off, (void*)r.set_val}) c->dfree(q);
        return a;
    }
    memset(out, 0, sizeof(*out));
    out->n = r.n_valid; out->key_hi = r.key_hi; out->key_lo = r.key_lo; out->exts = r.exts; out->count = r.count;
    out->set_off = r.set_off; out->set_val = r.set_val; out->n_set_val = r.n_set_val;
    out->n_kmer_instances = total_kmers; out->n_passes = R; out->on_device = 1;
    return 0;
}
