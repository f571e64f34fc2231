// The rank-spanning entry points of the C ABI driven from a compiled, multi-threaded host -- no Python, no torch, no RCCL:
// W ranks = W threads of this process, each with its own dbg_ctx on GPU 0, connected by the library's in-process transport
// (dbg_transport_inprocess_create: device-to-device copies between the ranks' buffers).  What a Rust host that drives a node's GPUs
// from one process would do, one thread per GPU.
//
//   every rank:  its share of a synthetic read stream -> dbg_seqset_to_device -> dbg_shard_filter_kmers_dev
//                -> dbg_shard_compress_dev (gather to rank 0; then once more as a tree)
//   checks:      the ranks' tables are disjoint and their union is dbg_filter_kmers over ALL reads (keys, Exts, counts);
//                the gathered graph equals, array for array, the flow composed call by call from the ranks' tables
//                (per-table dbg_compress_kmers_with_hash -> dbg_graph_combine -> dbg_compress_graph: src/test.rs:459-470);
//                the tree gives the same number of nodes and bases; records owned add up to records scanned.
#include "dbg_mi355x.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define FAIL(code, ...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return code; } while (0)

struct Reads {
    std::vector<uint64_t> words, start;
    std::vector<uint32_t> length;
    dbg_seqset ss{};
    void make(uint64_t n, uint64_t first, uint64_t genome) {
        dbg_synth_params p{};
        p.n_reads = n; p.read_len = 150; p.genome_len = genome; p.genome_seed = 0xDB60001; p.read_seed = 0xDB60002;
        p.error_rate = 0.001; p.stranded = 0; p.n_colours = 0; p.first_read = first;
        words.assign(dbg_synth_words(&p), 0); start.assign(n, 0); length.assign(n, 0);
        dbg_synth_reads_host(&p, words.data(), start.data(), length.data(), nullptr);
        ss.words = words.data(); ss.n_words = words.size(); ss.start = start.data(); ss.length = length.data(); ss.n_seqs = n;
    }
};

struct HostTable { std::vector<uint64_t> hi, lo; std::vector<uint8_t> exts; std::vector<uint16_t> count; };

static bool same_graph(const dbg_graph& a, const dbg_graph& b) {
    if (a.n_nodes != b.n_nodes || a.seq_len_bases != b.seq_len_bases || a.stranded != b.stranded) return false;
    const uint64_t nw = (a.seq_len_bases + 31) / 32;
    return !memcmp(a.seq_words, b.seq_words, nw * 8) && !memcmp(a.start, b.start, a.n_nodes * 8) && !memcmp(a.length, b.length, a.n_nodes * 4) &&
           !memcmp(a.exts, b.exts, a.n_nodes) && !memcmp(a.data, b.data, a.n_nodes * 4);
}

int main(int argc, char** argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 3;
    const uint32_t K = 47;
    const uint64_t per = 20000, total = per * W, genome = total * 150 / 30;
    std::vector<dbg_transport*> tr(W);
    if (dbg_transport_inprocess_create(W, tr.data())) FAIL(2, "dbg_transport_inprocess_create failed");

    std::vector<HostTable> tabs(W);
    std::vector<dbg_shard_stats> stats(W);
    std::vector<int> rc(W, 0);
    dbg_graph fin_gather{}, fin_tree{};
    dbg_ctx* ctx0 = nullptr;
    std::vector<dbg_ctx*> ctxs(W, nullptr);
    // argv[2] = number of devices to spread the thread-ranks over (1: all on device 0; on a multi-GPU node rank r drives device
    // r % n, and the in-process transport's copies are real peer copies)
    const int n_dev = argc > 2 ? std::max(1, atoi(argv[2])) : 1;
    for (int r = 0; r < W; r++)
        if (dbg_ctx_create(r % n_dev, &ctxs[r])) FAIL(3, "dbg_ctx_create: %s", dbg_last_error(nullptr));
    ctx0 = ctxs[0];

    auto rank_main = [&](int r) {
        dbg_ctx* c = ctxs[r];
        Reads rd;
        rd.make(per, (uint64_t)r * per, genome);
        dbg_seqset dev{};
        if (dbg_seqset_to_device(c, &rd.ss, &dev)) { rc[r] = 10; return; }
        dbg_shard_params p{};
        p.k = K; p.stranded = 0; p.summarizer = DBG_COUNT_FILTER; p.min_kmer_obs = 2; p.n_rounds = 0; p.merge_dups = r == 1 ? 1 : 0; p.balance = 1;
        p.force_exchange = W == 1;                       // one rank: take the collective route anyway (the transport's one-rank case)
        dbg_kmer_table t{}, h{};
        if (dbg_shard_filter_kmers_dev(c, tr[r], &dev, &p, &t, &stats[r])) { fprintf(stderr, "rank %d: %s\n", r, dbg_last_error(c)); rc[r] = 11; return; }
        dbg_seqset_free_device(c, &dev);
        if (dbg_table_to_host(c, &t, &h)) { rc[r] = 12; return; }
        tabs[r].hi.assign(h.key_hi, h.key_hi + h.n); tabs[r].lo.assign(h.key_lo, h.key_lo + h.n);
        tabs[r].exts.assign(h.exts, h.exts + h.n); tabs[r].count.assign(h.count, h.count + h.n);
        dbg_free_table(c, &h);
        for (int mode = 0; mode < 2; mode++) {
            dbg_graph g{};
            if (dbg_shard_compress_dev(c, tr[r], K, 0, DBG_SPEC_SIMPLE_SAT_ADD_U16, DBG_SPEC_SIMPLE_MAX_U16, &t, mode ? DBG_REDUCE_TREE : DBG_REDUCE_GATHER, 0, &g,
                                       nullptr, nullptr)) { fprintf(stderr, "rank %d: %s\n", r, dbg_last_error(c)); rc[r] = 13; return; }
            if (r == 0) (mode ? fin_tree : fin_gather) = g;
            else { if (g.n_nodes) rc[r] = 14; dbg_free_graph(c, &g); }
        }
        dbg_free_table(c, &t);
    };
    std::vector<std::thread> th;
    for (int r = 0; r < W; r++) th.emplace_back(rank_main, r);
    for (auto& t : th) t.join();
    for (int r = 0; r < W; r++) if (rc[r]) FAIL(rc[r], "rank %d failed with %d", r, rc[r]);

    // the single call over all reads
    Reads all;
    all.make(total, 0, genome);
    dbg_filter_params fp{K, 0, DBG_COUNT_FILTER, 2, 0, 4};
    dbg_kmer_table ref{};
    if (dbg_filter_kmers(ctx0, &all.ss, &fp, &ref)) FAIL(20, "dbg_filter_kmers: %s", dbg_last_error(ctx0));
    uint64_t n_sum = 0, owned = 0, scanned = 0;
    for (int r = 0; r < W; r++) { n_sum += tabs[r].lo.size(); owned += stats[r].records_owned; scanned += stats[r].records_scanned; }
    if (n_sum != ref.n || ref.n < 50000) FAIL(21, "tables hold %llu k-mers, the single call %llu", (unsigned long long)n_sum, (unsigned long long)ref.n);
    if (owned != scanned || !owned) FAIL(22, "records owned %llu != scanned %llu", (unsigned long long)owned, (unsigned long long)scanned);
    if (W > 1 && (!stats[1].merge_dups || stats[0].merge_dups)) FAIL(23, "merge_dups was not a per-rank choice");
    // merge the ranks' ascending tables by key: it must reproduce the single call's table row for row
    std::vector<size_t> pos(W, 0);
    for (uint64_t i = 0; i < ref.n; i++) {
        int best = -1;
        for (int r = 0; r < W; r++) {
            if (pos[r] >= tabs[r].lo.size()) continue;
            if (best < 0 || tabs[r].hi[pos[r]] < tabs[best].hi[pos[best]] ||
                (tabs[r].hi[pos[r]] == tabs[best].hi[pos[best]] && tabs[r].lo[pos[r]] < tabs[best].lo[pos[best]])) best = r;
        }
        const size_t q = pos[best]++;
        if (tabs[best].hi[q] != ref.key_hi[i] || tabs[best].lo[q] != ref.key_lo[i] || tabs[best].exts[q] != ref.exts[i] || tabs[best].count[q] != ref.count[i])
            FAIL(24, "row %llu of the merged tables differs from the single call", (unsigned long long)i);
    }
    dbg_free_table(ctx0, &ref);

    // the second stage composed call by call from the ranks' tables (host boundary)
    std::vector<dbg_graph> shard(W);
    for (int r = 0; r < W; r++) {
        std::vector<uint32_t> d(tabs[r].count.begin(), tabs[r].count.end());
        if (dbg_compress_kmers_with_hash(ctx0, K, 0, DBG_SPEC_SIMPLE_SAT_ADD_U16, tabs[r].lo.size(), tabs[r].hi.data(), tabs[r].lo.data(), tabs[r].exts.data(),
                                         d.data(), nullptr, &shard[r])) FAIL(30, "compress: %s", dbg_last_error(ctx0));
    }
    dbg_graph comb{}, want{};
    if (dbg_graph_combine(ctx0, shard.data(), (uint32_t)W, &comb)) FAIL(31, "combine: %s", dbg_last_error(ctx0));
    if (dbg_compress_graph(ctx0, K, 0, DBG_SPEC_SIMPLE_MAX_U16, &comb, nullptr, 0, &want)) FAIL(32, "compress_graph: %s", dbg_last_error(ctx0));
    if (!want.n_nodes || !same_graph(fin_gather, want)) FAIL(33, "gathered graph (%llu nodes) differs from the composed flow (%llu nodes)",
                                                              (unsigned long long)fin_gather.n_nodes, (unsigned long long)want.n_nodes);
    if (fin_tree.n_nodes != want.n_nodes || fin_tree.seq_len_bases != want.seq_len_bases) FAIL(34, "tree merge: %llu nodes / %llu bases, gather %llu / %llu",
        (unsigned long long)fin_tree.n_nodes, (unsigned long long)fin_tree.seq_len_bases, (unsigned long long)want.n_nodes, (unsigned long long)want.seq_len_bases);
    printf("%d ranks (threads, in-process transport): %llu valid k-mers, %llu unitigs, owned bins", W, (unsigned long long)n_sum, (unsigned long long)want.n_nodes);
    for (int r = 0; r < W; r++) printf(" [%u, %u)", stats[r].owned_lo, stats[r].owned_hi);
    printf("\n");
    for (auto& g : shard) dbg_free_graph(ctx0, &g);
    dbg_free_graph(ctx0, &comb); dbg_free_graph(ctx0, &want); dbg_free_graph(ctx0, &fin_gather); dbg_free_graph(ctx0, &fin_tree);
    for (int r = 0; r < W; r++) { dbg_transport_destroy(tr[r]); dbg_ctx_destroy(ctxs[r]); }
    printf("shard threads ok\n");
    return 0;
}
