// C++ rendition of the reference's own end-to-end checks (src/test.rs:299-414 `reassemble_contigs`,
// src/test.rs:418-504 `reassemble_sharded`, kmer.rs:10-34 doctest) written against the C++ host mirror
// include/debruijn_mi355x.hpp -- i.e. what the crate's tests look like after switching to the MI355X path.
// No oracle is involved: the assertions are the reference's invariants, computed natively.
#include <algorithm>
#include <cstdio>
#include <map>
#include <random>
#include <set>
#include "debruijn_mi355x.hpp"

using namespace debruijn;
typedef unsigned __int128 u128;

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

template <class K> static K rc(K a) { K r; for (int i = 0; i < K::k(); i++) r.storage = (r.storage << 2) | (3 - ((a.storage >> (2 * i)) & 3)); return r; }
template <class K> static K min_rc(K a) { K r = rc(a); return a < r ? a : r; }
template <class K> static std::vector<K> kmers_from_bytes(const std::vector<uint8_t>& s) {      // lib.rs:288-306
    std::vector<K> out;
    if ((int)s.size() < K::k()) return out;
    u128 mask = K::k() == 64 ? ~(u128)0 : (((u128)1 << (2 * K::k())) - 1);
    K cur;
    for (size_t i = 0; i < s.size(); i++) {
        cur.storage = ((cur.storage << 2) | s[i]) & mask;
        if ((int)i + 1 >= K::k()) out.push_back(cur);
    }
    return out;
}
template <class K> static K extend(K a, uint8_t b, Dir d) {                                       // kmer.rs:469-487
    u128 mask = K::k() == 64 ? ~(u128)0 : (((u128)1 << (2 * K::k())) - 1);
    K r;
    if (d == Dir::Right) r.storage = ((a.storage << 2) | b) & mask;
    else r.storage = (a.storage >> 2) | ((u128)b << (2 * (K::k() - 1)));
    return r;
}
static std::vector<uint8_t> random_dna(std::mt19937_64& r, size_t n) { std::vector<uint8_t> v(n); for (auto& x : v) x = r() % 4; return v; }
static std::vector<std::vector<uint8_t>> simple_random_contigs(std::mt19937_64& r) {              // test.rs:58-95
    auto p1 = random_dna(r, 40), p2 = random_dna(r, 30), pc = random_dna(r, 100), p3 = random_dna(r, 30), p4 = random_dna(r, 40);
    std::vector<uint8_t> c1 = p1, c2 = p2, c3 = random_dna(r, 30);
    c1.insert(c1.end(), pc.begin(), pc.end()); c1.insert(c1.end(), p3.begin(), p3.end());
    c2.insert(c2.end(), pc.begin(), pc.end()); c2.insert(c2.end(), p4.begin(), p4.end());
    auto pal = random_dna(r, 33);
    c3.insert(c3.end(), pal.begin(), pal.end());
    for (auto it = pal.rbegin(); it != pal.rend(); ++it) c3.push_back(3 - *it);
    auto tail = random_dna(r, 50); c3.insert(c3.end(), tail.begin(), tail.end());
    return {c1, c2, c3};
}

template <class K> static int check_graph(const BaseGraph<K, uint16_t>& graph, const std::set<u128>& truth) {   // test.rs:386-413
    std::set<u128> all;
    for (size_t i = 0; i < graph.len(); i++) {
        DnaString s = graph.sequences.get(i);
        std::vector<uint8_t> b; for (size_t j = 0; j < s.len; j++) b.push_back(s.get(j));
        auto ks = kmers_from_bytes<K>(b);
        CHECK(!ks.empty());
        for (auto& km : ks) { u128 c = min_rc(km).storage; CHECK(truth.count(c)); all.insert(c); }
        for (uint8_t x = 0; x < 4; x++) {
            if (graph.exts[i].has_ext(Dir::Left, x)) CHECK(truth.count(min_rc(extend(ks.front(), x, Dir::Left)).storage));
            if (graph.exts[i].has_ext(Dir::Right, x)) CHECK(truth.count(min_rc(extend(ks.back(), x, Dir::Right)).storage));
        }
    }
    CHECK(all == truth);
    return 0;
}

template <class K> static int reassemble(Context& ctx, const std::vector<std::vector<uint8_t>>& contigs) {
    const bool stranded = false;
    std::set<u128> truth;
    for (auto& c : contigs) for (auto& km : kmers_from_bytes<K>(c)) truth.insert(min_rc(km).storage);
    // msp pieces, each fed twice (test.rs:318-324)
    std::vector<std::tuple<DnaString, Exts, uint8_t>> seqs;
    std::map<uint32_t, std::vector<std::tuple<DnaString, Exts, uint8_t>>> shards;
    for (auto& c : contigs) {
        auto msps = msp_sequence<6>(ctx, K::k(), c, nullptr, true);
        for (uint8_t d = 0; d < 2; d++)
            for (auto& m : msps) { seqs.emplace_back(std::get<2>(m), std::get<1>(m), d); shards[std::get<0>(m)].emplace_back(std::get<2>(m), std::get<1>(m), d); }
    }
    // un-sharded: filter_kmers + compress_kmers_with_hash (test.rs:344-413)
    auto res = filter_kmers<K>(ctx, seqs, CountFilter(2), stranded, false, 4);
    std::set<u128> got; for (auto& k : res.first.keys) got.insert(k.storage);
    CHECK(got == truth);
    CHECK(std::is_sorted(res.first.keys.begin(), res.first.keys.end()));
    auto graph = compress_kmers_with_hash<K>(ctx, stranded, SimpleCompress(Reduce::SaturatingAdd), res.first);
    if (check_graph<K>(graph, truth)) return 1;
    {   // compress_kmers_no_exts (compression.rs:619-659) on the bare k-mer set: the Exts found by neighbour look-ups make the
        // same unitigs' k-mers (every contig k-mer is in the set, so the look-ups see at least what the reads showed)
        std::vector<std::pair<K, uint16_t>> bare;
        for (size_t i = 0; i < res.first.len(); i++) bare.emplace_back(res.first.keys[i], res.first.data[i]);
        auto g2 = compress_kmers_no_exts<K>(ctx, stranded, SimpleCompress(Reduce::SaturatingAdd), bare);
        if (check_graph<K>(g2, truth)) return 1;
    }
    // sharded: per-shard filter + compress, combine, compress_graph (test.rs:446-503)
    std::vector<BaseGraph<K, uint16_t>> shard_asms;
    for (auto& kv : shards) {
        auto r = filter_kmers<K>(ctx, kv.second, CountFilter(2), stranded, false, 4);
        shard_asms.push_back(compress_kmers_with_hash<K>(ctx, stranded, SimpleCompress(Reduce::SaturatingAdd), r.first));
    }
    auto combined = combine<K, uint16_t>(ctx, shard_asms);
    auto dbg = compress_graph<K, uint16_t>(ctx, false, SimpleCompress(Reduce::Max), combined);
    if (check_graph<K>(dbg, truth)) return 1;
    size_t nk = 0; for (auto l : dbg.sequences.length) nk += l - K::k() + 1;
    CHECK(nk == truth.size());                               // each k-mer in exactly one node
    return 0;
}

int main() {
    // kmer.rs:10-34 doctest (ordering + formatting of the key type that crosses the boundary)
    typedef Kmer<16> Kmer16;
    std::vector<Kmer16> ks;
    std::string s = "TACGTACGTACGTACGTT";
    for (int i = 0; i < 3; i++) ks.push_back(Kmer16::from_ascii(s.substr(i, 16)));
    std::sort(ks.begin(), ks.end());
    if (!(ks[0].to_string() == "ACGTACGTACGTACGT" && ks[1].to_string() == "CGTACGTACGTACGTT" && ks[2].to_string() == "TACGTACGTACGTACG")) return 2;
    try {
        Context ctx(0);
        std::mt19937_64 rng(12345);
        for (int it = 0; it < 3; it++) {
            auto contigs = simple_random_contigs(rng);
            if (reassemble<Kmer<32>>(ctx, contigs)) return 1;
            if (reassemble<Kmer<31>>(ctx, contigs)) return 1;
            if (reassemble<Kmer<47>>(ctx, contigs)) return 1;
        }
        // CountFilterSet through the mirror
        std::vector<std::tuple<DnaString, Exts, uint8_t>> seqs;
        auto g = random_dna(rng, 200);
        for (int i = 0; i < 30; i++) seqs.emplace_back(DnaString::from_bytes(g), Exts::empty(), (uint8_t)(i % 3));
        auto rs = filter_kmers<Kmer<47>>(ctx, seqs, CountFilterSet(2), true, false, 4);
        if (rs.first.len() != 200 - 47 + 1) return 3;
        for (auto& v : rs.first.data) if (v != std::vector<uint8_t>({0, 1, 2})) return 4;
        // from_acgt_bytes on the device == from_dna_string (dna_string.rs:968-971), incl. non-ACGT -> A
        {
            std::string a = "ACGTAAAAAAAAAATTATATAACGTacgtNNACGTACGTACGTACGTACGTACGTACGTACGTACGTAC";
            DnaString x = from_acgt_bytes(ctx, a), y = DnaString::from_dna_string(a);
            if (x.len != y.len || x.storage != y.storage) return 6;
            // from_acgt_bytes_hashn (dna_string.rs:255-278): valid characters as above, the others repeatable per (name, pos)
            DnaString h1 = from_acgt_bytes_hashn(ctx, a, "read1"), h2 = from_acgt_bytes_hashn(ctx, a, "read1");
            if (h1.len != y.len || h1.storage != h2.storage) return 60;
            for (size_t i = 0; i < a.size(); i++)
                if (a[i] != 'N' && h1.get(i) != y.get(i)) return 61;
            std::string clean = "ACGTACGTTTGACCA";
            if (from_acgt_bytes_hashn(ctx, clean, "r").storage != DnaString::from_dna_string(clean).storage) return 62;
        }
        // GFA text of a compressed graph: header, one S line per node, every L line overlaps by k-1 (graph.rs:537-611)
        {
            auto contigs = simple_random_contigs(rng);
            std::vector<std::tuple<DnaString, Exts, uint8_t>> cs;
            for (auto& c : contigs) cs.emplace_back(DnaString::from_bytes(c), Exts::empty(), (uint8_t)0);
            auto fk = filter_kmers<Kmer<31>>(ctx, cs, CountFilter(1), false, false, 4);
            auto gr = compress_kmers_with_hash<Kmer<31>>(ctx, false, SimpleCompress(), fk.first);
            std::string gfa = write_gfa(ctx, gr);
            if (gfa.rfind("H\tVN:Z:debruijn-rs\n", 0) != 0) return 7;
            size_t n_s = 0, n_l = 0;
            for (size_t p = 0; (p = gfa.find('\n', p)) != std::string::npos; p++) {
                if (p + 1 < gfa.size() && gfa[p + 1] == 'S') n_s++;
                if (p + 1 < gfa.size() && gfa[p + 1] == 'L') n_l++;
            }
            if (n_s != gr.len()) return 7;
            auto ed = edges(ctx, gr);
            size_t want_l = 0;
            for (size_t i = 0; i < ed.size(); i++) {
                for (auto& e : ed[i].first) if (e.node >= i) want_l++;
                for (auto& e : ed[i].second) if (e.node > i) want_l++;
            }
            if (n_l != want_l) return 8;
        }
        // the rank-spanning pipeline through the mirror, one rank (no transport): dbg_seqset_to_device -> dbg_shard_filter_kmers_dev ->
        // dbg_shard_compress_dev == filter_kmers -> compress_kmers_with_hash -> combine -> compress_graph composed call by call
        {
            auto contigs = simple_random_contigs(rng);
            std::vector<std::tuple<DnaString, Exts, uint8_t>> cs;
            for (auto& c : contigs) if (c.size() >= 47) cs.emplace_back(DnaString::from_bytes(c), Exts::empty(), (uint8_t)0);
            Transport none;
            dbg_shard_stats st{};
            auto g1 = sharded_pipeline<Kmer<47>>(ctx, none, cs, CountFilter(1), false, SimpleCompress(), SimpleCompress(Reduce::Max), DBG_REDUCE_GATHER, 0, &st);
            auto fk = filter_kmers<Kmer<47>>(ctx, cs, CountFilter(1), false, false, 4);
            auto g0 = compress_kmers_with_hash<Kmer<47>>(ctx, false, SimpleCompress(), fk.first);
            auto g2 = compress_graph<Kmer<47>, uint16_t>(ctx, false, SimpleCompress(Reduce::Max), combine<Kmer<47>, uint16_t>(ctx, {g0}));
            if (g1.len() == 0 || g1.len() != g2.len() || g1.sequences.sequence.len != g2.sequences.sequence.len) return 70;
            if (g1.sequences.start != g2.sequences.start || g1.sequences.length != g2.sequences.length || g1.data != g2.data) return 71;
            for (size_t i = 0; i < g1.len(); i++) if (g1.exts[i].val != g2.exts[i].val) return 72;
            const size_t nw = (g1.sequences.sequence.len + 31) / 32;
            for (size_t i = 0; i < nw; i++) if (g1.sequences.sequence.storage[i] != g2.sequences.sequence.storage[i]) return 73;
            if (st.total_kmers == 0 || st.total_kmers != st.local_kmers) return 74;
        }
        // the reference would panic on memory_size = 0 (filter.rs:158)
        bool threw = false;
        try { filter_kmers<Kmer<31>>(ctx, seqs, CountFilter(1), false, false, 0); } catch (const Panic&) { threw = true; }
        if (!threw) return 5;
    } catch (const Panic& e) { fprintf(stderr, "panic: %s\n", e.what()); return 9; }
    printf("cpp mirror ok\n");
    return 0;
}
