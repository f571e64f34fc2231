// debruijn_mi355x.hpp -- C++ host-side mirror of the `debruijn` crate's interface for the hot path,
// over the C ABI in dbg_mi355x.h.  The crate is Rust (no toolchain in the build image), so this header
// plays the part of the Rust shim: same names, argument meaning and error behaviour
//     filter_kmers<K>(seqs, summarizer, stranded, report_all_kmers, memory_size)     src/filter.rs:139-148
//     msp_sequence<P>(k, seq, permutation, rc)                                        src/msp.rs:279-288
//     compress_kmers_with_hash<K>(stranded, spec, index)                              src/compression.rs:588-594
//     compress_graph<K>(stranded, spec, graph, censor_nodes), BaseGraph::combine      src/compression.rs:338-349, graph.rs:71-100
// K and P are type parameters in the reference and template parameters here (Kmer<31>, Kmer<47>, ...).
// Where the reference panics, these functions throw debruijn::Panic.  All computation happens on the GPU
// behind the C ABI; this header only flattens and un-flattens plain data.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>
#include "dbg_mi355x.h"

namespace debruijn {

struct Panic : std::runtime_error { using std::runtime_error::runtime_error; };

enum class Dir { Left = 0, Right = 1 };

inline uint8_t base_to_bits(char c) {                        // lib.rs:65-73
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2;
                 case 'T': case 't': return 3; default: return 0; }
}
inline char bits_to_base(uint8_t b) { return b < 4 ? "ACGT"[b] : 'X'; }   // lib.rs:94-102

struct Exts {                                                // lib.rs:577-749
    uint8_t val = 0;
    Exts() = default;
    explicit Exts(uint8_t v) : val(v) {}
    static Exts empty() { return Exts(0); }
    bool has_ext(Dir d, uint8_t b) const { return (val >> (4 * (int)d + b)) & 1; }
    uint8_t num_ext_dir(Dir d) const { return (uint8_t)__builtin_popcount((val >> (4 * (int)d)) & 0xf); }
    bool operator==(const Exts& o) const { return val == o.val; }
};

// VarIntKmer<u64|u128, K> (kmer.rs:438-662): right-aligned 2K bits, base 0 most significant
template <int K_>
struct Kmer {
    static_assert(K_ >= 1 && K_ <= 64, "1 <= K <= 64");
    unsigned __int128 storage = 0;
    static constexpr int k() { return K_; }
    static Kmer from_hi_lo(uint64_t hi, uint64_t lo) { Kmer r; r.storage = ((unsigned __int128)hi << 64) | lo; return r; }
    uint64_t hi() const { return (uint64_t)(storage >> 64); }
    uint64_t lo() const { return (uint64_t)storage; }
    uint8_t get(int pos) const { return (uint8_t)((storage >> (2 * (K_ - 1 - pos))) & 3); }     // kmer.rs:574-577
    static Kmer from_ascii(const std::string& s) {                                              // lib.rs:264-276
        if ((int)s.size() < K_) throw Panic("bytes not long enough to form kmer");
        Kmer r;
        for (int i = 0; i < K_; i++) r.storage = (r.storage << 2) | base_to_bits(s[i]);
        return r;
    }
    std::string to_string() const { std::string s; for (int i = 0; i < K_; i++) s.push_back(bits_to_base(get(i))); return s; }
    bool operator<(const Kmer& o) const { return storage < o.storage; }
    bool operator==(const Kmer& o) const { return storage == o.storage; }
};

// DnaString (dna_string.rs:72-76): 32 bases per u64, base i at bits 63-2(i%32)..62-2(i%32)
struct DnaString {
    std::vector<uint64_t> storage;
    size_t len = 0;
    void push(uint8_t v) {                                                                      // dna_string.rs:303-310
        if ((len & 31) == 0) storage.push_back(0);
        storage.back() |= (uint64_t)(v & 3) << (62 - 2 * (len & 31));
        len++;
    }
    uint8_t get(size_t i) const { return (uint8_t)((storage[i >> 5] >> (62 - 2 * (i & 31))) & 3); }
    static DnaString from_dna_string(const std::string& s) { DnaString d; for (char c : s) d.push(base_to_bits(c)); return d; }
    static DnaString from_bytes(const std::vector<uint8_t>& b) { DnaString d; for (uint8_t x : b) d.push(x); return d; }
    std::string to_string() const { std::string s; for (size_t i = 0; i < len; i++) s.push_back(bits_to_base(get(i))); return s; }
};

// PackedDnaStringSet (dna_string.rs:762-822)
struct PackedDnaStringSet {
    DnaString sequence;
    std::vector<uint64_t> start;
    std::vector<uint32_t> length;
    size_t len() const { return start.size(); }
    void add(const DnaString& s) { start.push_back(sequence.len); for (size_t i = 0; i < s.len; i++) sequence.push(s.get(i)); length.push_back((uint32_t)s.len); }
    DnaString get(size_t i) const { DnaString d; for (uint32_t p = 0; p < length[i]; p++) d.push(sequence.get(start[i] + p)); return d; }
};

struct CountFilter { size_t min_kmer_obs; explicit CountFilter(size_t m) : min_kmer_obs(m) {} };         // filter.rs:40-63
struct CountFilterSet { size_t min_kmer_obs; explicit CountFilterSet(size_t m) : min_kmer_obs(m) {} };   // filter.rs:68-101

enum class Reduce { SaturatingAdd = 0, AddMod65535 = 1, Max = 2, WrappingAdd = 4 };
struct SimpleCompress { Reduce f; explicit SimpleCompress(Reduce r = Reduce::SaturatingAdd) : f(r) {} int kind() const { return (int)f; } };  // compression.rs:40-65
struct ScmapCompress { int kind() const { return 3; } };                                                  // compression.rs:68-98

// The BoomHashMap2<K, Exts, DS> contents in the order the reference feeds BoomHashMap2::new (filter.rs:227-230)
template <class K, class DS>
struct KmerIndex {
    std::vector<K> keys;
    std::vector<Exts> exts;
    std::vector<DS> data;
    size_t len() const { return keys.size(); }
};

template <class K, class D>
struct BaseGraph {                                                                                        // graph.rs:43-50
    PackedDnaStringSet sequences;
    std::vector<Exts> exts;
    std::vector<D> data;
    bool stranded = false;
    size_t len() const { return sequences.len(); }
};

class Context {
public:
    explicit Context(int device = 0) { if (dbg_ctx_create(device, &c_)) throw Panic(dbg_last_error(nullptr)); }
    ~Context() { dbg_ctx_destroy(c_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    dbg_ctx* raw() const { return c_; }
    void check(int rc) const { if (rc) throw Panic(dbg_last_error(c_)); }
private:
    dbg_ctx* c_ = nullptr;
};

namespace detail {
template <class V>
struct Flat {                      // &[(V, Exts, D1)] -> dbg_seqset (PackedDnaStringSet layout)
    PackedDnaStringSet set;
    std::vector<uint8_t> exts, data;
    dbg_seqset ss{};
    template <class Seqs> explicit Flat(const Seqs& seqs) {
        for (auto& t : seqs) { set.add(std::get<0>(t)); exts.push_back(std::get<1>(t).val); data.push_back(std::get<2>(t)); }
        set.sequence.storage.push_back(0);
        ss.words = set.sequence.storage.data(); ss.n_words = set.sequence.storage.size();
        ss.start = set.start.data(); ss.length = set.length.data(); ss.exts = exts.data();
        ss.data = data.data(); ss.data_width = 1; ss.n_seqs = set.len();
    }
};
template <class K, class D>
inline void graph_from_c(const dbg_graph& g, BaseGraph<K, D>* out) {
    out->sequences.sequence.storage.assign(g.seq_words, g.seq_words + g.n_seq_words);
    out->sequences.sequence.len = g.seq_len_bases;
    out->sequences.start.assign(g.start, g.start + g.n_nodes);
    out->sequences.length.assign(g.length, g.length + g.n_nodes);
    out->exts.clear(); out->data.clear();
    for (uint64_t i = 0; i < g.n_nodes; i++) { out->exts.push_back(Exts(g.exts[i])); out->data.push_back((D)g.data[i]); }
    out->stranded = g.stranded != 0;
}
struct CGraph {                    // BaseGraph -> dbg_graph view over temporaries
    std::vector<uint64_t> words; std::vector<uint8_t> exts; std::vector<uint32_t> data; dbg_graph g{};
    template <class K, class D> explicit CGraph(const BaseGraph<K, D>& b) {
        words = b.sequences.sequence.storage; words.push_back(0); words.push_back(0);
        for (auto& e : b.exts) exts.push_back(e.val);
        for (auto& d : b.data) data.push_back((uint32_t)d);
        g.n_nodes = b.len(); g.seq_words = words.data(); g.n_seq_words = b.sequences.sequence.storage.size();
        g.seq_len_bases = b.sequences.sequence.len; g.start = const_cast<uint64_t*>(b.sequences.start.data());
        g.length = const_cast<uint32_t*>(b.sequences.length.data()); g.exts = exts.data(); g.data = data.data();
        g.stranded = b.stranded;
    }
};
}  // namespace detail

// filter_kmers::<K, DnaString, u8, u16, CountFilter> (filter.rs:139-231)
template <class K>
std::pair<KmerIndex<K, uint16_t>, std::vector<K>> filter_kmers(Context& ctx, const std::vector<std::tuple<DnaString, Exts, uint8_t>>& seqs,
                                                               const CountFilter& summarizer, bool stranded, bool report_all_kmers,
                                                               size_t memory_size) {
    detail::Flat<DnaString> f(seqs);
    dbg_filter_params p{(uint32_t)K::k(), stranded, DBG_COUNT_FILTER, summarizer.min_kmer_obs, report_all_kmers, memory_size};
    dbg_kmer_table t{};
    ctx.check(dbg_filter_kmers(ctx.raw(), &f.ss, &p, &t));
    KmerIndex<K, uint16_t> idx;
    std::vector<K> all;
    for (uint64_t i = 0; i < t.n; i++) { idx.keys.push_back(K::from_hi_lo(t.key_hi[i], t.key_lo[i])); idx.exts.push_back(Exts(t.exts[i])); idx.data.push_back(t.count[i]); }
    for (uint64_t i = 0; i < t.n_all; i++) all.push_back(K::from_hi_lo(t.all_hi[i], t.all_lo[i]));
    dbg_free_table(ctx.raw(), &t);
    return {std::move(idx), std::move(all)};
}

// filter_kmers::<K, DnaString, u8, Vec<u8>, CountFilterSet<u8>>
template <class K>
std::pair<KmerIndex<K, std::vector<uint8_t>>, std::vector<K>> filter_kmers(Context& ctx, const std::vector<std::tuple<DnaString, Exts, uint8_t>>& seqs,
                                                                           const CountFilterSet& summarizer, bool stranded,
                                                                           bool report_all_kmers, size_t memory_size) {
    detail::Flat<DnaString> f(seqs);
    dbg_filter_params p{(uint32_t)K::k(), stranded, DBG_COUNT_FILTER_SET, summarizer.min_kmer_obs, report_all_kmers, memory_size};
    dbg_kmer_table t{};
    ctx.check(dbg_filter_kmers(ctx.raw(), &f.ss, &p, &t));
    KmerIndex<K, std::vector<uint8_t>> idx;
    std::vector<K> all;
    for (uint64_t i = 0; i < t.n; i++) {
        idx.keys.push_back(K::from_hi_lo(t.key_hi[i], t.key_lo[i])); idx.exts.push_back(Exts(t.exts[i]));
        std::vector<uint8_t> v;
        for (uint64_t j = t.set_off[i]; j < t.set_off[i + 1]; j++) v.push_back((uint8_t)t.set_val[j]);
        idx.data.push_back(std::move(v));
    }
    for (uint64_t i = 0; i < t.n_all; i++) all.push_back(K::from_hi_lo(t.all_hi[i], t.all_lo[i]));
    dbg_free_table(ctx.raw(), &t);
    return {std::move(idx), std::move(all)};
}

// msp_sequence::<P, DnaString> (msp.rs:279-324); P is the p-mer length
template <int P>
std::vector<std::tuple<uint32_t, Exts, DnaString>> msp_sequence(Context& ctx, size_t k, const std::vector<uint8_t>& seq,
                                                                const std::vector<uint32_t>* permutation, bool rc) {
    DnaString d = DnaString::from_bytes(seq);
    d.storage.push_back(0);
    uint64_t start = 0; uint32_t length = (uint32_t)seq.size();
    dbg_seqset ss{d.storage.data(), d.storage.size(), &start, &length, nullptr, nullptr, 0, 1};
    dbg_msp_params p{(uint32_t)k, (uint32_t)P, permutation ? permutation->data() : nullptr, rc, 0};
    dbg_msp_pieces pc{};
    ctx.check(dbg_msp_sequence(ctx.raw(), &ss, &p, &pc));
    std::vector<std::tuple<uint32_t, Exts, DnaString>> out;
    for (uint64_t i = 0; i < pc.n_pieces; i++) {
        DnaString v;
        for (uint32_t j = 0; j < pc.len[i]; j++) v.push(seq[pc.start[i] + j]);
        out.emplace_back(pc.bucket[i], Exts(pc.exts[i]), std::move(v));
    }
    dbg_free_pieces(ctx.raw(), &pc);
    return out;
}

// compress_kmers_with_hash::<K, u16, S> (compression.rs:588-594); seed_order = the MPHF slot order when known
template <class K, class S>
BaseGraph<K, uint16_t> compress_kmers_with_hash(Context& ctx, bool stranded, const S& spec, const KmerIndex<K, uint16_t>& index,
                                                const std::vector<uint64_t>* seed_order = nullptr) {
    std::vector<uint64_t> hi, lo; std::vector<uint8_t> ex; std::vector<uint32_t> da;
    for (size_t i = 0; i < index.len(); i++) { hi.push_back(index.keys[i].hi()); lo.push_back(index.keys[i].lo()); ex.push_back(index.exts[i].val); da.push_back(index.data[i]); }
    dbg_graph g{};
    ctx.check(dbg_compress_kmers_with_hash(ctx.raw(), (uint32_t)K::k(), stranded, spec.kind(), index.len(), hi.data(), lo.data(), ex.data(),
                                           da.data(), seed_order ? seed_order->data() : nullptr, &g));
    BaseGraph<K, uint16_t> out;
    detail::graph_from_c(g, &out);
    dbg_free_graph(ctx.raw(), &g);
    return out;
}

// compress_kmers_no_exts::<K, u16, S> (compression.rs:619-659): kmer_exts = &[(K, D)]
template <class K, class S>
BaseGraph<K, uint16_t> compress_kmers_no_exts(Context& ctx, bool stranded, const S& spec, const std::vector<std::pair<K, uint16_t>>& kmer_exts) {
    std::vector<uint64_t> hi, lo; std::vector<uint32_t> da;
    for (auto& kd : kmer_exts) { hi.push_back(kd.first.hi()); lo.push_back(kd.first.lo()); da.push_back(kd.second); }
    dbg_graph g{};
    ctx.check(dbg_compress_kmers_no_exts(ctx.raw(), (uint32_t)K::k(), stranded, spec.kind(), kmer_exts.size(), hi.data(), lo.data(), da.data(), nullptr, &g));
    BaseGraph<K, uint16_t> out;
    detail::graph_from_c(g, &out);
    dbg_free_graph(ctx.raw(), &g);
    return out;
}

// BaseGraph::combine (graph.rs:71-100)
template <class K, class D>
BaseGraph<K, D> combine(Context& ctx, const std::vector<BaseGraph<K, D>>& graphs) {
    std::vector<detail::CGraph> cg; cg.reserve(graphs.size());
    std::vector<dbg_graph> raw;
    for (auto& g : graphs) { cg.emplace_back(g); }
    for (auto& c : cg) raw.push_back(c.g);
    dbg_graph out{};
    ctx.check(dbg_graph_combine(ctx.raw(), raw.data(), (uint32_t)raw.size(), &out));
    BaseGraph<K, D> r;
    detail::graph_from_c(out, &r);
    dbg_free_graph(ctx.raw(), &out);
    return r;
}

// DebruijnGraph::write_gfa (graph.rs:603-616) as a string, and to_gfa (graph.rs:598-601)
template <class K, class D>
std::string write_gfa(Context& ctx, const BaseGraph<K, D>& graph) {
    detail::CGraph cg(graph);
    char* t = nullptr; uint64_t n = 0;
    ctx.check(dbg_graph_to_gfa(ctx.raw(), (uint32_t)K::k(), &cg.g, &t, &n));
    std::string s(t, n);
    dbg_free_text(t);
    return s;
}
template <class K, class D>
void to_gfa(Context& ctx, const BaseGraph<K, D>& graph, const std::string& path) {
    detail::CGraph cg(graph);
    ctx.check(dbg_graph_write_gfa(ctx.raw(), (uint32_t)K::k(), &cg.g, path.c_str()));
}

// Node::l_edges / r_edges (graph.rs:1041-1049): (target node, incoming side of the target: 0 = Left, 1 = Right, flipped)
struct Edge { size_t node; int dir; bool flip; };
template <class K, class D>
std::vector<std::pair<std::vector<Edge>, std::vector<Edge>>> edges(Context& ctx, const BaseGraph<K, D>& graph) {
    detail::CGraph cg(graph);
    dbg_edges e{};
    ctx.check(dbg_graph_edges(ctx.raw(), (uint32_t)K::k(), &cg.g, &e));
    std::vector<std::pair<std::vector<Edge>, std::vector<Edge>>> out(e.n_nodes);
    for (uint64_t i = 0; i < e.n_nodes; i++)
        for (int b = 0; b < 8; b++) {
            uint32_t t = e.target[i * 8 + b];
            if (t == 0xFFFFFFFFu) continue;
            Edge ed{t, e.info[i * 8 + b] & 1, (e.info[i * 8 + b] & 2) != 0};
            (b < 4 ? out[i].first : out[i].second).push_back(ed);
        }
    dbg_free_edges(&e);
    return out;
}

// DnaString::from_acgt_bytes (dna_string.rs:222-250), packed on the device
inline DnaString from_acgt_bytes(Context& ctx, const std::string& ascii) {
    DnaString d;
    d.storage.assign((ascii.size() + 31) / 32, 0);
    d.len = ascii.size();
    ctx.check(dbg_pack_acgt(ctx.raw(), (const uint8_t*)ascii.data(), ascii.size(), d.storage.data(), nullptr));
    return d;
}

// DnaString::from_acgt_bytes_hashn (dna_string.rs:255-278): non-ACGT -> DefaultHasher(read_name, pos) % 4
inline DnaString from_acgt_bytes_hashn(Context& ctx, const std::string& ascii, const std::string& read_name) {
    DnaString d;
    d.storage.assign((ascii.size() + 31) / 32, 0);
    d.len = ascii.size();
    const uint64_t seq_off[2] = {0, ascii.size()}, name_off[2] = {0, read_name.size()};
    ctx.check(dbg_pack_acgt_hashn(ctx.raw(), (const uint8_t*)ascii.data(), seq_off, 1, (const uint8_t*)read_name.data(), name_off,
                                  d.storage.data(), nullptr));
    return d;
}

// ---- the rank-spanning flow (one process per GPU): src/test.rs:433-470 across ranks, two calls into the library -------------
// A Transport wraps the C table of collective operations: Transport::rccl(comm, rank, world) for a host that made its ncclComm_t
// with ncclCommInitRank, or none (one rank).  sharded_pipeline = filter_kmers over the reads of ALL ranks (this rank passes its own)
// -> per-shard compress_kmers_with_hash -> BaseGraph::combine -> compress_graph; the final graph arrives on `root` (empty elsewhere).
struct Transport {
    dbg_transport* t = nullptr;
    Transport() {}
    static Transport rccl(void* nccl_comm, int rank, int world, const char* librccl_path = nullptr) {
        Transport tr;
        char err[512] = {0};
        if (dbg_transport_rccl_create(nccl_comm, rank, world, librccl_path, &tr.t, err, sizeof(err))) throw Panic(err);
        return tr;
    }
    Transport(Transport&& o) noexcept : t(o.t) { o.t = nullptr; }
    Transport(const Transport&) = delete;
    ~Transport() { if (t) dbg_transport_destroy(t); }
    int rank() const { return t ? t->rank : 0; }
    int world() const { return t ? t->world : 1; }
};

template <class K, class S1, class S2>
BaseGraph<K, uint16_t> sharded_pipeline(Context& ctx, const Transport& tr, const std::vector<std::tuple<DnaString, Exts, uint8_t>>& my_seqs,
                                        const CountFilter& summarizer, bool stranded, const S1& spec, const S2& second_spec,
                                        int reduce = DBG_REDUCE_GATHER, int root = 0, dbg_shard_stats* stats = nullptr) {
    detail::Flat<DnaString> f(my_seqs);
    f.ss.data = nullptr; f.ss.data_width = 0;                      // CountFilter ignores D1 (filter.rs:52-62)
    dbg_seqset dev{};
    ctx.check(dbg_seqset_to_device(ctx.raw(), &f.ss, &dev));
    dbg_shard_params p{(uint32_t)K::k(), stranded, DBG_COUNT_FILTER, summarizer.min_kmer_obs, 0, -1, 1, 0};
    dbg_kmer_table t{};
    int rc = dbg_shard_filter_kmers_dev(ctx.raw(), tr.t, &dev, &p, &t, stats);
    dbg_seqset_free_device(ctx.raw(), &dev);
    ctx.check(rc);
    dbg_graph fin{};
    rc = dbg_shard_compress_dev(ctx.raw(), tr.t, (uint32_t)K::k(), stranded, spec.kind(), second_spec.kind(), &t, reduce, root, &fin, nullptr, nullptr);
    dbg_free_table(ctx.raw(), &t);
    ctx.check(rc);
    BaseGraph<K, uint16_t> out;
    out.stranded = stranded;
    if (tr.rank() == root) detail::graph_from_c(fin, &out);
    dbg_free_graph(ctx.raw(), &fin);
    return out;
}

// compress_graph (compression.rs:338-349)
template <class K, class D, class S>
BaseGraph<K, D> compress_graph(Context& ctx, bool stranded, const S& spec, const BaseGraph<K, D>& old_graph,
                               const std::vector<uint64_t>* censor_nodes = nullptr) {
    detail::CGraph cg(old_graph);
    dbg_graph out{};
    ctx.check(dbg_compress_graph(ctx.raw(), (uint32_t)K::k(), stranded, spec.kind(), &cg.g, censor_nodes ? censor_nodes->data() : nullptr,
                                 censor_nodes ? censor_nodes->size() : 0, &out));
    BaseGraph<K, D> r;
    detail::graph_from_c(out, &r);
    dbg_free_graph(ctx.raw(), &out);
    return r;
}

}  // namespace debruijn
