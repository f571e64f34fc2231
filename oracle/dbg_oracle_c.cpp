// =============================================================================
// dbg_oracle_c.cpp -- TEST INFRASTRUCTURE ONLY.  Flat C entry points over the
// CPU oracle (dbg_oracle.hpp) so that pytest / bench.py's cpu_baseline leg can
// drive it through ctypes.  Never linked into the product library.
// =============================================================================
#include "dbg_oracle.hpp"
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include <chrono>
#include <thread>
#include <atomic>

using namespace orc;

namespace {
thread_local std::string g_err;
static inline uint64_t lo64(Kmer k) { return (uint64_t)k.storage; }
static inline uint64_t hi64(Kmer k) { return (uint64_t)(k.storage >> 64); }
static inline Kmer mk(uint64_t hi, uint64_t lo) { return Kmer(((u128)hi << 64) | lo); }
}

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

// ---- building blocks (known-answer tests) ---------------------------------
void orc_kmer_from_ascii(int k, const char* s, uint64_t* hi, uint64_t* lo) {     // lib.rs:264-276
    KmerSpec ks(k); Kmer km;
    for (int i = 0; i < k; i++) kmer_set_mut(ks, km, i, base_to_bits((uint8_t)s[i]));
    *hi = hi64(km); *lo = lo64(km);
}
void orc_kmer_to_ascii(int k, uint64_t hi, uint64_t lo, char* out) {
    std::string s = kmer_to_string(KmerSpec(k), mk(hi, lo));
    memcpy(out, s.data(), s.size()); out[s.size()] = 0;
}
void orc_kmer_rc(int k, uint64_t hi, uint64_t lo, uint64_t* ohi, uint64_t* olo) {
    Kmer r = kmer_rc(KmerSpec(k), mk(hi, lo)); *ohi = hi64(r); *olo = lo64(r);
}
void orc_kmer_extend(int k, uint64_t hi, uint64_t lo, int base, int dir, uint64_t* ohi, uint64_t* olo) {
    Kmer r = kmer_extend(KmerSpec(k), mk(hi, lo), (uint8_t)base, (Dir)dir); *ohi = hi64(r); *olo = lo64(r);
}
int orc_kmer_min_rc_flip(int k, uint64_t hi, uint64_t lo, uint64_t* ohi, uint64_t* olo) {
    Kmer o; bool f; kmer_min_rc_flip(KmerSpec(k), mk(hi, lo), o, f); *ohi = hi64(o); *olo = lo64(o); return f ? 1 : 0;
}
int orc_kmer_is_palindrome(int k, uint64_t hi, uint64_t lo) { return kmer_is_palindrome(KmerSpec(k), mk(hi, lo)) ? 1 : 0; }
int orc_kmer_get(int k, uint64_t hi, uint64_t lo, int pos) { return kmer_get(KmerSpec(k), mk(hi, lo), pos); }
uint8_t orc_exts_rc(uint8_t v) { return Exts(v).rc().val; }
uint8_t orc_exts_complement(uint8_t v) { return Exts(v).complement().val; }

// DnaString::from_bytes (0-3 bytes) -> packed words; returns number of words written
uint64_t orc_dnastring_pack(const uint8_t* bases, uint64_t n, uint64_t* words_out, uint64_t cap) {
    DnaString d = DnaString::from_bytes(bases, n);
    uint64_t nw = d.storage.size();
    for (uint64_t i = 0; i < nw && i < cap; i++) words_out[i] = d.storage[i];
    return nw;
}
// push base-by-base (PackedDnaStringSet::add path, dna_string.rs:811-821)
uint64_t orc_dnastring_push(const uint8_t* bases, uint64_t n, uint64_t* words_out, uint64_t cap) {
    DnaString d;
    for (uint64_t i = 0; i < n; i++) d.push(bases[i]);
    uint64_t nw = d.storage.size();
    for (uint64_t i = 0; i < nw && i < cap; i++) words_out[i] = d.storage[i];
    return nw;
}
void orc_seq_get_kmer(const uint64_t* words, uint64_t start, uint64_t length, uint64_t pos, int k, uint64_t* hi, uint64_t* lo) {
    Kmer km = seq_get_kmer(KmerSpec(k), SeqView{words, (size_t)start, (size_t)length}, pos);
    *hi = hi64(km); *lo = lo64(km);
}
int orc_seq_get(const uint64_t* words, uint64_t pos) { return SeqView{words, 0, (size_t)pos + 1}.get(pos); }

// ---- filter_kmers ---------------------------------------------------------
struct OrcTable { KmerTable t; };

void* orc_filter_kmers(const uint64_t* words, const uint64_t* start, const uint32_t* length,
                       const uint8_t* exts, const uint32_t* data, uint64_t n_seqs, uint32_t sizeof_d1,
                       int k, int summarizer, uint64_t min_obs, int stranded, int report_all, uint64_t memory_size) {
    SeqSet s{words, start, length, exts, data, (size_t)n_seqs, (size_t)sizeof_d1};
    OrcTable* h = new OrcTable();
    int r = filter_kmers(s, k, (Summarizer)summarizer, (size_t)min_obs, stranded != 0, report_all != 0,
                         (size_t)memory_size, h->t, g_err);
    if (r) { delete h; return nullptr; }
    return h;
}
uint64_t orc_table_len(void* h) { return ((OrcTable*)h)->t.keys.size(); }
uint64_t orc_table_all_len(void* h) { return ((OrcTable*)h)->t.all_kmers.size(); }
uint64_t orc_table_setval_len(void* h) { return ((OrcTable*)h)->t.set_val.size(); }
uint32_t orc_table_passes(void* h) { return ((OrcTable*)h)->t.n_passes; }
void orc_table_copy(void* h, uint64_t* key_hi, uint64_t* key_lo, uint8_t* exts, uint16_t* count,
                    uint64_t* set_off, uint32_t* set_val, uint64_t* all_hi, uint64_t* all_lo) {
    KmerTable& t = ((OrcTable*)h)->t;
    for (size_t i = 0; i < t.keys.size(); i++) {
        if (key_hi) key_hi[i] = hi64(t.keys[i]);
        if (key_lo) key_lo[i] = lo64(t.keys[i]);
        if (exts) exts[i] = t.exts[i].val;
        if (count && i < t.count.size()) count[i] = t.count[i];
    }
    if (set_off) for (size_t i = 0; i < t.set_off.size(); i++) set_off[i] = t.set_off[i];
    if (set_val) for (size_t i = 0; i < t.set_val.size(); i++) set_val[i] = t.set_val[i];
    for (size_t i = 0; i < t.all_kmers.size(); i++) {
        if (all_hi) all_hi[i] = hi64(t.all_kmers[i]);
        if (all_lo) all_lo[i] = lo64(t.all_kmers[i]);
    }
}
void orc_table_free(void* h) { delete (OrcTable*)h; }

// remove_censored_exts(_sharded) on sorted arrays, in place on exts
void orc_remove_censored_exts(int k, int stranded, uint64_t n, const uint64_t* key_hi, const uint64_t* key_lo,
                              uint8_t* exts, uint64_t n_all, const uint64_t* all_hi, const uint64_t* all_lo, int sharded) {
    std::vector<Kmer> keys(n), all(n_all); std::vector<Exts> e(n);
    for (uint64_t i = 0; i < n; i++) { keys[i] = mk(key_hi ? key_hi[i] : 0, key_lo[i]); e[i] = Exts(exts[i]); }
    for (uint64_t i = 0; i < n_all; i++) all[i] = mk(all_hi ? all_hi[i] : 0, all_lo[i]);
    if (sharded) remove_censored_exts_sharded(KmerSpec(k), stranded != 0, keys, e, all);
    else remove_censored_exts(KmerSpec(k), stranded != 0, keys, e);
    for (uint64_t i = 0; i < n; i++) exts[i] = e[i].val;
}

// ---- MSP ------------------------------------------------------------------
// Scanner::scan: fills (minimizer, start, len, minimizer_pos); returns count or -1
int64_t orc_msp_scan(const uint8_t* seq, uint64_t m, int k, int p, const uint64_t* perm, int rc, int score_mode,
                     uint64_t* minimizer, uint32_t* start, uint16_t* len, uint32_t* mpos, uint64_t cap) {
    std::vector<MspInterval> iv;
    if (msp_scan(seq, m, k, p, perm, rc != 0, (ScoreMode)score_mode, iv, g_err)) return -1;
    for (size_t i = 0; i < iv.size() && i < cap; i++) {
        minimizer[i] = iv[i].minimizer; start[i] = iv[i].start; len[i] = iv[i].len; mpos[i] = iv[i].minimizer_pos;
    }
    return (int64_t)iv.size();
}
// msp_sequence: (bucket, exts, start, len) + optional Lmer words (n_lmer_words per piece; 0 = skip)
int64_t orc_msp_sequence(const uint8_t* seq, uint64_t m, int k, int p, const uint64_t* perm, int rc,
                         uint64_t vmer_max_len, uint32_t* bucket, uint8_t* exts, uint32_t* start, uint16_t* len,
                         uint64_t* lmer_words, int n_lmer_words, uint64_t cap) {
    std::vector<MspPiece> pc;
    if (msp_sequence(seq, m, k, p, perm, rc != 0, (size_t)vmer_max_len, pc, g_err)) return -1;
    for (size_t i = 0; i < pc.size() && i < cap; i++) {
        bucket[i] = pc[i].bucket; exts[i] = pc[i].exts.val; start[i] = pc[i].start; len[i] = pc[i].len;
        if (lmer_words && n_lmer_words > 0) lmer_from_slice(seq + pc[i].start, pc[i].len, n_lmer_words, lmer_words + i * n_lmer_words);
    }
    return (int64_t)pc.size();
}

// ---- compress -------------------------------------------------------------
struct OrcGraph { BaseGraph g; DebruijnGraph dbg; bool finished = false; int k = 0; };

void* orc_compress_kmers(int k, int stranded, int spec, uint64_t n, const uint64_t* key_hi, const uint64_t* key_lo,
                         const uint8_t* exts, const uint32_t* data, const uint64_t* seed_order) {
    std::vector<Kmer> keys(n); std::vector<Exts> e(n); std::vector<uint32_t> d(n);
    for (uint64_t i = 0; i < n; i++) { keys[i] = mk(key_hi ? key_hi[i] : 0, key_lo[i]); e[i] = Exts(exts[i]); d[i] = data ? data[i] : 0; }
    OrcGraph* h = new OrcGraph(); h->k = k;
    if (compress_kmers_with_hash(k, stranded != 0, (Spec)spec, keys, e, d, seed_order, h->g, g_err)) { delete h; return nullptr; }
    return h;
}
void* orc_compress_kmers_no_exts(int k, int stranded, int spec, uint64_t n, const uint64_t* key_hi, const uint64_t* key_lo,
                                 const uint32_t* data, const uint64_t* seed_order, uint8_t* exts_out) {
    std::vector<Kmer> keys(n); std::vector<uint32_t> d(n); std::vector<Exts> e;
    for (uint64_t i = 0; i < n; i++) { keys[i] = mk(key_hi ? key_hi[i] : 0, key_lo[i]); d[i] = data ? data[i] : 0; }
    OrcGraph* h = new OrcGraph(); h->k = k;
    if (compress_kmers_no_exts(k, stranded != 0, (Spec)spec, keys, d, seed_order, h->g, &e, g_err)) { delete h; return nullptr; }
    if (exts_out) for (uint64_t i = 0; i < n; i++) exts_out[i] = e[i].val;
    return h;
}
static BaseGraph& bg(void* h) { OrcGraph* g = (OrcGraph*)h; return g->finished ? g->dbg.base : g->g; }
uint64_t orc_graph_len(void* h) { return bg(h).len(); }
uint64_t orc_graph_n_words(void* h) { return bg(h).sequences.sequence.storage.size(); }
uint64_t orc_graph_n_bases(void* h) { return bg(h).sequences.sequence.len; }
void orc_graph_copy(void* h, uint64_t* words, uint64_t* start, uint32_t* length, uint8_t* exts, uint32_t* data) {
    BaseGraph& g = bg(h);
    auto& st = g.sequences.sequence.storage;
    if (words) for (size_t i = 0; i < st.size(); i++) words[i] = st[i];
    for (size_t i = 0; i < g.len(); i++) {
        if (start) start[i] = g.sequences.start[i];
        if (length) length[i] = g.sequences.length[i];
        if (exts) exts[i] = g.exts[i].val;
        if (data) data[i] = g.data[i];
    }
}
void orc_graph_free(void* h) { delete (OrcGraph*)h; }

// build a BaseGraph from flat arrays (node sequences given as PackedDnaStringSet layout)
void* orc_graph_from_arrays(int k, int stranded, uint64_t n_nodes, const uint64_t* words, const uint64_t* start,
                            const uint32_t* length, const uint8_t* exts, const uint32_t* data) {
    OrcGraph* h = new OrcGraph(); h->k = k; h->g.stranded = stranded != 0;
    for (uint64_t i = 0; i < n_nodes; i++) {
        SeqView v{words, (size_t)start[i], (size_t)length[i]};
        std::vector<uint8_t> b(v.length);
        for (size_t j = 0; j < v.length; j++) b[j] = v.get(j);
        h->g.sequences.add(b.begin(), b.end());
        h->g.exts.push_back(Exts(exts[i]));
        h->g.data.push_back(data ? data[i] : 0);
    }
    return h;
}
// BaseGraph::combine over handles -> new handle
void* orc_graph_combine(void** hs, uint64_t n) {
    std::vector<BaseGraph> gs;
    int k = 0;
    for (uint64_t i = 0; i < n; i++) { gs.push_back(bg(hs[i])); k = ((OrcGraph*)hs[i])->k; }
    OrcGraph* h = new OrcGraph(); h->k = k;
    if (graph_combine(gs, h->g, g_err)) { delete h; return nullptr; }
    return h;
}
// finish() in place
void orc_graph_finish(void* h) {
    OrcGraph* g = (OrcGraph*)h;
    if (!g->finished) { graph_finish(g->k, std::move(g->g), g->dbg); g->finished = true; }
}
// is_compressed: 1 = None (compressed), 0 = Some(a,b)
int orc_graph_is_compressed(void* h, int spec, uint64_t* a, uint64_t* b) {
    orc_graph_finish(h);
    size_t x = 0, y = 0;
    bool ok = graph_is_compressed(((OrcGraph*)h)->dbg, (Spec)spec, x, y);
    if (a) *a = x;
    if (b) *b = y;
    return ok ? 1 : 0;
}
// compress_graph(stranded, spec, finished graph, censor) -> new finished handle; consumes nothing (copies)
void* orc_compress_graph(void* h, int stranded, int spec, const uint64_t* censor, uint64_t n_censor) {
    orc_graph_finish(h);
    OrcGraph* src = (OrcGraph*)h;
    DebruijnGraph copy = src->dbg;
    std::vector<size_t> c;
    if (censor) for (uint64_t i = 0; i < n_censor; i++) c.push_back((size_t)censor[i]);
    OrcGraph* out = new OrcGraph(); out->k = src->k; out->finished = true;
    if (compress_graph(stranded != 0, (Spec)spec, std::move(copy), censor ? &c : nullptr, out->dbg, g_err)) { delete out; return nullptr; }
    return out;
}

// write_gfa -> copies min(cap, len) bytes, returns len
uint64_t orc_graph_gfa(void* h, char* buf, uint64_t cap) {
    orc_graph_finish(h);
    std::string t = graph_write_gfa(((OrcGraph*)h)->dbg);
    if (buf) memcpy(buf, t.data(), std::min<uint64_t>(cap, t.size()));
    return t.size();
}
// from_acgt_bytes -> storage words (ceil(n/32)); to_ascii_vec of a packed range
void orc_pack_acgt(const uint8_t* ascii, uint64_t n, uint64_t* words) {
    DnaString d = DnaString::from_acgt_bytes(ascii, n);
    for (size_t i = 0; i < d.storage.size(); i++) words[i] = d.storage[i];
}
void orc_pack_acgt_hashn(const uint8_t* ascii, uint64_t n, const uint8_t* name, uint64_t name_len, uint64_t* words) {
    DnaString d = DnaString::from_acgt_bytes_hashn(ascii, n, name, name_len);
    for (size_t i = 0; i < d.storage.size(); i++) words[i] = d.storage[i];
}
uint64_t orc_siphash(int c, int d, uint64_t k0, uint64_t k1, const uint8_t* msg, uint64_t len) { return siphash(c, d, k0, k1, msg, len); }
void orc_unpack_acgt(const uint64_t* words, uint64_t first_base, uint64_t n, uint8_t* ascii) {
    SeqView v{words, (size_t)first_base, (size_t)n};
    for (uint64_t i = 0; i < n; i++) ascii[i] = (uint8_t)"ACGT"[v.get(i)];
}

// ---- timing helper for bench.py's cpu_baseline leg: runs filter_kmers and returns seconds ----
double orc_time_filter_kmers(const uint64_t* words, const uint64_t* start, const uint32_t* length,
                             const uint8_t* exts, const uint32_t* data, uint64_t n_seqs, uint32_t sizeof_d1,
                             int k, int summarizer, uint64_t min_obs, int stranded, uint64_t memory_size,
                             uint64_t* n_valid_out) {
    SeqSet s{words, start, length, exts, data, (size_t)n_seqs, (size_t)sizeof_d1};
    KmerTable t;
    auto t0 = std::chrono::steady_clock::now();
    int r = filter_kmers(s, k, (Summarizer)summarizer, (size_t)min_obs, stranded != 0, false, (size_t)memory_size, t, g_err);
    auto t1 = std::chrono::steady_clock::now();
    if (n_valid_out) *n_valid_out = t.keys.size();
    if (r) return -1.0;
    return std::chrono::duration<double>(t1 - t0).count();
}

// ---- many-core CPU baseline: the parallelisation downstream users put around the crate ----
// (the crate itself has no parallel path).  It is the sharded pipeline of the reference's own end-to-end test
// (test.rs:418-504): every read is cut by msp_sequence (msp.rs:279-324, p-mer order = identity permutation) into
// pieces whose k-mers share a minimizer bucket, the pieces are grouped into shards by bucket, and filter_kmers
// (filter.rs:139-231) runs on each shard independently.  Phase 1 threads over read slices, phase 2 threads over
// shards; returns wall seconds, *n_valid_out = total valid k-mers over the shards (equals the unsharded count).
double orc_time_filter_kmers_sharded_mt(const uint64_t* words, const uint64_t* start, const uint32_t* length,
                                        const uint32_t* data, uint64_t n_seqs, uint32_t sizeof_d1, int k, int p,
                                        int summarizer, uint64_t min_obs, int stranded, uint32_t n_threads,
                                        uint32_t n_shards, uint64_t* n_valid_out, double* phase1_seconds) {
    if (n_threads < 1) n_threads = 1;
    if (n_shards < 1) n_shards = 1;
    struct Piece { uint64_t start; uint32_t len; uint8_t exts; uint32_t d; };
    std::vector<uint64_t> perm((size_t)1 << (2 * p));
    for (size_t i = 0; i < perm.size(); i++) perm[i] = i;
    std::vector<std::vector<std::vector<Piece>>> lists(n_threads, std::vector<std::vector<Piece>>(n_shards));
    std::atomic<int> failed{0};
    auto t0 = std::chrono::steady_clock::now();
    {
        std::vector<std::thread> th;
        for (uint32_t t = 0; t < n_threads; t++) th.emplace_back([&, t]() {
            const uint64_t lo = n_seqs * t / n_threads, hi = n_seqs * (t + 1) / n_threads;
            std::vector<uint8_t> bases; std::vector<MspPiece> pc; std::string err;
            for (uint64_t i = lo; i < hi; i++) {
                SeqView v{words, (size_t)start[i], (size_t)length[i]};
                if (v.length < (size_t)k) continue;
                bases.resize(v.length);
                for (size_t j = 0; j < v.length; j++) bases[j] = v.get(j);
                pc.clear();
                if (msp_sequence(bases.data(), bases.size(), k, p, perm.data(), stranded == 0, 1u << 20, pc, err)) { failed = 1; return; }
                for (const MspPiece& q : pc)
                    lists[t][q.bucket % n_shards].push_back(Piece{start[i] + q.start, q.len, q.exts.val, data ? data[i] : 0});
            }
        });
        for (auto& x : th) x.join();
    }
    if (phase1_seconds) *phase1_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::atomic<uint32_t> next{0};
    std::atomic<uint64_t> valid{0};
    {
        std::vector<std::thread> th;
        for (uint32_t t = 0; t < n_threads; t++) th.emplace_back([&]() {
            std::vector<uint64_t> st; std::vector<uint32_t> ln, dd; std::vector<uint8_t> ex; std::string err;
            for (;;) {
                const uint32_t sh = next.fetch_add(1);
                if (sh >= n_shards) break;
                st.clear(); ln.clear(); dd.clear(); ex.clear();
                for (uint32_t u = 0; u < n_threads; u++)
                    for (const Piece& q : lists[u][sh]) { st.push_back(q.start); ln.push_back(q.len); ex.push_back(q.exts); dd.push_back(q.d); }
                SeqSet s{words, st.data(), ln.data(), ex.data(), data ? dd.data() : nullptr, st.size(), (size_t)sizeof_d1};
                KmerTable tb;
                if (filter_kmers(s, k, (Summarizer)summarizer, (size_t)min_obs, stranded != 0, false, 1u << 20, tb, err)) { failed = 1; return; }
                valid += tb.keys.size();
            }
        });
        for (auto& x : th) x.join();
    }
    auto t1 = std::chrono::steady_clock::now();
    if (n_valid_out) *n_valid_out = valid.load();
    if (failed) return -1.0;
    return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
