// serde forms of the result type BaseGraph<K, D> -- SURVEY.md section 8(f)-3: the on-disk / wire formats on the far side of the
// path, so that a graph built here can be loaded by the crate (and vice versa).  Host-side formatting only.
//
// The derives (src/graph.rs:43-50 BaseGraph{sequences, exts, data, stranded, phantom}; src/dna_string.rs:762-767
// PackedDnaStringSet{sequence, start, length}; :72-76 DnaString{storage, len}; src/lib.rs:577-580 Exts{val}) fix the field
// order; the two data formats restated here are
//   DBG_SERDE_JSON    serde_json (a dependency of the crate, Cargo.toml:18): structs as objects in field order, Vec as arrays,
//                     PhantomData as null, no whitespace (serde_json::to_string / to_writer);
//   DBG_SERDE_BINCODE bincode 1.x default options (not a dependency of the crate; the usual binary companion of serde derives):
//                     fields in order, little-endian fixed-width integers, usize as u64, Vec = u64 length + elements, bool =
//                     one byte, PhantomData = nothing.
#include "dbg_internal.hpp"
#include <string>
#include <vector>

namespace {
thread_local std::string g_serde_err;
int sfail(dbg_ctx* c, int code, const std::string& m) { if (c) return c->fail(code, m); g_serde_err = m; return code; }

void put_u64(std::vector<uint8_t>& o, uint64_t v) { for (int i = 0; i < 8; i++) o.push_back((uint8_t)(v >> (8 * i))); }
void put_le(std::vector<uint8_t>& o, uint64_t v, int w) { for (int i = 0; i < w; i++) o.push_back((uint8_t)(v >> (8 * i))); }

struct Rd {
    const uint8_t* p; uint64_t n, i = 0; bool bad = false;
    uint64_t le(int w) { if (i + w > n) { bad = true; return 0; } uint64_t v = 0; for (int q = 0; q < w; q++) v |= (uint64_t)p[i + q] << (8 * q); i += w; return v; }
};

// ---- a JSON reader for exactly the shapes serde_json writes for BaseGraph (objects with known keys, arrays of unsigned integers) ----
struct Js {
    const char* p; uint64_t n, i = 0; bool bad = false;
    void ws() { while (i < n && (p[i] == ' ' || p[i] == '\n' || p[i] == '\t' || p[i] == '\r')) i++; }
    bool eat(char ch) { ws(); if (i < n && p[i] == ch) { i++; return true; } bad = true; return false; }
    bool peek(char ch) { ws(); return i < n && p[i] == ch; }
    bool key(const char* k) {
        ws();
        const size_t l = strlen(k);
        if (i + l + 2 <= n && p[i] == '"' && !memcmp(p + i + 1, k, l) && p[i + 1 + l] == '"') { i += l + 2; return eat(':'); }
        bad = true; return false;
    }
    uint64_t num() {
        ws();
        if (i >= n || p[i] < '0' || p[i] > '9') { bad = true; return 0; }
        uint64_t v = 0;
        while (i < n && p[i] >= '0' && p[i] <= '9') {           // serde_json rejects integers beyond the target type: so does this
            const uint64_t d = (uint64_t)(p[i++] - '0');
            if (v > (UINT64_MAX - d) / 10) { bad = true; return 0; }
            v = v * 10 + d;
        }
        if (i < n && (p[i] == '.' || p[i] == 'e' || p[i] == 'E')) { bad = true; return 0; }     // a float is not a u64 for serde either
        return v;
    }
    uint64_t num_max(uint64_t mx) { const uint64_t v = num(); if (v > mx) { bad = true; return 0; } return v; }
    bool lit(const char* s) { ws(); const size_t l = strlen(s); if (i + l <= n && !memcmp(p + i, s, l)) { i += l; return true; } return false; }
    template <class F> void array(F f) {
        if (!eat('[')) return;
        if (peek(']')) { i++; return; }
        do { f(); } while (!bad && peek(',') && ++i);
        eat(']');
    }
};

void json_u64_array(std::string& s, const uint64_t* a, uint64_t n) {
    s += '[';
    char b[24];
    for (uint64_t i = 0; i < n; i++) { if (i) s += ','; s.append(b, (size_t)snprintf(b, sizeof(b), "%llu", (unsigned long long)a[i])); }
    s += ']';
}
}  // namespace

extern "C" const char* dbg_serde_last_error(void) { return g_serde_err.c_str(); }
extern "C" void dbg_free_bytes(uint8_t* p) { free(p); }

extern "C" int dbg_graph_serialize(dbg_ctx* c, const dbg_graph* g, int format, uint32_t data_width, uint8_t** bytes, uint64_t* len) {
    if (!g || !bytes || !len) return sfail(c, 10, "null argument");
    *bytes = nullptr; *len = 0;
    if (!(data_width == 0 || data_width == 1 || data_width == 2 || data_width == 4)) return sfail(c, 170, "data_width must be 0 (unit), 1, 2 or 4");
    const uint64_t n = g->n_nodes, nw = (g->seq_len_bases + 31) / 32;
    if (nw > g->n_seq_words) return sfail(c, 171, "seq_len_bases runs past n_seq_words");
    std::vector<uint8_t> o;
    if (format == DBG_SERDE_BINCODE) {
        o.reserve(nw * 8 + n * (8 + 4 + 1 + data_width) + 64);
        put_u64(o, nw);                                             // sequences.sequence.storage
        for (uint64_t i = 0; i < nw; i++) put_u64(o, g->seq_words[i]);
        put_u64(o, g->seq_len_bases);                               // sequences.sequence.len
        put_u64(o, n);                                              // sequences.start
        for (uint64_t i = 0; i < n; i++) put_u64(o, g->start[i]);
        put_u64(o, n);                                              // sequences.length
        for (uint64_t i = 0; i < n; i++) put_le(o, g->length[i], 4);
        put_u64(o, n);                                              // exts: Exts { val: u8 }
        for (uint64_t i = 0; i < n; i++) o.push_back(g->exts[i]);
        put_u64(o, n);                                              // data: D
        if (data_width) for (uint64_t i = 0; i < n; i++) {
            const uint32_t v = g->data ? g->data[i] : 0u;
            if (data_width < 4 && (v >> (8 * data_width))) return sfail(c, 172, "data value does not fit data_width");
            put_le(o, v, (int)data_width);
        }
        o.push_back(g->stranded ? 1 : 0);                           // stranded; phantom: nothing
    } else if (format == DBG_SERDE_JSON) {
        std::string s;
        s.reserve(nw * 21 + n * 48 + 128);
        s += "{\"sequences\":{\"sequence\":{\"storage\":";
        json_u64_array(s, g->seq_words, nw);
        char b[64];
        s.append(b, (size_t)snprintf(b, sizeof(b), ",\"len\":%llu},\"start\":", (unsigned long long)g->seq_len_bases));
        json_u64_array(s, g->start, n);
        s += ",\"length\":[";
        for (uint64_t i = 0; i < n; i++) { if (i) s += ','; s.append(b, (size_t)snprintf(b, sizeof(b), "%u", g->length[i])); }
        s += "]},\"exts\":[";
        for (uint64_t i = 0; i < n; i++) { if (i) s += ','; s.append(b, (size_t)snprintf(b, sizeof(b), "{\"val\":%u}", (unsigned)g->exts[i])); }
        s += "],\"data\":[";
        for (uint64_t i = 0; i < n; i++) {
            if (i) s += ',';
            if (data_width) s.append(b, (size_t)snprintf(b, sizeof(b), "%u", g->data ? g->data[i] : 0u)); else s += "null";     // () serialises as null
        }
        s += g->stranded ? "],\"stranded\":true,\"phantom\":null}" : "],\"stranded\":false,\"phantom\":null}";
        o.assign(s.begin(), s.end());
    } else return sfail(c, 173, "unknown serde format");
    uint8_t* out = (uint8_t*)malloc(o.size() ? o.size() : 1);
    if (!out) return sfail(c, 101, "out of host memory");
    memcpy(out, o.data(), o.size());
    *bytes = out; *len = o.size();
    return 0;
}

static int graph_deserialize_impl(dbg_ctx* c, const uint8_t* bytes, uint64_t len, int format, uint32_t data_width, dbg_graph* out);

extern "C" int dbg_graph_deserialize(dbg_ctx* c, const uint8_t* bytes, uint64_t len, int format, uint32_t data_width, dbg_graph* out) {
    if (!bytes || !out) return sfail(c, 10, "null argument");
    memset(out, 0, sizeof(*out));
    try {                                                           // untrusted input: nothing may unwind through the C ABI
        return graph_deserialize_impl(c, bytes, len, format, data_width, out);
    } catch (const std::exception& e) {
        dbg_free_graph(c, out); memset(out, 0, sizeof(*out));
        return sfail(c, 101, std::string("out of host memory while reading a BaseGraph: ") + e.what());
    }
}

static int graph_deserialize_impl(dbg_ctx* c, const uint8_t* bytes, uint64_t len, int format, uint32_t data_width, dbg_graph* out) {
    std::vector<uint64_t> words, start; std::vector<uint32_t> length, data; std::vector<uint8_t> exts;
    uint64_t n_bases = 0; bool stranded = false;
    if (format == DBG_SERDE_BINCODE) {
        if (!(data_width == 0 || data_width == 1 || data_width == 2 || data_width == 4)) return sfail(c, 170, "data_width must be 0 (unit), 1, 2 or 4");
        Rd r{bytes, len};
        auto vec_len = [&](int elem) -> uint64_t { const uint64_t v = r.le(8); if (!r.bad && elem && v > (r.n - r.i) / (uint64_t)elem) r.bad = true; return r.bad ? 0 : v; };
        uint64_t m = vec_len(8); words.resize(m); for (auto& x : words) x = r.le(8);
        n_bases = r.le(8);
        m = vec_len(8); start.resize(m); for (auto& x : start) x = r.le(8);
        m = vec_len(4); length.resize(m); for (auto& x : length) x = (uint32_t)r.le(4);
        m = vec_len(1); exts.resize(m); for (auto& x : exts) x = (uint8_t)r.le(1);
        m = vec_len((int)data_width); if (!data_width && m != start.size()) r.bad = true;       // Vec<()> has no bytes to bound its length: it must be one unit per node
        if (!r.bad) { data.resize(m); for (auto& x : data) x = data_width ? (uint32_t)r.le((int)data_width) : 0u; }
        stranded = r.le(1) != 0;
        if (r.bad || r.i != len) return sfail(c, 174, "malformed bincode BaseGraph");
    } else if (format == DBG_SERDE_JSON) {
        Js j{(const char*)bytes, len};
        j.eat('{'); j.key("sequences"); j.eat('{'); j.key("sequence"); j.eat('{'); j.key("storage");
        j.array([&] { words.push_back(j.num()); });
        j.eat(','); j.key("len"); n_bases = j.num(); j.eat('}');
        j.eat(','); j.key("start"); j.array([&] { start.push_back(j.num()); });
        j.eat(','); j.key("length"); j.array([&] { length.push_back((uint32_t)j.num_max(UINT32_MAX)); });
        j.eat('}'); j.eat(','); j.key("exts");
        j.array([&] { j.eat('{'); j.key("val"); exts.push_back((uint8_t)j.num_max(255)); j.eat('}'); });
        j.eat(','); j.key("data");
        j.array([&] { if (j.lit("null")) data.push_back(0u); else data.push_back((uint32_t)j.num_max(data_width == 1 ? 255u : data_width == 2 ? 65535u : UINT32_MAX)); });
        j.eat(','); j.key("stranded");
        if (j.lit("true")) stranded = true; else if (!j.lit("false")) j.bad = true;
        j.eat(','); j.key("phantom"); if (!j.lit("null")) j.bad = true;
        j.eat('}'); j.ws();
        if (j.bad || j.i != len) return sfail(c, 175, "malformed serde_json BaseGraph");
    } else return sfail(c, 173, "unknown serde format");
    const uint64_t n = start.size();
    if (length.size() != n || exts.size() != n || data.size() != n || words.size() != (n_bases + 31) / 32) return sfail(c, 176, "inconsistent BaseGraph lengths");
    for (uint64_t i = 0; i < n; i++) if (start[i] > n_bases || length[i] > n_bases - start[i]) return sfail(c, 177, "node runs past the sequence");
    out->n_nodes = n; out->n_seq_words = words.size(); out->seq_len_bases = n_bases; out->stranded = stranded ? 1 : 0;
    out->seq_words = (uint64_t*)malloc((words.size() + 2) * 8);     // two words of tail padding, like every graph this library returns
    out->start = (uint64_t*)malloc(std::max<uint64_t>(n, 1) * 8);
    out->length = (uint32_t*)malloc(std::max<uint64_t>(n, 1) * 4);
    out->exts = (uint8_t*)malloc(std::max<uint64_t>(n, 1));
    out->data = (uint32_t*)malloc(std::max<uint64_t>(n, 1) * 4);
    if (!out->seq_words || !out->start || !out->length || !out->exts || !out->data) { dbg_free_graph(c, out); return sfail(c, 101, "out of host memory"); }
    memcpy(out->seq_words, words.data(), words.size() * 8);
    out->seq_words[words.size()] = out->seq_words[words.size() + 1] = 0;
    if (n) { memcpy(out->start, start.data(), n * 8); memcpy(out->length, length.data(), n * 4); memcpy(out->exts, exts.data(), n); memcpy(out->data, data.data(), n * 4); }
    return 0;
}
