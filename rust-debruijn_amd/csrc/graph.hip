// Second-stage compaction of the sharded flow: BaseGraph::combine (src/graph.rs:71-100),
// BaseGraph::finish (graph.rs:116-142), DebruijnGraph::{find_link, get_valid_exts, fix_exts}
// (graph.rs:252-291, :337-377) and compress_graph / CompressFromGraph (src/compression.rs:100-349).
//
// The reference indexes node ends with two MPHF maps (first k-mer -> node, last k-mer -> node).  Lookups
// are exact, so any exact index gives the same answers: here the device extracts the terminal k-mers of
// every node, radix-sorts them with the node id as payload, and `find_link` is two binary searches.
// fix_exts and the availability-independent part of try_extend_node (compression.rs:115-205) are batched
// device kernels; the greedy walk over nodes (build_node, compression.rs:240-287) and the assembly of the
// path sequences (sequence_of_path, graph.rs:471-491) stay on the host.  Nodes are visited in index order
// (compression.rs:322), so the result does not depend on any MPHF internals.
#include "dbg_internal.hpp"
#include <algorithm>
#include <deque>
#include <cstring>

int compress_links_device(dbg_ctx* c, int k, uint32_t n, const uint64_t* key_hi, const uint64_t* key_lo, const uint8_t* exts,
                          const uint32_t* data, uint32_t* link_dev, const uint32_t* rank_dev, int spec, int stranded,
                          dbg_graph* out, bool* done, const UnitigNodes* nodes, const NodeRec* nrec = nullptr);

namespace {

struct EndIndex {            // sorted terminal k-mers of one side + owning node
    const uint64_t* hi;      // null when k <= 32
    const uint64_t* lo;
    const uint32_t* node;
    uint32_t n;
    // round 5: what the k-mer link builder has had since round 1 (compress.hip).  A plain binary search over 4.4e7 end k-mers is 26
    // dependent steps of two random cache lines each (hi and lo arrays), and fix_exts runs up to 16 searches per node: 60 ms each
    // for fix_exts and the node links at config 3's 4.4e7 shard-graph nodes, 15x the per-element cost of link_kernel.  With a prefix
    // index (pidx[p] = first position whose top pbits key bits are >= p: buckets of ~2 keys) over 16-byte {lo, hi} records a search
    // is one index line + one or two record lines.
    const ulonglong2* rec = nullptr;
    const uint32_t* pidx = nullptr;
    int pbits = 0, key_bits = 0;
};

// (the node words carry two zero words of slack -- dev_graph_build -- so the k-mer fetches need no clamp)
constexpr uint64_t NO_CLAMP = ~0ull;
__device__ __forceinline__ K128 ekey(const EndIndex& t, uint32_t i) { return K128{t.hi ? t.hi[i] : 0ull, t.lo[i]}; }
__device__ __forceinline__ int64_t search_kmer(const EndIndex& t, K128 q) {          // graph.rs:243-249
    uint32_t lo = 0, hi = t.n;
    if (t.pidx) {
        const uint32_t p = (uint32_t)k128_shr(q, t.key_bits - t.pbits).lo;
        lo = t.pidx[p]; hi = t.pidx[p + 1];
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            const ulonglong2 r = t.rec[mid];
            if (k128_lt(K128{r.y, r.x}, q)) lo = mid + 1; else hi = mid;
        }
        if (lo < t.n) { const ulonglong2 r = t.rec[lo]; if (r.y == q.hi && r.x == q.lo) return (int64_t)t.node[lo]; }
        return -1;
    }
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (k128_lt(ekey(t, mid), q)) lo = mid + 1; else hi = mid;
    }
    if (lo < t.n && k128_eq(ekey(t, lo), q)) return (int64_t)t.node[lo];
    return -1;
}
// records + prefix index of a sorted end-k-mer array: position i packs its key and fills the index entries of every prefix in
// (prefix(key[i-1]), prefix(key[i])]; position n closes the index
__global__ void __launch_bounds__(256) end_index_pack_kernel(const uint64_t* __restrict__ hi, const uint64_t* __restrict__ lo, uint32_t n, int key_bits, int pbits,
                                                              ulonglong2* __restrict__ rec, uint32_t* __restrict__ pidx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    const uint32_t np = 1u << pbits;
    uint32_t plo = 0, phi = np;
    if (i < n) {
        const K128 key{hi ? hi[i] : 0ull, lo[i]};
        rec[i] = make_ulonglong2(key.lo, key.hi);
        phi = (uint32_t)k128_shr(key, key_bits - pbits).lo;
    }
    if (i > 0) plo = (uint32_t)k128_shr(K128{hi ? hi[i - 1] : 0ull, lo[i - 1]}, key_bits - pbits).lo + 1u;
    for (uint64_t q = plo; q <= phi; q++) pidx[q] = i;
}

struct Link { int64_t node; int side; int flip; };
// find_link (graph.rs:252-291): dir Left looks for the k-mer at a node's RIGHT end, then (unstranded) for
// its reverse complement at a LEFT end; dir Right the mirror image.
__device__ __forceinline__ Link find_link(const EndIndex& left, const EndIndex& right, K128 kmer, int dir, int stranded, int k) {
    K128 rc = kmer_rc(kmer, k);
    int64_t idx;
    if (dir == 0) {
        if ((idx = search_kmer(right, kmer)) >= 0) return Link{idx, 1, 0};
        if (!stranded && (idx = search_kmer(left, rc)) >= 0) return Link{idx, 0, 1};
    } else {
        if ((idx = search_kmer(left, kmer)) >= 0) return Link{idx, 0, 0};
        if (!stranded && (idx = search_kmer(right, rc)) >= 0) return Link{idx, 1, 1};
    }
    return Link{-1, 0, 0};
}

__global__ void term_kmers_kernel(const uint64_t* __restrict__ words, const uint64_t* __restrict__ start,
                                  const uint32_t* __restrict__ length, uint32_t n, int k,
                                  uint64_t* f_hi, uint64_t* f_lo, uint64_t* l_hi, uint64_t* l_lo, uint32_t* ids_a, uint32_t* ids_b) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    K128 f = packed_get_kmer(words, start[i], k, NO_CLAMP);                                   // first_kmer (lib.rs:369-371)
    K128 l = packed_get_kmer(words, start[i] + length[i] - (uint32_t)k, k, NO_CLAMP);         // last_kmer (lib.rs:374-376)
    if (f_hi) { f_hi[i] = f.hi; l_hi[i] = l.hi; }
    f_lo[i] = f.lo; l_lo[i] = l.lo;
    ids_a[i] = i; ids_b[i] = i;
}

// get_valid_exts (graph.rs:344-377): keep an extension iff it links to a node (that is valid)
__global__ void fix_exts_kernel(EndIndex left, EndIndex right, const uint64_t* __restrict__ words,
                                const uint64_t* __restrict__ start, const uint32_t* __restrict__ length, uint32_t n, int k,
                                int stranded, const uint8_t* __restrict__ valid, const uint8_t* __restrict__ exts_in,
                                uint8_t* __restrict__ exts_out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    K128 lk = packed_get_kmer(words, start[i], k, NO_CLAMP);
    K128 rk = packed_get_kmer(words, start[i] + length[i] - (uint32_t)k, k, NO_CLAMP);
    uint32_t e = exts_in[i], ne = 0;
    for (uint32_t b = 0; b < 4; b++) {
        if (e & (1u << b)) {
            Link L = find_link(left, right, kmer_extend_left(lk, k, b), 0, stranded, k);
            if (L.node >= 0 && (!valid || valid[L.node])) ne |= 1u << b;
        }
        if (e & (16u << b)) {
            Link L = find_link(left, right, kmer_extend_right(rk, k, b), 1, stranded, k);
            if (L.node >= 0 && (!valid || valid[L.node])) ne |= 16u << b;
        }
    }
    exts_out[i] = (uint8_t)ne;
}

// link word per (node, dir): the availability-independent part of try_extend_node (compression.rs:115-205)
constexpr uint32_t NL_TERM = 0xFFFFFFFFu;        // Terminal whatever the availability
constexpr uint32_t NL_NOKMER = 0xFFFFFFFEu;      // reference panics "No kmer" (compression.rs:138)
constexpr uint32_t NL_INCONSISTENT = 0xFFFFFFFDu;// assert!(consistent) fails (compression.rs:165)
constexpr uint32_t NL_PANIC_BIT = 1u;            // incoming_count == 0 -> panic "unreachable" if the node is available (:190-195)
// otherwise (next_node << 2) | (next_side_outgoing << 1) | panic_bit

__device__ __forceinline__ bool join_ok(int spec, uint32_t a, uint32_t b) { return spec == DBG_SPEC_SCMAP_EQ ? a == b : true; }

__global__ void node_link_kernel(EndIndex left, EndIndex right, const uint64_t* __restrict__ words,
                                 const uint64_t* __restrict__ start, const uint32_t* __restrict__ length, uint32_t n, int k,
                                 int stranded /* the walk's */, int g_stranded /* the graph's: drives find_link */, int spec,
                                 const uint8_t* __restrict__ exts, const uint32_t* __restrict__ data,
                                 uint32_t* __restrict__ link /* [2][n] */, int partial = 0 /* a k-mer that is in no node lives in another shard: the path ends here for now */) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t e = exts[i];
    K128 fk = packed_get_kmer(words, start[i], k, NO_CLAMP);
    K128 lk = packed_get_kmer(words, start[i] + length[i] - (uint32_t)k, k, NO_CLAMP);
    const bool self_pal = !stranded && length[i] == (uint32_t)k && (k % 2 == 0) && k128_eq(fk, kmer_rc(fk, k));   // :121
    for (int dir = 0; dir < 2; dir++) {
        uint32_t out = NL_TERM;
        uint32_t bits = (e >> (4 * dir)) & 0xfu;
        if (__popc(bits) == 1 && !self_pal) {                                          // :120-123
            uint32_t base = 31 - __clz(bits);
            K128 end_kmer = dir == 0 ? fk : lk;                                        // term_kmer(dir) :127
            K128 next_kmer = dir == 0 ? kmer_extend_left(end_kmer, k, base) : kmer_extend_right(end_kmer, k, base);
            Link L = find_link(left, right, next_kmer, dir, g_stranded, k);           // :130 (self.graph.find_link)
            if (L.node < 0) out = partial ? NL_TERM : NL_NOKMER;
            else {
                const uint32_t nn = (uint32_t)L.node;
                bool consistent = length[nn] == (uint32_t)k ||                         // :145-165
                                  (dir == 0 && L.side == 1 && !L.flip) || (dir == 0 && L.side == 0 && L.flip) ||
                                  (dir == 1 && L.side == 0 && !L.flip) || (dir == 1 && L.side == 1 && L.flip);
                bool next_pal = !stranded && (k % 2 == 0) && k128_eq(next_kmer, kmer_rc(next_kmer, k));          // :174
                if (!consistent) out = NL_INCONSISTENT;
                else if (next_pal || (spec == DBG_SPEC_SCMAP_EQ && data && !join_ok(spec, data[i], data[nn]))) out = NL_TERM;   // :173-182 (only ScmapCompress reads the data)
                else {
                    uint32_t incoming = __popc((exts[nn] >> (4 * L.side)) & 0xfu);     // num_ext_dir(next_side_incoming) :187
                    int outgoing = 1 - L.side;                                         // next_side_incoming.flip() :185
                    if (incoming == 0) out = (nn << 2) | ((uint32_t)outgoing << 1) | NL_PANIC_BIT;
                    else if (incoming == 1) out = (nn << 2) | ((uint32_t)outgoing << 1);
                    else out = NL_TERM;                                                // :199-203
                }
            }
        }
        link[(uint64_t)dir * n + i] = out;
    }
}

// find_edges (graph.rs:223-241) for both sides of every node: slot b (left) / 4 + b (right) of node i holds the
// link reached through extension base b, or EDGE_NONE when the ext bit is clear or no node end matches
// ("this edge doesn't exist within this shard").  info: bit0 = incoming side of the target is Right,
// bit1 = the target is flipped.
constexpr uint32_t EDGE_NONE = 0xFFFFFFFFu;
__global__ void edges_kernel(EndIndex left, EndIndex right, const uint64_t* __restrict__ words,
                             const uint64_t* __restrict__ start, const uint32_t* __restrict__ length, uint32_t n, int k,
                             int stranded, const uint8_t* __restrict__ exts, uint32_t* __restrict__ target, uint8_t* __restrict__ info) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    K128 lk = packed_get_kmer(words, start[i], k, NO_CLAMP);                                  // term_kmer(Dir::Left)
    K128 rk = packed_get_kmer(words, start[i] + length[i] - (uint32_t)k, k, NO_CLAMP);        // term_kmer(Dir::Right)
    const uint32_t e = exts[i];
    for (uint32_t b = 0; b < 4; b++) {
        uint32_t t = EDGE_NONE, f = 0;
        if (e & (1u << b)) {
            Link L = find_link(left, right, kmer_extend_left(lk, k, b), 0, stranded, k);
            if (L.node >= 0) { t = (uint32_t)L.node; f = (uint32_t)L.side | ((uint32_t)L.flip << 1); }
            // no link: "this edge doesn't exist within this shard, so ignore it" (graph.rs:231-236; the expect is commented out)
        }
        target[(uint64_t)i * 8 + b] = t; info[(uint64_t)i * 8 + b] = (uint8_t)f;
        t = EDGE_NONE; f = 0;
        if (e & (16u << b)) {
            Link L = find_link(left, right, kmer_extend_right(rk, k, b), 1, stranded, k);
            if (L.node >= 0) { t = (uint32_t)L.node; f = (uint32_t)L.side | ((uint32_t)L.flip << 1); }
        }
        target[(uint64_t)i * 8 + 4 + b] = t; info[(uint64_t)i * 8 + 4 + b] = (uint8_t)f;
    }
}

// node links -> the element-link format of unitig.hip ((next << 1) | side we leave `next` through; PANIC flag; TERM) + weights
__global__ void node_links_to_unitig_kernel(const uint32_t* __restrict__ nl, const uint32_t* __restrict__ length, uint32_t n, int k,
                                            uint32_t* __restrict__ ul, uint32_t* __restrict__ weight) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= 2 * n) return;
    const uint32_t L = nl[s];
    uint32_t o;
    if (L == NL_TERM) o = 0xFFFFFFFFu;
    else if (L == NL_NOKMER || L == NL_INCONSISTENT) o = 0x80000000u | ((s % n) << 1);   // the reference panics (target = self: never censored away): the literal walk reports it
    else o = ((L >> 2) << 1) | ((L >> 1) & 1u) | ((L & NL_PANIC_BIT) ? 0x80000000u : 0u);
    ul[s] = o;
    if (s < n) weight[s] = length[s] - (uint32_t)k + 1u;
}

// ---- host helpers --------------------------------------------------------------------------
struct HostBits {           // growing packed base stream (DnaString::push, dna_string.rs:303-310)
    std::vector<uint64_t> words;
    uint64_t len = 0;
    inline void push(uint32_t b) {
        if ((len & 31) == 0) words.push_back(0);
        words.back() |= (uint64_t)(b & 3u) << (62 - 2 * (len & 31));
        len++;
    }
};
inline uint32_t hget(const uint64_t* w, uint64_t o) { return (uint32_t)((w[o >> 5] >> (62 - 2 * (o & 31))) & 3ull); }

inline uint32_t spec_reduce_h(int spec, uint32_t a, uint32_t b, bool* panic) {
    switch (spec) {
        case DBG_SPEC_SIMPLE_SAT_ADD_U16: { uint32_t s = a + b; return s > 65535u ? 65535u : s; }
        case DBG_SPEC_SIMPLE_ADD_MOD_U16: return (a + b) % 65535u;
        case DBG_SPEC_SIMPLE_MAX_U16: return std::max(a, b);
        case DBG_SPEC_SCMAP_EQ: if (a != b) *panic = true; return a;
        case DBG_SPEC_SIMPLE_WRAP_ADD_U16: return (a + b) & 0xFFFFu;
    }
    return a;
}

void* dup_bytes(const void* p, size_t bytes) { void* q = malloc(bytes ? bytes : 1); if (bytes) memcpy(q, p, bytes); return q; }

void graph_out(const HostBits& seq, const std::vector<uint64_t>& st, const std::vector<uint32_t>& ln,
               const std::vector<uint8_t>& ex, const std::vector<uint32_t>& da, int stranded, dbg_graph* out) {
    out->n_nodes = st.size(); out->n_seq_words = seq.words.size(); out->seq_len_bases = seq.len;
    out->seq_words = (uint64_t*)dup_bytes(seq.words.data(), seq.words.size() * 8);
    out->start = (uint64_t*)dup_bytes(st.data(), st.size() * 8);
    out->length = (uint32_t*)dup_bytes(ln.data(), ln.size() * 4);
    out->exts = (uint8_t*)dup_bytes(ex.data(), ex.size());
    out->data = (uint32_t*)dup_bytes(da.data(), da.size() * 4);
    out->stranded = stranded;
}

// Device-resident DebruijnGraph index (finish, graph.rs:116-142) + fix_exts / links on it.
struct DevGraph {
    DBuf<uint64_t> words, start, f_hi, f_lo, l_hi, l_lo, t_hi, t_lo;
    DBuf<uint32_t> length, f_id, l_id, t_id, data, f_pidx, l_pidx;
    DBuf<ulonglong2> f_rec, l_rec;
    DBuf<uint8_t> exts;
    EndIndex left{nullptr, nullptr, nullptr, 0}, right{nullptr, nullptr, nullptr, 0};
    uint32_t n = 0;
};

int dev_graph_index(dbg_ctx* c, int k, DevGraph* d);
int dev_graph_build(dbg_ctx* c, int k, const dbg_graph* g, DevGraph* d) {
    const uint32_t n = (uint32_t)g->n_nodes;
    d->n = n;
    const size_t na = std::max<uint32_t>(n, 1), nw = std::max<uint64_t>(g->n_seq_words, 1);
    ALLOC_OR_FAIL(c, d->words, nw + 2); ALLOC_OR_FAIL(c, d->start, na); ALLOC_OR_FAIL(c, d->length, na);
    ALLOC_OR_FAIL(c, d->exts, na); ALLOC_OR_FAIL(c, d->data, na);
    HIP_TRY(c, hipMemsetAsync(d->words.p, 0, (nw + 2) * 8, c->stream));
    if (g->n_seq_words) HIP_TRY(c, hipMemcpyAsync(d->words.p, g->seq_words, g->n_seq_words * 8, hipMemcpyHostToDevice, c->stream));
    if (n) {
        HIP_TRY(c, hipMemcpyAsync(d->start.p, g->start, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(d->length.p, g->length, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(d->exts.p, g->exts, (size_t)n, hipMemcpyHostToDevice, c->stream));
        if (g->data) HIP_TRY(c, hipMemcpyAsync(d->data.p, g->data, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
        else HIP_TRY(c, hipMemsetAsync(d->data.p, 0, (size_t)n * 4, c->stream));
    }
    return dev_graph_index(c, k, d);
}

// finish (graph.rs:116-142) over node arrays that are already in HBM: sorted (first k-mer, node) and (last k-mer, node)
int dev_graph_index(dbg_ctx* c, int k, DevGraph* d) {
    const uint32_t n = d->n;
    const bool has_hi = k > 32;
    const size_t na = std::max<uint32_t>(n, 1);
    if (has_hi) { ALLOC_OR_FAIL(c, d->f_hi, na); ALLOC_OR_FAIL(c, d->l_hi, na); ALLOC_OR_FAIL(c, d->t_hi, na); }
    ALLOC_OR_FAIL(c, d->f_lo, na); ALLOC_OR_FAIL(c, d->l_lo, na); ALLOC_OR_FAIL(c, d->t_lo, na);
    ALLOC_OR_FAIL(c, d->f_id, na); ALLOC_OR_FAIL(c, d->l_id, na); ALLOC_OR_FAIL(c, d->t_id, na);
    if (n) {
        term_kmers_kernel<<<cdiv(n, 256), 256, 0, c->stream>>>(d->words.p, d->start.p, d->length.p, n, k, d->f_hi.p, d->f_lo.p,
                                                                d->l_hi.p, d->l_lo.p, d->f_id.p, d->l_id.p);
        LAUNCH_CHECK(c, "term_kmers");
    }
    // sort (first k-mer, node) and (last k-mer, node); a copy of the unsorted keys is not needed afterwards
    bool in_b = false;
    RecArrays F{has_hi ? d->f_hi.p : nullptr, d->f_lo.p, d->f_id.p}, T{has_hi ? d->t_hi.p : nullptr, d->t_lo.p, d->t_id.p};
    DBG_TRY(radix_sort_records(c, n, F, T, 2 * k, 0, 0, &in_b));
    if (in_b) { std::swap(d->f_hi, d->t_hi); std::swap(d->f_lo, d->t_lo); std::swap(d->f_id, d->t_id); }
    RecArrays L{has_hi ? d->l_hi.p : nullptr, d->l_lo.p, d->l_id.p}, T2{has_hi ? d->t_hi.p : nullptr, d->t_lo.p, d->t_id.p};
    DBG_TRY(radix_sort_records(c, n, L, T2, 2 * k, 0, 0, &in_b));
    if (in_b) { std::swap(d->l_hi, d->t_hi); std::swap(d->l_lo, d->t_lo); std::swap(d->l_id, d->t_id); }
    d->left = EndIndex{has_hi ? d->f_hi.p : nullptr, d->f_lo.p, d->f_id.p, n};
    d->right = EndIndex{has_hi ? d->l_hi.p : nullptr, d->l_lo.p, d->l_id.p, n};
    // prefix index + packed records for graphs that are worth it (DBG_PIDX_BITS=0: plain binary searches, for A/B runs); if the
    // memory is not to be had the plain search serves
    const char* knob = c->opt("DBG_PIDX_BITS");
    if (n >= 64 && !(knob && !strcmp(knob, "0"))) {
        int pb = 8;
        while (pb < 27 && pb < 2 * k && (n >> pb) > 2) pb++;
        if (d->f_rec.alloc(c, n) && d->l_rec.alloc(c, n) && d->f_pidx.alloc(c, ((size_t)1 << pb) + 1) && d->l_pidx.alloc(c, ((size_t)1 << pb) + 1)) {
            c->t_begin("graph_end_index", n);
            end_index_pack_kernel<<<cdiv((uint64_t)n + 1, 256), 256, 0, c->stream>>>(d->left.hi, d->left.lo, n, 2 * k, pb, d->f_rec.p, d->f_pidx.p);
            end_index_pack_kernel<<<cdiv((uint64_t)n + 1, 256), 256, 0, c->stream>>>(d->right.hi, d->right.lo, n, 2 * k, pb, d->l_rec.p, d->l_pidx.p);
            c->t_end();
            LAUNCH_CHECK(c, "end_index_pack");
            d->left.rec = d->f_rec.p; d->left.pidx = d->f_pidx.p; d->left.pbits = pb; d->left.key_bits = 2 * k;
            d->right.rec = d->l_rec.p; d->right.pidx = d->l_pidx.p; d->right.pbits = pb; d->right.key_bits = 2 * k;
        } else { d->f_rec.release(); d->l_rec.release(); d->f_pidx.release(); d->l_pidx.release(); }
    }
    return 0;
}

// fix_exts (graph.rs:337-342) in place on the device exts; valid may be null
int dev_fix_exts(dbg_ctx* c, int k, int stranded, DevGraph* d, const uint8_t* valid_dev) {
    if (!d->n) return 0;
    DBuf<uint8_t> out;
    ALLOC_OR_FAIL(c, out, d->n);
    c->t_begin("graph_fix_exts", d->n);
    fix_exts_kernel<<<cdiv(d->n, 256), 256, 0, c->stream>>>(d->left, d->right, d->words.p, d->start.p, d->length.p, d->n, k, stranded,
                                                             valid_dev, d->exts.p, out.p);
    c->t_end();
    LAUNCH_CHECK(c, "fix_exts");
    std::swap(d->exts, out);
    return 0;
}
}  // namespace

// BaseGraph::combine (graph.rs:71-100): sequences re-added back-to-back, exts/data concatenated
extern "C" int dbg_graph_combine(dbg_ctx* c, const dbg_graph* graphs, uint32_t n_graphs, dbg_graph* out) {
    memset(out, 0, sizeof(*out));
    bool all_s = true, none_s = true;
    HostBits seq;
    std::vector<uint64_t> st; std::vector<uint32_t> ln, da; std::vector<uint8_t> ex;
    for (uint32_t gi = 0; gi < n_graphs; gi++) {
        const dbg_graph& g = graphs[gi];
        for (uint64_t i = 0; i < g.n_nodes; i++) {
            st.push_back(seq.len);
            for (uint32_t p = 0; p < g.length[i]; p++) seq.push(hget(g.seq_words, g.start[i] + p));
            ln.push_back(g.length[i]);
            ex.push_back(g.exts[i]);
            da.push_back(g.data ? g.data[i] : 0u);
        }
        all_s = all_s && g.stranded; none_s = none_s && !g.stranded;
    }
    if (!all_s && !none_s) return c->fail(50, "attempted to combine stranded and unstranded graphs (graph.rs:90)");
    graph_out(seq, st, ln, ex, da, all_s ? 1 : 0, out);
    return 0;
}

// compress_graph (compression.rs:338-349): old_graph is an (unfinished or finished) BaseGraph on the host
extern "C" int dbg_compress_graph(dbg_ctx* c, uint32_t k_, int stranded, int spec, const dbg_graph* old_graph,
                                  const uint64_t* censor_nodes, uint64_t n_censor, dbg_graph* out) {
    const int k = (int)k_;
    if (k < 1 || k > 64) return c->fail(40, "k must be in 1..=64");
    if (spec < 0 || spec > 4) return c->fail(41, "unknown CompressionSpec");
    if (old_graph->n_nodes >= (1ull << 30)) return c->fail(51, "compress_graph: at most 2^30-1 nodes per call in this build");
    HIP_TRY(c, hipSetDevice(c->device));
    c->t_clear();
    memset(out, 0, sizeof(*out));
    const uint32_t n = (uint32_t)old_graph->n_nodes;
    for (uint32_t i = 0; i < n; i++) if (old_graph->length[i] < (uint32_t)k) return c->fail(52, "node shorter than k");
    // old_graph.finish() is implied by the DebruijnGraph argument of the reference; the graph's own
    // strandedness drives find_link (graph.rs:270,282), the `stranded` argument drives the walk
    const int g_stranded = old_graph->stranded;
    std::vector<uint8_t> available(n, 1);                                              // :297-307
    for (uint64_t i = 0; i < n_censor; i++) { if (censor_nodes[i] >= n) return c->fail(53, "censor node out of range"); available[censor_nodes[i]] = 0; }

    DevGraph d;
    DBG_TRY(dev_graph_build(c, k, old_graph, &d));
    DBuf<uint8_t> d_avail;
    ALLOC_OR_FAIL(c, d_avail, std::max<uint32_t>(n, 1));
    if (n) HIP_TRY(c, hipMemcpyAsync(d_avail.p, available.data(), n, hipMemcpyHostToDevice, c->stream));
    DBG_TRY(dev_fix_exts(c, k, g_stranded, &d, d_avail.p));                            // old_graph.fix_exts(Some(&available)) :309
    std::vector<uint32_t> link(2 * (size_t)n);
    std::vector<uint8_t> exts(n);
    if (n) {
        DBuf<uint32_t> d_link;
        ALLOC_OR_FAIL(c, d_link, 2 * (size_t)n);
        c->t_begin("graph_node_links", n);
        node_link_kernel<<<cdiv(n, 256), 256, 0, c->stream>>>(d.left, d.right, d.words.p, d.start.p, d.length.p, n, k, stranded, g_stranded, spec,
                                                               d.exts.p, d.data.p, d_link.p);
        c->t_end();
        LAUNCH_CHECK(c, "node_links");
        // the whole walk on the device (same chain construction as compress_kmers_with_hash, elements = nodes; censored nodes
        // are neither entered nor emitted), unless the links are not mutual / carry a panic marker: the literal host walk
        // below then reproduces the reference's behaviour (including its panics)
        const char* mode = c->opt("DBG_COMPRESS");
        if (!(mode && !strcmp(mode, "host"))) {
            DBuf<uint32_t> u_link, u_weight;
            ALLOC_OR_FAIL(c, u_link, 2 * (size_t)n);
            ALLOC_OR_FAIL(c, u_weight, n);
            node_links_to_unitig_kernel<<<cdiv(2 * (uint64_t)n, 256), 256, 0, c->stream>>>(d_link.p, d.length.p, n, k, u_link.p, u_weight.p);
            LAUNCH_CHECK(c, "node_links_to_unitig");
            UnitigNodes un{u_weight.p, d.words.p, d.start.p, d.length.p, n_censor ? d_avail.p : nullptr};
            dbg_graph ng;
            bool done = false;
            DBG_TRY(compress_links_device(c, k, n, nullptr, nullptr, d.exts.p, d.data.p, u_link.p, nullptr, spec, stranded, &ng, &done, &un));
            if (done) {
                // ---- graph.finish(); dbg.fix_exts(None) (compression.rs:330-331) ----
                DevGraph d2;
                int r = dev_graph_build(c, k, &ng, &d2);
                if (!r) r = dev_fix_exts(c, k, stranded, &d2, nullptr);
                if (!r && ng.n_nodes) {
                    if (hipMemcpyAsync(ng.exts, d2.exts.p, ng.n_nodes, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                        hipStreamSynchronize(c->stream) != hipSuccess) r = c->fail(100, "copy of fixed exts failed");
                }
                if (r) { dbg_free_graph(c, &ng); return r; }
                *out = ng;
                return 0;
            }
            if (mode && !strcmp(mode, "device")) return c->fail(48, "DBG_COMPRESS=device but the node links are not mutual");
        }
        HIP_TRY(c, hipMemcpyAsync(link.data(), d_link.p, 2 * (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(exts.data(), d.exts.p, n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    // ---- host: greedy walk in node order (compression.rs:322-327) ----
    const uint64_t* W = old_graph->seq_words;
    HostBits seq;
    std::vector<uint64_t> g_start; std::vector<uint32_t> g_len, g_data; std::vector<uint8_t> g_exts;
    std::vector<std::pair<uint32_t, int>> l_path, r_path;
    bool spec_panic = false;
    auto single_dir = [&](uint32_t id, int dir) -> uint32_t { return dir ? (exts[id] >> 4) : (exts[id] & 0xfu); };
    // extend_node (compression.rs:208-235): path of (node, incoming dir) + terminal exts nibble
    auto extend = [&](uint32_t start_node, int start_dir, std::vector<std::pair<uint32_t, int>>& path, int* err) -> uint32_t {
        path.clear();
        uint32_t cur = start_node; int dir = start_dir;
        available[start_node] = 0;                                                     // :214
        for (;;) {
            uint32_t L = link[(uint64_t)dir * n + cur];
            if (L == NL_TERM) return single_dir(cur, dir);
            if (L == NL_NOKMER) { *err = 1; return 0; }
            if (L == NL_INCONSISTENT) { *err = 2; return 0; }
            uint32_t nn = L >> 2; int outgoing = (int)((L >> 1) & 1u);
            if (!available[nn]) return single_dir(cur, dir);                           // :173-182
            if (L & NL_PANIC_BIT) { *err = 3; return 0; }                              // :190-195
            path.push_back({nn, 1 - outgoing});                                        // (next_node, next_dir_incoming) :221-222
            available[nn] = 0;
            cur = nn; dir = outgoing;
        }
    };
    for (uint32_t nc = 0; nc < n; nc++) {
        if (!available[nc]) continue;                                                  // :323
        int err = 0;
        uint32_t l_ext = extend(nc, 0, l_path, &err);                                  // :241
        uint32_t r_ext = err ? 0 : extend(nc, 1, r_path, &err);                        // :242
        if (err == 1) return c->fail(55, "No kmer (compression.rs:138)");
        if (err == 2) return c->fail(56, "assertion failed: consistent (compression.rs:165)");
        if (err == 3) return c->fail(57, "unreachable (compression.rs:195)");
        std::deque<std::pair<uint32_t, int>> node_path;                                // (node, Dir) with Left = as stored
        uint32_t node_data = old_graph->data ? old_graph->data[nc] : 0u;               // :247
        node_path.push_back({nc, 0});
        for (auto& pr : l_path) {                                                      // :251-256
            node_path.push_front({pr.first, 1 - pr.second});
            node_data = spec_reduce_h(spec, node_data, old_graph->data ? old_graph->data[pr.first] : 0u, &spec_panic);
        }
        for (auto& pr : r_path) {                                                      // :259-264
            node_path.push_back({pr.first, pr.second});
            node_data = spec_reduce_h(spec, node_data, old_graph->data ? old_graph->data[pr.first] : 0u, &spec_panic);
        }
        if (spec_panic) return c->fail(46, "ScmapCompress::reduce on unequal data: Should not happen (compression.rs:90)");
        uint32_t left_extend = (!l_path.empty() && l_path.back().second == 0) ? (exts_complement(l_ext) & 0xfu) : l_ext;   // :266-270
        uint32_t right_extend = (!r_path.empty() && r_path.back().second == 1) ? (exts_complement(r_ext) & 0xfu) : r_ext;  // :272-276
        // sequence_of_path (graph.rs:471-491)
        g_start.push_back(seq.len);
        uint32_t total = 0, idx = 0;
        for (auto& pr : node_path) {
            const uint64_t s0 = old_graph->start[pr.first];
            const uint32_t len = old_graph->length[pr.first];
            for (uint32_t p = idx == 0 ? 0 : (uint32_t)k - 1; p < len; p++) {
                uint32_t b = pr.second == 0 ? hget(W, s0 + p) : 3u - hget(W, s0 + len - 1 - p);   // DnaStringSlice::rc get (dna_string.rs:572-578)
                seq.push(b);
                total++;
            }
            idx++;
        }
        g_len.push_back(total);
        g_exts.push_back((uint8_t)(((right_extend & 0xfu) << 4) | (left_extend & 0xfu)));
        g_data.push_back(node_data);
    }
    // ---- graph.finish(); dbg.fix_exts(None) (compression.rs:330-331) ----
    dbg_graph ng;
    memset(&ng, 0, sizeof(ng));
    graph_out(seq, g_start, g_len, g_exts, g_data, stranded, &ng);
    {
        DevGraph d2;
        int r = dev_graph_build(c, k, &ng, &d2);
        if (!r) r = dev_fix_exts(c, k, stranded, &d2, nullptr);
        if (!r && ng.n_nodes) {
            if (hipMemcpyAsync(ng.exts, d2.exts.p, ng.n_nodes, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                hipStreamSynchronize(c->stream) != hipSuccess) r = c->fail(100, "copy of fixed exts failed");
        }
        if (r) { dbg_free_graph(c, &ng); return r; }
    }
    *out = ng;
    return 0;
}

extern "C" void dbg_free_edges(dbg_edges* e);
static int graph_edges_impl(dbg_ctx* c, int k, const dbg_graph* g, dbg_edges* out) {
    const uint32_t n = (uint32_t)g->n_nodes;
    DevGraph d;
    DBG_TRY(dev_graph_build(c, k, g, &d));
    DBuf<uint32_t> d_t;
    DBuf<uint8_t> d_i;
    ALLOC_OR_FAIL(c, d_t, (size_t)n * 8);
    ALLOC_OR_FAIL(c, d_i, (size_t)n * 8);
    c->t_begin("graph_edges", n);
    edges_kernel<<<cdiv(n, 256), 256, 0, c->stream>>>(d.left, d.right, d.words.p, d.start.p, d.length.p, n, k, g->stranded, d.exts.p, d_t.p, d_i.p);
    c->t_end();
    LAUNCH_CHECK(c, "graph_edges");
    HIP_TRY(c, hipMemcpyAsync(out->target, d_t.p, (size_t)n * 8 * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out->info, d_i.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

// Node::l_edges / r_edges (graph.rs:1041-1049) of every node of a (finished) graph, computed on the device.
extern "C" int dbg_graph_edges(dbg_ctx* c, uint32_t k_, const dbg_graph* g, dbg_edges* out) {
    const int k = (int)k_;
    memset(out, 0, sizeof(*out));
    if (k < 1 || k > 64) return c->fail(40, "k must be in 1..=64");
    if (g->n_nodes >= (1ull << 30)) return c->fail(51, "graph_edges: at most 2^30-1 nodes per call in this build");
    HIP_TRY(c, hipSetDevice(c->device));
    const uint32_t n = (uint32_t)g->n_nodes;
    for (uint32_t i = 0; i < n; i++) {
        if (g->length[i] < (uint32_t)k) return c->fail(52, "node shorter than k");
        if (g->start[i] + g->length[i] > g->seq_len_bases) return c->fail(54, "node runs past seq_len_bases");
    }
    out->n_nodes = n;
    out->target = (uint32_t*)malloc(std::max<size_t>((size_t)n * 8 * 4, 1));
    out->info = (uint8_t*)malloc(std::max<size_t>((size_t)n * 8, 1));
    if (!out->target || !out->info) { dbg_free_edges(out); return c->fail(101, "out of host memory"); }
    if (!n) return 0;
    const int r = graph_edges_impl(c, k, g, out);
    if (r) dbg_free_edges(out);                                  // nothing is handed out on failure
    return r;
}
extern "C" void dbg_free_edges(dbg_edges* e) {
    if (!e) return;
    free(e->target); free(e->info);
    memset(e, 0, sizeof(*e));
}

// DebruijnGraph::write_gfa (graph.rs:537-611): header, then per node its S line, the L lines of its left
// edges with target >= node (":563"), then of its right edges with target > node (":579").  Node sequences
// are unpacked to ASCII and edges resolved on the device; the host only formats lines.
extern "C" int dbg_graph_to_gfa(dbg_ctx* c, uint32_t k_, const dbg_graph* g, char** text, uint64_t* len) {
    *text = nullptr; *len = 0;
    dbg_edges E;
    DBG_TRY(dbg_graph_edges(c, k_, g, &E));
    const uint32_t n = (uint32_t)g->n_nodes;
    std::vector<uint8_t> ascii(std::max<uint64_t>(g->seq_len_bases, 1));
    if (g->seq_len_bases) {
        int r = dbg_unpack_acgt(c, g->seq_words, 0, g->seq_len_bases, ascii.data());
        if (r) { dbg_free_edges(&E); return r; }
    }
    std::string s;
    s.reserve((size_t)g->seq_len_bases + (size_t)n * 48 + 64);
    s += "H\tVN:Z:debruijn-rs\n";
    char buf[96];
    const unsigned ov = k_ - 1;
    for (uint32_t i = 0; i < n; i++) {
        int m = snprintf(buf, sizeof(buf), "S\t%u\t", i);
        s.append(buf, m);
        s.append((const char*)ascii.data() + g->start[i], g->length[i]);
        s += '\n';
        for (int side = 0; side < 2; side++) {
            for (int b = 0; b < 4; b++) {
                const uint32_t t = E.target[(size_t)i * 8 + side * 4 + b];
                if (t == EDGE_NONE) continue;
                if (side == 0 ? t < i : t <= i) continue;
                const char to_dir = (E.info[(size_t)i * 8 + side * 4 + b] & 1) ? '-' : '+';     // Dir::Left => "+", Dir::Right => "-"
                m = snprintf(buf, sizeof(buf), "L\t%u\t%c\t%u\t%c\t%uM\n", i, side == 0 ? '-' : '+', t, to_dir, ov);
                s.append(buf, m);
            }
        }
    }
    dbg_free_edges(&E);
    char* o = (char*)malloc(s.size() + 1);
    if (!o) return c->fail(101, "out of host memory");
    memcpy(o, s.data(), s.size());
    o[s.size()] = 0;
    *text = o; *len = s.size();
    return 0;
}
extern "C" void dbg_free_text(char* t) { free(t); }

// DebruijnGraph::to_gfa (graph.rs:598-601)
extern "C" int dbg_graph_write_gfa(dbg_ctx* c, uint32_t k, const dbg_graph* g, const char* path) {
    char* t = nullptr; uint64_t len = 0;
    DBG_TRY(dbg_graph_to_gfa(c, k, g, &t, &len));
    FILE* f = fopen(path, "wb");
    if (!f) { free(t); return c->fail(102, std::string("cannot create ") + path); }
    size_t w = fwrite(t, 1, len, f);
    int r = fclose(f);
    free(t);
    if (w != len || r) return c->fail(103, std::string("short write to ") + path);
    return 0;
}


// ================================================================================================
// The rank-spanning end of the flow with the graphs resident in HBM (round 4): dbg_shard_compress_dev.
// Reference flow: per-shard compress_kmers_with_hash -> BaseGraph::combine (graph.rs:71-100) -> compress_graph
// (compression.rs:291-349), src/test.rs:459-470.  Rounds 2-3 ran it in Python: every rank's graph device -> host numpy ->
// device -> p2p -> host, and one rank combined on the host.  Here a shard graph never leaves HBM between the steps: the unitig
// construction hands its buffers over (dbg_ctx::graph_sink), combine is a packing kernel, compress_graph runs its device
// route on device-resident nodes, and graphs travel as device buffers over the dbg_transport.
// ================================================================================================
extern "C" int dbg_compress_table_dev(dbg_ctx* c, uint32_t k, int stranded, int spec, const dbg_kmer_table* t, dbg_graph* out,
                                      dbg_label_classes* classes);
#include "shard_comm.hpp"
namespace {


int graph_dev_from_host(dbg_ctx* c, const dbg_graph* g, GraphDev* d) {
    const uint64_t n = g->n_nodes, nw = g->n_seq_words;
    ALLOC_OR_FAIL(c, d->words, nw + 3); ALLOC_OR_FAIL(c, d->start, n + 1); ALLOC_OR_FAIL(c, d->length, std::max<uint64_t>(n, 1));
    ALLOC_OR_FAIL(c, d->exts, std::max<uint64_t>(n, 1)); ALLOC_OR_FAIL(c, d->data, std::max<uint64_t>(n, 1));
    HIP_TRY(c, hipMemsetAsync(d->words.p, 0, (nw + 3) * 8, c->stream));
    if (nw) HIP_TRY(c, hipMemcpyAsync(d->words.p, g->seq_words, nw * 8, hipMemcpyHostToDevice, c->stream));
    if (n) {
        HIP_TRY(c, hipMemcpyAsync(d->start.p, g->start, n * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(d->length.p, g->length, n * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(d->exts.p, g->exts, n, hipMemcpyHostToDevice, c->stream));
        if (g->data) HIP_TRY(c, hipMemcpyAsync(d->data.p, g->data, n * 4, hipMemcpyHostToDevice, c->stream));
        else HIP_TRY(c, hipMemsetAsync(d->data.p, 0, n * 4, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    d->n_nodes = n; d->n_words = nw; d->n_bases = g->seq_len_bases; d->stranded = g->stranded; d->filled = true;
    return 0;
}

int graph_dev_to_host(dbg_ctx* c, const GraphDev& d, dbg_graph* g) {
    memset(g, 0, sizeof(*g));
    const uint64_t n = d.n_nodes, nw = d.n_words;
    g->n_nodes = n; g->n_seq_words = nw; g->seq_len_bases = d.n_bases; g->stranded = d.stranded;
    g->seq_words = (uint64_t*)ctx_halloc(c, std::max<uint64_t>(nw, 1) * 8);
    g->start = (uint64_t*)ctx_halloc(c, std::max<uint64_t>(n, 1) * 8);
    g->length = (uint32_t*)ctx_halloc(c, std::max<uint64_t>(n, 1) * 4);
    g->exts = (uint8_t*)ctx_halloc(c, std::max<uint64_t>(n, 1));
    g->data = (uint32_t*)ctx_halloc(c, std::max<uint64_t>(n, 1) * 4);
    if (nw) HIP_TRY(c, hipMemcpyAsync(g->seq_words, d.words.p, nw * 8, hipMemcpyDeviceToHost, c->stream));
    if (n) {
        HIP_TRY(c, hipMemcpyAsync(g->start, d.start.p, n * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(g->length, d.length.p, n * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(g->exts, d.exts.p, n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(g->data, d.data.p, n * 4, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

void graph_dev_clear(GraphDev* d) {
    d->words.release(); d->start.release(); d->length.release(); d->data.release(); d->exts.release();
    d->n_nodes = d->n_words = d->n_bases = 0; d->filled = false;
}

// ---- BaseGraph::combine on the device (graph.rs:71-100): node sequences re-added back to back -----------------------------
__global__ void __launch_bounds__(256) add_offset_kernel(const uint64_t* __restrict__ in, uint64_t n, uint64_t add, uint64_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + add;
}
// one output word per thread: the (up to 32) bases it holds come from the nodes that cover base positions [32 w, 32 w + 32)
__global__ void __launch_bounds__(256) combine_pack_kernel(const uint64_t* __restrict__ src_words, const uint64_t* __restrict__ src_start,
                                                            const uint32_t* __restrict__ length, const uint64_t* __restrict__ out_start,
                                                            uint64_t n_nodes, uint64_t n_words_out, uint64_t total_bases, uint64_t* __restrict__ out) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words_out) return;
    uint64_t pos = w * 32;
    const uint64_t end = pos + 32 < total_bases ? pos + 32 : total_bases;
    uint64_t lo = 0, hi = n_nodes;                                   // largest node with out_start <= pos (zero-length nodes: the last of them)
    while (hi - lo > 1) { const uint64_t m = (lo + hi) >> 1; if (out_start[m] <= pos) lo = m; else hi = m; }
    uint64_t node = lo, acc = 0;
    while (pos < end) {
        const uint64_t o = pos - out_start[node], len = length[node];
        if (o >= len) { node++; continue; }
        const uint64_t take = (len - o < end - pos) ? len - o : end - pos;       // 1..32 bases of this node
        const uint64_t sp = src_start[node] + o;
        const uint64_t w0 = src_words[sp >> 5], w1 = src_words[(sp >> 5) + 1];   // (the source arrays carry a zero word of slack)
        const uint32_t sh = 2u * (uint32_t)(sp & 31);
        uint64_t v = sh ? (w0 << sh) | (w1 >> (64u - sh)) : w0;                  // bases from sp on, left-aligned
        if (take < 32) v &= ~0ull << (64u - 2u * (uint32_t)take);
        acc |= v >> (2u * (uint32_t)(pos & 31));
        pos += take;
    }
    out[w] = acc;
}

// graphs in order -> one graph (inputs are consumed)
int graph_dev_combine(dbg_ctx* c, std::vector<GraphDev*>& gs, GraphDev* out) {
    uint64_t n = 0, nw_src = 0;
    bool all_s = true, none_s = true;
    for (GraphDev* g : gs) { n += g->n_nodes; nw_src += g->n_words + 1; all_s = all_s && g->stranded; none_s = none_s && !g->stranded; }
    if (!all_s && !none_s) return c->fail(50, "attempted to combine stranded and unstranded graphs (graph.rs:90)");
    if (n >= (1ull << 32)) return c->fail(51, "combine: more than 2^32-1 nodes");
    DBuf<uint64_t> src_words, src_start, out_start;
    DBuf<uint32_t> length, data;
    DBuf<uint8_t> exts;
    const uint64_t na = std::max<uint64_t>(n, 1);
    ALLOC_OR_FAIL(c, src_words, nw_src + 2); ALLOC_OR_FAIL(c, src_start, na); ALLOC_OR_FAIL(c, out_start, n + 1);
    ALLOC_OR_FAIL(c, length, na); ALLOC_OR_FAIL(c, data, na); ALLOC_OR_FAIL(c, exts, na);
    HIP_TRY(c, hipMemsetAsync(src_words.p, 0, (nw_src + 2) * 8, c->stream));
    uint64_t no = 0, wo = 0;
    for (GraphDev* g : gs) {
        const uint64_t gn = g->n_nodes;
        if (g->n_words) HIP_TRY(c, hipMemcpyAsync(src_words.p + wo, g->words.p, g->n_words * 8, hipMemcpyDeviceToDevice, c->stream));
        if (gn) {
            add_offset_kernel<<<cdiv(gn, 256), 256, 0, c->stream>>>(g->start.p, gn, wo * 32, src_start.p + no);
            LAUNCH_CHECK(c, "add_offset");
            HIP_TRY(c, hipMemcpyAsync(length.p + no, g->length.p, gn * 4, hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(c, hipMemcpyAsync(data.p + no, g->data.p, gn * 4, hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(c, hipMemcpyAsync(exts.p + no, g->exts.p, gn, hipMemcpyDeviceToDevice, c->stream));
        }
        no += gn; wo += g->n_words + 1;
    }
    DBG_TRY(scan_exclusive_u32_u64(c, length.p, out_start.p, n));
    uint64_t total = 0;
    HIP_TRY(c, hipMemcpyAsync(&total, out_start.p + n, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (GraphDev* g : gs) graph_dev_clear(g);
    const uint64_t nw = (total + 31) / 32;
    DBuf<uint64_t> words;
    ALLOC_OR_FAIL(c, words, nw + 3);
    HIP_TRY(c, hipMemsetAsync(words.p + nw, 0, 3 * 8, c->stream));
    if (nw) {
        c->t_begin("graph_combine", n);
        combine_pack_kernel<<<cdiv(nw, 256), 256, 0, c->stream>>>(src_words.p, src_start.p, length.p, out_start.p, n, nw, total, words.p);
        c->t_end();
        LAUNCH_CHECK(c, "combine_pack");
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    out->words = std::move(words); out->start = std::move(out_start); out->length = std::move(length); out->data = std::move(data);
    out->exts = std::move(exts);
    out->n_nodes = n; out->n_words = nw; out->n_bases = total; out->stranded = all_s ? 1 : 0; out->filled = true;
    return 0;
}

__global__ void __launch_bounds__(256) short_node_kernel(const uint32_t* __restrict__ length, uint32_t n, uint32_t k, uint32_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && length[i] < k) *flag = 1u;
}

// compress_graph (compression.rs:338-349) on a device-resident graph, no censored nodes: the device route of dbg_compress_graph
// with nothing staged through the host.  `in` is consumed.  Links that are not mutual / carry a panic marker take the literal
// host walk of dbg_compress_graph (the graph then does make the trip; such inputs are those the reference panics on).
// partial (an inner level of the tree merge): the graph holds SOME of the shards, so an extension whose k-mer is in no node
// points into a shard that has not arrived yet: such a path ends there for now, and neither fix_exts runs -- stripping the Exts
// that have no target yet (compression.rs:309, :331) would cut the graph at the shard boundary for good.
int graph_dev_compress(dbg_ctx* c, int k, int stranded, int spec, GraphDev* in, GraphDev* out, bool partial = false) {
    const uint64_t n64 = in->n_nodes;
    if (n64 >= (1ull << 30)) return c->fail(51, "compress_graph: at most 2^30-1 nodes per call in this build");
    const uint32_t n = (uint32_t)n64;
    const int g_stranded = in->stranded;
    if (!n) { *out = std::move(*in); out->stranded = stranded ? 1 : 0; return 0; }
    DBuf<uint32_t> flag;
    ALLOC_OR_FAIL(c, flag, 1);
    HIP_TRY(c, hipMemsetAsync(flag.p, 0, 4, c->stream));
    short_node_kernel<<<cdiv(n, 256), 256, 0, c->stream>>>(in->length.p, n, (uint32_t)k, flag.p);
    LAUNCH_CHECK(c, "short_node");
    uint32_t fl = 0;
    HIP_TRY(c, hipMemcpyAsync(&fl, flag.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (fl) return c->fail(52, "node shorter than k");
    DevGraph d;
    d.n = n;
    d.words = std::move(in->words); d.start = std::move(in->start); d.length = std::move(in->length);
    d.exts = std::move(in->exts); d.data = std::move(in->data);
    const uint64_t in_words = in->n_words, in_bases = in->n_bases;
    DBG_TRY(dev_graph_index(c, k, &d));
    if (!partial) DBG_TRY(dev_fix_exts(c, k, g_stranded, &d, nullptr));                // old_graph.fix_exts(Some(&available)), all available :309
    DBuf<uint32_t> d_link, u_link, u_weight;
    ALLOC_OR_FAIL(c, d_link, 2 * (size_t)n); ALLOC_OR_FAIL(c, u_link, 2 * (size_t)n); ALLOC_OR_FAIL(c, u_weight, n);
    c->t_begin("graph_node_links", n);
    node_link_kernel<<<cdiv(n, 256), 256, 0, c->stream>>>(d.left, d.right, d.words.p, d.start.p, d.length.p, n, k, stranded, g_stranded, spec,
                                                           d.exts.p, d.data.p, d_link.p, partial ? 1 : 0);
    c->t_end();
    LAUNCH_CHECK(c, "node_links");
    bool done = false;
    const char* mode = c->opt("DBG_COMPRESS");
    GraphDev sink;
    if (partial || !(mode && !strcmp(mode, "host"))) {
        node_links_to_unitig_kernel<<<cdiv(2 * (uint64_t)n, 256), 256, 0, c->stream>>>(d_link.p, d.length.p, n, k, u_link.p, u_weight.p);
        LAUNCH_CHECK(c, "node_links_to_unitig");
        UnitigNodes un{u_weight.p, d.words.p, d.start.p, d.length.p, nullptr};
        dbg_graph ng;
        c->graph_sink = &sink;
        const int r = compress_links_device(c, k, n, nullptr, nullptr, d.exts.p, d.data.p, u_link.p, nullptr, spec, stranded, &ng, &done, &un);
        c->graph_sink = nullptr;
        if (r) return r;
        if (!done && partial)
            return c->fail(48, "tree reduce: the node links of a partial merge are not mutual (inconsistent Exts: input on which the reference's compress_graph "
                               "panics or depends on visiting order); DBG_REDUCE_GATHER takes the literal walk for such input");
        if (!done && mode && !strcmp(mode, "device")) return c->fail(48, "DBG_COMPRESS=device but the node links are not mutual");
    }
    if (!done) {
        // the literal walk (and the reference's panics): through the host entry point
        // (d.exts holds the Exts fix_exts has already rewritten; the host entry point applies fix_exts again, which changes
        //  nothing: an extension that links to a node still does)
        GraphDev back;
        back.words = std::move(d.words); back.start = std::move(d.start); back.length = std::move(d.length); back.exts = std::move(d.exts);
        back.data = std::move(d.data); back.n_nodes = n; back.n_words = in_words; back.n_bases = in_bases; back.stranded = g_stranded;
        dbg_graph hg, ho;
        DBG_TRY(graph_dev_to_host(c, back, &hg));
        graph_dev_clear(&back);
        const int r = dbg_compress_graph(c, (uint32_t)k, stranded, spec, &hg, nullptr, 0, &ho);
        dbg_free_graph(c, &hg);
        if (r) return r;
        const int r2 = graph_dev_from_host(c, &ho, out);
        dbg_free_graph(c, &ho);
        return r2;
    }
    // ---- graph.finish(); dbg.fix_exts(None) (compression.rs:330-331) ----
    DevGraph d2;
    d2.n = (uint32_t)sink.n_nodes;
    d2.words = std::move(sink.words); d2.start = std::move(sink.start); d2.length = std::move(sink.length);
    d2.exts = std::move(sink.exts); d2.data = std::move(sink.data);
    if (!partial) {
        DBG_TRY(dev_graph_index(c, k, &d2));
        DBG_TRY(dev_fix_exts(c, k, stranded, &d2, nullptr));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    out->words = std::move(d2.words); out->start = std::move(d2.start); out->length = std::move(d2.length);
    out->exts = std::move(d2.exts); out->data = std::move(d2.data);
    out->n_nodes = sink.n_nodes; out->n_words = sink.n_words; out->n_bases = sink.n_bases; out->stranded = stranded ? 1 : 0; out->filled = true;
    return 0;
}

// ---- graphs as device buffers over the transport ------------------------------------------------------------------------
// A graph travels in two steps with an agreement of all ranks in between (shard_comm.hpp): first its sizes, so that the receiver
// can reserve its buffers -- the one thing that may fail on its side -- and only when every rank has what it needs, the arrays.
struct GraphMeta { uint64_t n_nodes, n_words, n_bases, stranded; };

int graph_send_meta(dbg_ctx* c, ShardComm& X, const GraphDev& g, int32_t peer) {
    const dbg_transport* tr = X.tr;
    DBuf<uint64_t> meta;
    if (!meta.alloc(c, 4)) { X.abort(); return c->fail(101, "sharded compress: no device memory for the graph sizes (communicator aborted)"); }
    const uint64_t m[4] = {g.n_nodes, g.n_words, g.n_bases, (uint64_t)g.stranded};
    if (hipMemcpyAsync(meta.p, m, 32, hipMemcpyHostToDevice, c->stream) != hipSuccess) { (void)hipGetLastError(); return X.op_failed("copy of the graph sizes"); }
    if (tr->send(tr->self, meta.p, 32, peer, c->stream)) return X.op_failed("send (graph sizes)");
    return X.wait_stream(c->stream, "the graph sizes to be sent");
}
int graph_recv_meta(dbg_ctx* c, ShardComm& X, GraphMeta* gm, int32_t peer) {
    const dbg_transport* tr = X.tr;
    DBuf<uint64_t> meta;
    if (!meta.alloc(c, 4)) { X.abort(); return c->fail(101, "sharded compress: no device memory for the graph sizes (communicator aborted)"); }
    if (tr->recv(tr->self, meta.p, 32, peer, c->stream)) return X.op_failed("recv (graph sizes)");
    uint64_t m[4] = {0, 0, 0, 0};
    if (hipMemcpyAsync(m, meta.p, 32, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { (void)hipGetLastError(); return X.op_failed("copy of the graph sizes"); }
    DBG_TRY(X.wait_stream(c->stream, "the graph sizes of a peer"));
    gm->n_nodes = m[0]; gm->n_words = m[1]; gm->n_bases = m[2]; gm->stranded = m[3];
    return 0;
}
// rank-local: room for a graph of these sizes
int graph_dev_reserve(dbg_ctx* c, const GraphMeta& gm, GraphDev* g) {
    const uint64_t n = gm.n_nodes, nw = gm.n_words;
    ALLOC_OR_FAIL(c, g->words, nw + 3); ALLOC_OR_FAIL(c, g->start, n + 1); ALLOC_OR_FAIL(c, g->length, std::max<uint64_t>(n, 1));
    ALLOC_OR_FAIL(c, g->exts, std::max<uint64_t>(n, 1)); ALLOC_OR_FAIL(c, g->data, std::max<uint64_t>(n, 1));
    HIP_TRY(c, hipMemsetAsync(g->words.p + nw, 0, 3 * 8, c->stream));
    return 0;
}
int graph_send_data(dbg_ctx* c, ShardComm& X, const GraphDev& g, int32_t peer) {
    const dbg_transport* tr = X.tr;
    if (g.n_words && tr->send(tr->self, g.words.p, g.n_words * 8, peer, c->stream)) return X.op_failed("send (sequence words)");
    if (g.n_nodes) {
        if (tr->send(tr->self, g.start.p, g.n_nodes * 8, peer, c->stream) || tr->send(tr->self, g.length.p, g.n_nodes * 4, peer, c->stream) ||
            tr->send(tr->self, g.exts.p, g.n_nodes, peer, c->stream) || tr->send(tr->self, g.data.p, g.n_nodes * 4, peer, c->stream))
            return X.op_failed("send (node arrays)");
    }
    return X.wait_stream(c->stream, "a shard graph to be sent");
}
int graph_recv_data(dbg_ctx* c, ShardComm& X, const GraphMeta& gm, GraphDev* g, int32_t peer) {
    const dbg_transport* tr = X.tr;
    const uint64_t n = gm.n_nodes, nw = gm.n_words;
    if (nw && tr->recv(tr->self, g->words.p, nw * 8, peer, c->stream)) return X.op_failed("recv (sequence words)");
    if (n) {
        if (tr->recv(tr->self, g->start.p, n * 8, peer, c->stream) || tr->recv(tr->self, g->length.p, n * 4, peer, c->stream) ||
            tr->recv(tr->self, g->exts.p, n, peer, c->stream) || tr->recv(tr->self, g->data.p, n * 4, peer, c->stream))
            return X.op_failed("recv (node arrays)");
    }
    DBG_TRY(X.wait_stream(c->stream, "a shard graph of a peer"));
    g->n_nodes = n; g->n_words = nw; g->n_bases = gm.n_bases; g->stranded = (int)gm.stranded; g->filled = true;
    return 0;
}

// ---- label-list classes of all ranks -> one numbering ----------------------------------------------------------------------
__global__ void __launch_bounds__(256) remap_u32_kernel(uint32_t* __restrict__ data, uint64_t n, const uint32_t* __restrict__ map, uint32_t n_map,
                                                         uint32_t* __restrict__ flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = data[i];
    if (v < n_map) data[i] = map[v]; else *flag = 1u;
}

// all_gather of one variable-size host blob per rank (sizes first, then the blobs padded to the largest).  `lrc`: this rank's
// status so far; the ranks agree on it (and on their allocations) between the two steps.
int gather_blobs(dbg_ctx* c, ShardComm& X, int lrc, const std::vector<uint8_t>& mine, std::vector<std::vector<uint8_t>>* all) {
    const dbg_transport* tr = X.tr;
    const uint32_t W = X.W;
    all->assign(W, {});
    if (W == 1) { (*all)[0] = mine; return lrc; }
    if (!tr->all_gather) return c->fail(161, "sharded compress: the transport lacks all_gather");
    // (the two size words come out of the status reserve: nothing can fail locally before the first collective)
    uint64_t* d_sz = X.word.p + 6;
    DBuf<uint64_t> d_all;
    if (!lrc && !d_all.alloc(c, W)) lrc = c->fail(101, "sharded compress: no device memory for the class-table sizes");
    DBG_TRY(X.agree(lrc, "classes (sizes)"));
    const uint64_t sz = mine.size();
    if (hipMemcpyAsync(d_sz, &sz, 8, hipMemcpyHostToDevice, c->stream) != hipSuccess) { (void)hipGetLastError(); return X.op_failed("copy of the class-table size"); }
    if (tr->all_gather(tr->self, d_sz, d_all.p, 8, c->stream)) return X.op_failed("all_gather (class table sizes)");
    std::vector<uint64_t> sizes(W);
    if (hipMemcpyAsync(sizes.data(), d_all.p, (size_t)W * 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { (void)hipGetLastError(); return X.op_failed("copy of the class-table sizes"); }
    DBG_TRY(X.wait_stream(c->stream, "the class-table sizes of the other ranks"));
    uint64_t mx = 0;
    for (uint64_t v : sizes) mx = std::max(mx, v);
    mx = (mx + 7) & ~7ull;
    if (!mx) return 0;
    DBuf<uint8_t> d_mine, d_blobs;
    lrc = [&]() -> int {
        if (X.inject("classes")) return X.injected("classes");
        ALLOC_OR_FAIL(c, d_mine, mx); ALLOC_OR_FAIL(c, d_blobs, mx * W);
        HIP_TRY(c, hipMemsetAsync(d_mine.p, 0, mx, c->stream));
        if (sz) HIP_TRY(c, hipMemcpyAsync(d_mine.p, mine.data(), sz, hipMemcpyHostToDevice, c->stream));
        return 0;
    }();
    DBG_TRY(X.agree(lrc, "classes (tables)"));
    if (tr->all_gather(tr->self, d_mine.p, d_blobs.p, mx, c->stream)) return X.op_failed("all_gather (class tables)");
    std::vector<uint8_t> flat(mx * W);
    if (hipMemcpyAsync(flat.data(), d_blobs.p, mx * W, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { (void)hipGetLastError(); return X.op_failed("copy of the class tables"); }
    DBG_TRY(X.wait_stream(c->stream, "the class tables of the other ranks"));
    for (uint32_t r = 0; r < W; r++) (*all)[r].assign(flat.begin() + r * mx, flat.begin() + r * mx + sizes[r]);
    return 0;
}

}  // namespace

// compress_graph's device route on a device-resident graph without either fix_exts: the second level of unitig.hip's segment route
// (segments of long chains -> unitigs; a segment end whose extension leads into no node end is a chain end, as at the k-mer level)
int graph_dev_join_segments(dbg_ctx* c, int k, int stranded, int spec, GraphDev* in, GraphDev* out) {
    return graph_dev_compress(c, k, stranded, spec, in, out, true);
}

extern "C" int dbg_shard_compress_dev(dbg_ctx* c, const dbg_transport* tr, uint32_t k, int stranded, int spec, int second_spec,
                                      const dbg_kmer_table* table, int32_t reduce, int32_t root, dbg_graph* final_out, dbg_graph* local_out,
                                      dbg_label_classes* classes) {
    if (!c) return 1;
    if (!table || !final_out) return c->fail(10, "null argument");
    if (!table->on_device) return c->fail(162, "dbg_shard_compress_dev needs a device-resident table");
    const uint32_t W = tr ? (uint32_t)tr->world : 1u, me = tr ? (uint32_t)tr->rank : 0u;
    if (W == 0 || me >= W || root < 0 || (uint32_t)root >= W) return c->fail(161, "sharded compress: bad rank / world / root");
    if (W > 1 && (!tr->send || !tr->recv || !tr->all_reduce_u64)) return c->fail(161, "sharded compress: the transport lacks send / recv / all_reduce_u64");
    if (reduce != DBG_REDUCE_GATHER && reduce != DBG_REDUCE_TREE) return c->fail(161, "sharded compress: unknown reduce mode");
    if (second_spec < 0) second_spec = spec;
    HIP_TRY(c, hipSetDevice(c->device));
    memset(final_out, 0, sizeof(*final_out));
    if (local_out) memset(local_out, 0, sizeof(*local_out));
    if (classes) memset(classes, 0, sizeof(*classes));
    // Failure agreement (shard_comm.hpp): rank-local work keeps its status in `lrc`; every phase ends in X.agree() before the
    // phase's data moves, so that all ranks leave with the same verdict and nobody waits for a graph that will not come.
    ShardComm X(c, tr);
    DBG_TRY(X.prepare());
    int lrc = 0;
    // per-phase host time of this rank (DBG_DEBUG: one line per phase on stderr; tools/rehearse_shard.py collects them)
    const bool dbg_t = c->opt("DBG_DEBUG") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t_last = now();
    auto phase_done = [&](const char* name, uint64_t nodes) {
        if (!dbg_t) return;
        (void)hipStreamSynchronize(c->stream);
        const auto t = now();
        fprintf(stderr, "[shard_compress] rank=%u phase=%s ms=%.2f nodes=%llu\n", me, name, std::chrono::duration<double, std::milli>(t - t_last).count(),
                (unsigned long long)nodes);
        t_last = t;
    };
    auto release_outputs = [&]() {
        if (local_out) { dbg_free_graph(c, local_out); memset(local_out, 0, sizeof(*local_out)); }
        if (classes) { dbg_free_label_classes(classes); memset(classes, 0, sizeof(*classes)); }
        dbg_free_graph(c, final_out); memset(final_out, 0, sizeof(*final_out));
    };

    // ---- phase "shard": this rank's compress_kmers_with_hash on the table of the bins it owns; the graph stays in HBM ----
    GraphDev mine;
    dbg_label_classes lc;
    memset(&lc, 0, sizeof(lc));
    lrc = [&]() -> int {
        if (X.inject("shard")) return X.injected("shard");
        dbg_graph hg;
        memset(&hg, 0, sizeof(hg));
        c->graph_sink = &mine;
        const int r = dbg_compress_table_dev(c, k, stranded, spec, table, &hg, table->set_off ? &lc : nullptr);
        c->graph_sink = nullptr;
        if (r) return r;
        if (!mine.filled) {                     // the literal host walk ran (or the table is empty): the result is on the host
            if (hg.n_nodes) { const int r2 = graph_dev_from_host(c, &hg, &mine); if (r2) { dbg_free_graph(c, &hg); return r2; } }
            else { GraphDev e; if (int r3 = graph_dev_from_host(c, &hg, &e)) return r3; mine = std::move(e); }
            mine.stranded = stranded ? 1 : 0;
            dbg_free_graph(c, &hg);
        }
        return 0;
    }();
    // ---- label-list classes: one numbering for all ranks (ranks of the sorted distinct lists) ----
    if (table->set_off) {
        std::vector<uint8_t> blob(8 + (lc.n_classes + 1) * 8 + lc.n_set_val * 4);
        memcpy(blob.data(), &lc.n_classes, 8);
        if (lc.set_off) memcpy(blob.data() + 8, lc.set_off, (lc.n_classes + 1) * 8);       // (else: the one zero offset the vector already holds)
        if (lc.n_set_val) memcpy(blob.data() + 8 + (lc.n_classes + 1) * 8, lc.set_val, lc.n_set_val * 4);
        std::vector<std::vector<uint8_t>> all;
        const int r = gather_blobs(c, X, lrc, blob, &all);                                 // (agrees on `lrc` before anything moves)
        if (r) { dbg_free_label_classes(&lc); return r; }
        typedef std::vector<uint32_t> List;
        std::vector<std::vector<List>> tabs(W);
        std::vector<List> glob;
        for (uint32_t rk = 0; rk < W; rk++) {
            const uint8_t* b = all[rk].data();
            uint64_t nc = 0;
            if (all[rk].size() >= 8) memcpy(&nc, b, 8);
            const uint64_t* off = (const uint64_t*)(b + 8);
            const uint32_t* val = (const uint32_t*)(b + 8 + (nc + 1) * 8);
            for (uint64_t i = 0; i < nc; i++) { tabs[rk].emplace_back(val + off[i], val + off[i + 1]); glob.push_back(tabs[rk].back()); }
        }
        std::sort(glob.begin(), glob.end());                       // lexicographic, a prefix before its extensions: the order of Rust's Vec<D1> Ord
        glob.erase(std::unique(glob.begin(), glob.end()), glob.end());
        std::vector<uint32_t> remap(tabs[me].size());
        for (size_t i = 0; i < remap.size(); i++) remap[i] = (uint32_t)(std::lower_bound(glob.begin(), glob.end(), tabs[me][i]) - glob.begin());
        lrc = [&]() -> int {
            if (!mine.n_nodes) return 0;
            DBuf<uint32_t> d_map, fl;
            ALLOC_OR_FAIL(c, d_map, std::max<size_t>(remap.size(), 1)); ALLOC_OR_FAIL(c, fl, 1);
            HIP_TRY(c, hipMemsetAsync(fl.p, 0, 4, c->stream));
            if (!remap.empty()) HIP_TRY(c, hipMemcpyAsync(d_map.p, remap.data(), remap.size() * 4, hipMemcpyHostToDevice, c->stream));
            remap_u32_kernel<<<cdiv(mine.n_nodes, 256), 256, 0, c->stream>>>(mine.data.p, mine.n_nodes, d_map.p, (uint32_t)remap.size(), fl.p);
            LAUNCH_CHECK(c, "remap_u32");
            uint32_t bad = 0;
            HIP_TRY(c, hipMemcpyAsync(&bad, fl.p, 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (bad) return c->fail(164, "sharded compress: a node's class id is outside its rank's class table");
            return 0;
        }();
        dbg_free_label_classes(&lc);
        if (classes && !lrc) {
            uint64_t nv = 0;
            for (auto& l : glob) nv += l.size();
            classes->n_classes = glob.size(); classes->n_set_val = nv;
            classes->set_off = (uint64_t*)malloc((glob.size() + 1) * 8);
            classes->set_val = (uint32_t*)malloc(std::max<uint64_t>(nv, 1) * 4);
            if (!classes->set_off || !classes->set_val) {
                free(classes->set_off); free(classes->set_val);
                classes->set_off = nullptr; classes->set_val = nullptr; classes->n_classes = 0; classes->n_set_val = 0;
                lrc = c->fail(102, "sharded compress: host allocation of the class table failed");
            }
            uint64_t o = 0;
            for (size_t i = 0; i < glob.size() && !lrc; i++) { classes->set_off[i] = o; if (!glob[i].empty()) memcpy(classes->set_val + o, glob[i].data(), glob[i].size() * 4); o += glob[i].size(); }
            if (!lrc) classes->set_off[glob.size()] = o;
        }
    }
    if (local_out && !lrc) lrc = graph_dev_to_host(c, mine, local_out);
    phase_done("shard_compress", mine.n_nodes);
    if (int r = X.agree(lrc, "shard")) { release_outputs(); return r; }

    // ---- merge the shard graphs ----
    // (a transport failure cannot be agreed on -- the peers end in their bounded waits -- but this rank's outputs are released like on
    //  every other failure path: a caller that treats outputs as invalid on error must not leak them)
#define XFER_TRY(expr) do { if (int r_ = (expr)) { release_outputs(); return r_; } } while (0)
    GraphDev result;
    bool have_result = false;
    if (reduce == DBG_REDUCE_GATHER || W == 1) {
        // phase "gather (sizes)": the sizes go to root, root reserves room for every shard graph
        std::vector<GraphDev> got(W);
        std::vector<GraphMeta> gm(W);
        if (me != (uint32_t)root) XFER_TRY(graph_send_meta(c, X, mine, root));
        else {
            for (uint32_t r = 0; r < W; r++) if (r != me) XFER_TRY(graph_recv_meta(c, X, &gm[r], (int32_t)r));
            lrc = [&]() -> int {
                if (W > 1 && X.inject("gather")) return X.injected("gather");
                for (uint32_t r = 0; r < W; r++) if (r != me) DBG_TRY(graph_dev_reserve(c, gm[r], &got[r]));
                return 0;
            }();
        }
        if (int r = X.agree(lrc, "gather (sizes)")) { release_outputs(); return r; }
        if (me != (uint32_t)root) XFER_TRY(graph_send_data(c, X, mine, root));
        else {
            std::vector<GraphDev*> order(W);
            for (uint32_t r = 0; r < W; r++) {
                if (r == me) order[r] = &mine;
                else { XFER_TRY(graph_recv_data(c, X, gm[r], &got[r], (int32_t)r)); order[r] = &got[r]; }
            }
            phase_done("gather_transfer", 0);
            lrc = [&]() -> int {
                if (X.inject("merge")) return X.injected("merge");
                GraphDev comb;
                DBG_TRY(graph_dev_combine(c, order, &comb));                                       // BaseGraph::combine(shard graphs in shard order), test.rs:468
                phase_done("combine", comb.n_nodes);
                DBG_TRY(graph_dev_compress(c, (int)k, stranded, second_spec, &comb, &result));     // compress_graph, test.rs:469
                phase_done("compress_graph", result.n_nodes);
                return 0;
            }();
            have_result = !lrc;
        }
    } else {
        // binary tree over ranks renumbered so that `root` is 0: at level l, position q with bit l set sends to q - 2^l and leaves.
        // Every rank takes part in every level's agreement, whether or not it still holds a graph.
        const uint32_t pos = (me + W - (uint32_t)root) % W;
        bool active = true;
        for (uint32_t st = 1; st < W; st <<= 1) {
            const bool sender = active && (pos & st), receiver = active && !(pos & st) && pos + st < W;
            const uint32_t to = ((pos - st) + (uint32_t)root) % W, from = ((pos + st) + (uint32_t)root) % W;
            GraphDev other;
            GraphMeta gm{0, 0, 0, 0};
            // (a rank whose own merge failed at the level below still exchanges sizes, so that its partner is not left waiting; the
            //  status is agreed right after)
            if (sender) XFER_TRY(graph_send_meta(c, X, mine, (int32_t)to));
            else if (receiver) {
                XFER_TRY(graph_recv_meta(c, X, &gm, (int32_t)from));
                if (!lrc) lrc = (X.inject("tree") && st == 1) ? X.injected("tree") : graph_dev_reserve(c, gm, &other);
            }
            if (int r = X.agree(lrc, "tree (sizes)")) { release_outputs(); return r; }
            if (sender) {
                XFER_TRY(graph_send_data(c, X, mine, (int32_t)to));
                graph_dev_clear(&mine);
                active = false;
            } else if (receiver) {
                XFER_TRY(graph_recv_data(c, X, gm, &other, (int32_t)from));
                phase_done("tree_transfer", other.n_nodes);
                lrc = [&]() -> int {
                    GraphDev comb, merged;
                    std::vector<GraphDev*> pair{&mine, &other};
                    DBG_TRY(graph_dev_combine(c, pair, &comb));
                    phase_done("tree_combine", comb.n_nodes);
                    // every merge but the root's last one sees a part of the shards only
                    const bool last = pos == 0 && 2 * st >= W;
                    DBG_TRY(graph_dev_compress(c, (int)k, stranded, second_spec, &comb, &merged, !last));
                    phase_done(last ? "tree_compress_graph_last" : "tree_compress_graph_partial", merged.n_nodes);
                    mine = std::move(merged);
                    return 0;
                }();
            }
        }
        if (pos == 0 && !lrc) {
            if (W == 1) {
                lrc = [&]() -> int { GraphDev comb; std::vector<GraphDev*> one{&mine}; DBG_TRY(graph_dev_combine(c, one, &comb)); return graph_dev_compress(c, (int)k, stranded, second_spec, &comb, &result); }();
            } else result = std::move(mine);
            have_result = !lrc;
        }
    }
    if (have_result) { lrc = graph_dev_to_host(c, result, final_out); phase_done("graph_to_host", result.n_nodes); }
    else if (!lrc) final_out->stranded = stranded ? 1 : 0;
    if (int r = X.agree(lrc, "merge")) { release_outputs(); return r; }
    return 0;
}
