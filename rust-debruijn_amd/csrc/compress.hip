// Path compression: compress_kmers_with_hash (src/compression.rs:355-594) and the neighbour-probe
// clean-up remove_censored_exts(_sharded) (src/filter.rs:238-306).
//
// Device: for every k-mer and both directions, one thread evaluates everything in
// CompressFromHash::try_extend_kmer (compression.rs:382-444) that does not depend on the greedy
// `available_kmers` state -- unique extension, canonicalisation + direction flip, presence of the
// neighbour (binary search in the ascending key array replaces the BoomHashMap2 probe), its incoming
// extension count, join_test and the palindrome rules -- and stores the result as a 32-bit link
// ("GPU-resident neighbour bitmaps").  Host: the branchy greedy walk (extend_kmer / build_node /
// compress_kmers, compression.rs:450-583) follows those links in the reference's seed order.
#include "dbg_internal.hpp"
#include <algorithm>
#include <numeric>
#include <deque>

namespace {
constexpr uint32_t LINK_TERM = 0xFFFFFFFFu;     // ExtMode::Terminal regardless of availability
constexpr uint32_t LINK_PANIC = 0x80000000u;    // incoming_count == 0 && !palindrome (compression.rs:428-434) if reached
// otherwise: (neighbour id << 1) | next_dir   (0 = Left, 1 = Right)

struct KeysDev {
    const uint64_t* hi;     // null when k <= 32
    const uint64_t* lo;
    uint64_t n;
    // optional prefix index over the ascending keys: pidx[p] = first position whose top `pbits` key bits are >= p
    // (2^pbits + 1 entries).  Turns the 27-step binary search over 10^8 keys into a 2-3 step one inside a bucket.
    const uint32_t* pidx = nullptr;
    int pbits = 0, key_bits = 0;
    // optional 16-byte records {lo, hi | Exts << 56} (k <= 60: hi holds at most 56 key bits): a neighbour probe then touches
    // one cache line for key and Exts together instead of three (hi, lo, exts arrays); measured 195 -> ~100 bytes per probe
    const ulonglong2* rec = nullptr;
};

__device__ __forceinline__ K128 key_at(const KeysDev& t, uint64_t i) { return K128{t.hi ? t.hi[i] : 0ull, t.lo[i]}; }

__device__ __forceinline__ uint32_t key_prefix_bits(K128 a, int key_bits, int pbits) {
    return (uint32_t)k128_shr(a, key_bits - pbits).lo;
}
// index of `q` in the ascending key array, or -1
__device__ __forceinline__ int64_t find_key(const KeysDev& t, K128 q) {
    uint64_t lo = 0, hi = t.n;
    if (t.pidx) { const uint32_t p = key_prefix_bits(q, t.key_bits, t.pbits); lo = t.pidx[p]; hi = t.pidx[p + 1]; }
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        K128 m = key_at(t, mid);
        if (k128_lt(m, q)) lo = mid + 1; else hi = mid;
    }
    if (lo < t.n && k128_eq(key_at(t, lo), q)) return (int64_t)lo;
    return -1;
}

constexpr uint64_t REC_HI_MASK = (1ull << 56) - 1;
// the same over the packed records; *exts_out = the found k-mer's Exts
__device__ __forceinline__ int64_t find_key_rec(const KeysDev& t, K128 q, uint32_t* exts_out) {
    uint64_t lo = 0, hi = t.n;
    if (t.pidx) { const uint32_t p = key_prefix_bits(q, t.key_bits, t.pbits); lo = t.pidx[p]; hi = t.pidx[p + 1]; }
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        const ulonglong2 r = t.rec[mid];
        if (k128_lt(K128{r.y & REC_HI_MASK, r.x}, q)) lo = mid + 1; else hi = mid;
    }
    if (lo < t.n) {
        const ulonglong2 r = t.rec[lo];
        if (r.x == q.lo && (r.y & REC_HI_MASK) == q.hi) { *exts_out = (uint32_t)(r.y >> 56); return (int64_t)lo; }
    }
    return -1;
}
// (sorted_flag, optional: |= 1 unless the keys are strictly ascending -- the check the device-resident entry point needs, made
//  while the keys are in registers anyway instead of in a pass of its own)
__global__ void pack_keys_kernel(KeysDev t, const uint8_t* __restrict__ exts, ulonglong2* __restrict__ rec, uint32_t* __restrict__ sorted_flag = nullptr) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.n) return;
    const uint64_t lo = t.lo[i], hi = t.hi ? t.hi[i] : 0ull;
    rec[i] = make_ulonglong2(lo, hi | ((uint64_t)exts[i] << 56));
    if (sorted_flag && i && !k128_lt(key_at(t, i - 1), K128{hi, lo})) atomicOr(sorted_flag, 1u);
}

__device__ __forceinline__ bool is_palindrome(K128 a, int k) { return (k & 1) == 0 && k128_eq(a, kmer_rc(a, k)); }   // lib.rs:244-246
__device__ __forceinline__ uint32_t num_ext_dir(uint32_t e, int dir) { return __popc((e >> (4 * dir)) & 0xfu); }      // lib.rs:687-690
__device__ __forceinline__ bool join_test(int spec, uint32_t a, uint32_t b) { return spec == DBG_SPEC_SCMAP_EQ ? a == b : true; }

// One direction of one k-mer: everything of try_extend_kmer (compression.rs:382-444) that does not depend on the greedy state.
// Every neighbour relation is seen from both ends -- k-mer i looks its neighbour j up, j looks i up -- and a lookup is two or
// three random cache lines in a 10 GB table: the kernel's whole cost.  With consistent Exts the second lookup tells nothing new:
// when i (unique extension towards j) has found j, j's Exts say whether j's own link in the facing direction D is the mirror
// image (j has exactly one extension there and it is the base that leads back to i).  So the lookup is made from ONE end, the
// end with the smaller key ("designated"; the keys are known before the lookup), which also writes the facing link of j when it
// is derivable; a second pass evaluates, with its own lookup, every link that is still unset (the other end had a branch, no
// extension, or the Exts are not symmetric) and assembles the node records.  Half the lookups in the interior of unitigs.
//   j's evaluation in direction D, derived from i's (dir): flip' = flip, next_dir' = 1 - dir, incoming' = num_ext_dir(e_i, dir) = 1,
//   pal' = is_palindrome(i) = false, join_test is symmetric -- so its link is (i << 1 | 1 - dir) if can_join, Terminal if not;
//   valid only if i's key is canonical (non-stranded), j is no palindrome, and j's single extension in D is the expected base.
constexpr uint32_t LINK_UNSET = 0xFFFFFFFEu;   // (ids are < 2^30 - 1: the largest link word is PANIC | id << 1 | 1 <= 0xFFFFFFFD)
struct LinkEval {
    uint32_t out;          // link word of (i, dir); LINK_UNSET: not evaluated (another k-mer's thread is designated)
    bool rev_ok;           // the facing link of the neighbour follows from this evaluation
    uint32_t rev_id, rev_dir, rev_val;
};
template <bool DESIGNATED_ONLY>
__device__ __forceinline__ LinkEval eval_link(const KeysDev& t, const uint8_t* __restrict__ exts, const uint32_t* __restrict__ data, uint64_t i,
                                              K128 kmer, uint32_t e, bool self_pal, bool canon_i, int dir, int k, int stranded, int spec) {
    LinkEval r{LINK_TERM, false, 0u, 0u, 0u};
    if (num_ext_dir(e, dir) != 1 || self_pal) return r;                            // compression.rs:386
    const uint32_t bits = (e >> (4 * dir)) & 0xfu;
    const uint32_t base = 31 - __clz(bits);                                        // get_unique_extension (lib.rs:704-717)
    K128 next = dir == 0 ? kmer_extend_left(kmer, k, base) : kmer_extend_right(kmer, k, base);   // :392
    bool flip = false;
    if (!stranded) {                                                               // :396-400
        K128 rc = kmer_rc(next, k);
        if (!k128_lt(next, rc)) { next = rc; flip = true; }
    }
    if (DESIGNATED_ONLY && k128_lt(next, kmer)) { r.out = LINK_UNSET; return r; }  // the neighbour's thread is designated for this pair
    const int next_dir = flip ? 1 - dir : dir;                                     // :402
    const bool pal = !stranded && is_palindrome(next, k);                          // :403
    uint32_t ne = 0;
    const int64_t nid = t.rec ? find_key_rec(t, next, &ne) : find_key(t, next);    // :410
    if (nid < 0) return r;
    const int new_incoming_dir = flip ? dir : 1 - dir;                             // dir.flip().cond_flip(flip) :419
    if (!t.rec) ne = exts[nid];
    const uint32_t incoming = num_ext_dir(ne, new_incoming_dir);                   // :422
    // join_test :426 -- only ScmapCompress looks at the data: for the SimpleCompress specs the neighbour's D is not fetched (a fourth
    // random cache line per lookup otherwise)
    const bool can_join = spec != DBG_SPEC_SCMAP_EQ || !data || join_test(spec, data[i], data[nid]);
    if (incoming == 0 && !pal) r.out = LINK_PANIC | ((uint32_t)nid << 1) | (uint32_t)next_dir;
    else if (can_join && incoming == 1 && !pal) r.out = ((uint32_t)nid << 1) | (uint32_t)next_dir;   // :435-437
    if (DESIGNATED_ONLY && incoming == 1 && !pal && canon_i) {
        const uint32_t far = dir == 1 ? kmer_get(kmer, k, 0) : (uint32_t)(kmer.lo & 3ull);       // the base of i that j's extension must name
        const uint32_t want = flip ? 3u - far : far;
        if (((ne >> (4 * new_incoming_dir)) & 0xfu) == (1u << want)) {
            r.rev_ok = true; r.rev_id = (uint32_t)nid; r.rev_dir = (uint32_t)new_incoming_dir;
            r.rev_val = can_join ? (((uint32_t)i << 1) | (uint32_t)(1 - dir)) : LINK_TERM;
        }
    }
    return r;
}

// PHASE 0: every link from its own lookup, node records written (the plain form).  PHASE 1: designated lookups + derived facing
// links (link[] pre-set to LINK_UNSET).  PHASE 2: the links still unset from their own lookup, then the node records.
template <int PHASE>
__global__ void link_kernel(KeysDev t, const uint8_t* __restrict__ exts, const uint32_t* __restrict__ data, int k,
                            int stranded, int spec, uint32_t* __restrict__ link /* [2][n] */, NodeRec* __restrict__ nrec /* or null */) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.n) return;
    K128 kmer = key_at(t, i);
    uint32_t e = exts[i];
    bool self_pal = !stranded && is_palindrome(kmer, k);
    const bool canon_i = stranded || !k128_lt(kmer_rc(kmer, k), kmer);
    uint32_t both[2];
    for (int dir = 0; dir < 2; dir++) {
        if (PHASE == 2) {
            uint32_t v = link[(uint64_t)dir * t.n + i];
            if (v == LINK_UNSET) {
                v = eval_link<false>(t, exts, data, i, kmer, e, self_pal, canon_i, dir, k, stranded, spec).out;
                link[(uint64_t)dir * t.n + i] = v;
            }
            both[dir] = v;
            continue;
        }
        const LinkEval r = eval_link<PHASE == 1>(t, exts, data, i, kmer, e, self_pal, canon_i, dir, k, stranded, spec);
        if (r.out != LINK_UNSET) link[(uint64_t)dir * t.n + i] = r.out;
        if (PHASE == 1 && r.rev_ok) link[(uint64_t)r.rev_dir * t.n + r.rev_id] = r.rev_val;
        both[dir] = r.out;
    }
    if (nrec && PHASE != 1) {
        NodeRec r;
        r.lo = kmer.lo; r.hi = kmer.hi; r.link[0] = both[0]; r.link[1] = both[1]; r.data = data ? data[i] : 0u; r.exts = e;
        nrec[i] = r;
    }
}

// links of every k-mer (+ node records): two passes with one lookup per neighbour pair, or (DBG_LINKS=plain) one pass with two
static int build_links(dbg_ctx* c, const KeysDev& t, const uint8_t* exts_dev, const uint32_t* data_dev, int k, int stranded, int spec,
                       uint32_t* link_dev, NodeRec* nrec_dev) {
    const uint64_t n = t.n;
    const char* mode = c->opt("DBG_LINKS");
    if (mode && !strcmp(mode, "plain")) {
        link_kernel<0><<<cdiv(n, 256), 256, 0, c->stream>>>(t, exts_dev, data_dev, k, stranded, spec, link_dev, nrec_dev);
        LAUNCH_CHECK(c, "link_kernel");
        return 0;
    }
    HIP_TRY(c, hipMemsetD32Async((hipDeviceptr_t)link_dev, (int)LINK_UNSET, 2 * n, c->stream));
    link_kernel<1><<<cdiv(n, 256), 256, 0, c->stream>>>(t, exts_dev, data_dev, k, stranded, spec, link_dev, nullptr);
    LAUNCH_CHECK(c, "link_kernel<1>");
    link_kernel<2><<<cdiv(n, 256), 256, 0, c->stream>>>(t, exts_dev, data_dev, k, stranded, spec, link_dev, nrec_dev);
    LAUNCH_CHECK(c, "link_kernel<2>");
    return 0;
}

// pidx[p] = lower bound of prefix p: position i fills the entries of every prefix in (prefix(key[i-1]), prefix(key[i])]
__global__ void prefix_index_kernel(KeysDev t, int pbits, uint32_t* __restrict__ pidx) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > t.n) return;
    const uint32_t np = 1u << pbits;
    const uint32_t lo = i == 0 ? 0u : key_prefix_bits(key_at(t, i - 1), t.key_bits, pbits) + 1u;
    const uint32_t hi = i == t.n ? np : key_prefix_bits(key_at(t, i), t.key_bits, pbits);     // inclusive
    for (uint64_t p = lo; p <= hi; p++) pidx[p] = (uint32_t)i;
}

// strictly ascending keys?  (flag |= 1 if not)
__global__ void sorted_check_kernel(KeysDev t, uint32_t* __restrict__ flag) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 || i >= t.n) return;
    if (!k128_lt(key_at(t, i - 1), key_at(t, i))) atomicOr(flag, 1u);
}
__global__ void widen_u16_kernel(const uint16_t* __restrict__ in, uint64_t n, uint32_t* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}

// remove_censored_exts(_sharded) (filter.rs:238-306)
__global__ void censor_kernel(KeysDev valid, KeysDev all, uint8_t* __restrict__ exts, int k, int stranded, int sharded) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= valid.n) return;
    K128 kmer = key_at(valid, i);
    uint32_t e = exts[i], ne = 0;
    for (int dir = 0; dir < 2; dir++) {
        for (uint32_t b = 0; b < 4; b++) {
            if (!((e >> (4 * dir + b)) & 1u)) continue;
            K128 x = dir == 0 ? kmer_extend_left(kmer, k, b) : kmer_extend_right(kmer, k, b);
            if (!stranded) { K128 rc = kmer_rc(x, k); if (!k128_lt(x, rc)) x = rc; }   // min_rc (lib.rs:234-241)
            uint32_t unused;
            bool is_valid = (valid.rec ? find_key_rec(valid, x, &unused) : find_key(valid, x)) >= 0;   // (packed records: one 16-byte load per step)
            bool keep = sharded ? (is_valid || find_key(all, x) < 0) : is_valid;       // filter.rs:259-269 / :295-299
            if (keep) ne |= 1u << (4 * dir + b);
        }
    }
    exts[i] = (uint8_t)ne;
}

// compress_kmers_no_exts (compression.rs:619-659): Exts of every k-mer from eight neighbour probes into the k-mer set.  The
// reference canonicalises the neighbour with min_rc whether or not the graph is stranded (`let can = |k: K| k.min_rc()`).
__global__ void neighbour_exts_kernel(KeysDev probe, KeysDev set, uint8_t* __restrict__ exts, int k) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= probe.n) return;
    const K128 kmer = key_at(probe, i);
    uint32_t e = 0;
    for (int dir = 0; dir < 2; dir++) {
        for (uint32_t b = 0; b < 4; b++) {
            K128 x = dir == 0 ? kmer_extend_left(kmer, k, b) : kmer_extend_right(kmer, k, b);
            const K128 rc = kmer_rc(x, k);
            if (!k128_lt(x, rc)) x = rc;                                               // min_rc (lib.rs:234-241)
            if (find_key(set, x) >= 0) e |= 1u << (4 * dir + b);                       // Exts::set(dir, base) (lib.rs:640-652)
        }
    }
    exts[i] = (uint8_t)e;
}

// ---- host side -----------------------------------------------------------------------------
inline uint32_t spec_reduce(int spec, uint32_t a, uint32_t b, bool* panic) {
    switch (spec) {
        case DBG_SPEC_SIMPLE_SAT_ADD_U16: { uint32_t s = a + b; return s > 65535u ? 65535u : s; }
        case DBG_SPEC_SIMPLE_ADD_MOD_U16: return (a + b) % 65535u;
        case DBG_SPEC_SIMPLE_MAX_U16: return std::max(a, b);
        case DBG_SPEC_SCMAP_EQ: if (a != b) *panic = true; return a;                  // compression.rs:88-93
        case DBG_SPEC_SIMPLE_WRAP_ADD_U16: return (a + b) & 0xFFFFu;
    }
    return a;
}

struct BitPusher {          // DnaString::push (dna_string.rs:303-310) into a growing word vector
    std::vector<uint64_t> words;
    uint64_t len = 0;
    inline void push(uint32_t b) {
        if ((len & 31) == 0) words.push_back(0);
        words.back() |= (uint64_t)(b & 3u) << (62 - 2 * (len & 31));
        len++;
    }
};
}  // namespace

struct UnitigNodes;
int compress_links_device(dbg_ctx* c, int k, uint32_t n, const uint64_t* key_hi, const uint64_t* key_lo, const uint8_t* exts,
                          const uint32_t* data, uint32_t* link_dev, const uint32_t* rank_dev, int spec, int stranded,
                          dbg_graph* out, bool* done, const UnitigNodes* nodes = nullptr, const NodeRec* nrec = nullptr);

// packed {key, Exts} records for the link builder's probes (k <= 60); *t gets them attached
static int attach_key_records(dbg_ctx* c, KeysDev* t, int k, const uint8_t* exts_dev, DBuf<ulonglong2>* store, uint32_t* sorted_flag = nullptr) {
    if (k > 60 || !t->n || c->opt("DBG_NO_KEY_RECORDS")) return 0;
    ALLOC_OR_FAIL(c, (*store), t->n);
    pack_keys_kernel<<<cdiv(t->n, 256), 256, 0, c->stream>>>(*t, exts_dev, store->p, sorted_flag);
    LAUNCH_CHECK(c, "pack_keys");
    t->rec = store->p;
    return 0;
}

// builds the prefix index of an ascending key array (n < 2^32); *t gets the index attached
static int attach_prefix_index(dbg_ctx* c, KeysDev* t, int k, DBuf<uint32_t>* store) {
    t->key_bits = 2 * k;
    int pb = 8;
    int pb_max = 27;                                                 // ~4 keys per bucket, at most 2^27 buckets (512 MB)
    if (const char* e = c->opt("DBG_PIDX_BITS")) pb_max = std::max(8, std::min(28, atoi(e)));
    while (pb < pb_max && pb < 2 * k && (t->n >> pb) > 4) pb++;
    if (pb > 2 * k) pb = 2 * k;
    ALLOC_OR_FAIL(c, (*store), ((size_t)1 << pb) + 1);
    KeysDev plain = *t;
    plain.pidx = nullptr;
    prefix_index_kernel<<<cdiv(t->n + 1, 256), 256, 0, c->stream>>>(plain, pb, store->p);
    LAUNCH_CHECK(c, "prefix_index");
    t->pidx = store->p; t->pbits = pb;
    return 0;
}

extern "C" int dbg_compress_kmers_with_hash(dbg_ctx* c, uint32_t k_, int stranded, int spec, uint64_t n,
                                            const uint64_t* key_hi, const uint64_t* key_lo, const uint8_t* exts,
                                            const uint32_t* data, const uint64_t* seed_order, dbg_graph* out) {
    const int k = (int)k_;
    if (k < 1 || k > 64) return c->fail(40, "k must be in 1..=64");
    if (spec < 0 || spec > 4) return c->fail(41, "unknown CompressionSpec");
    if (n >= (1ull << 30)) return c->fail(42, "compress: at most 2^30-1 k-mers per call in this build");
    if (n && (!key_lo || !exts)) return c->fail(10, "null argument");
    HIP_TRY(c, hipSetDevice(c->device));
    c->t_clear();
    memset(out, 0, sizeof(*out));
    out->stranded = stranded ? 1 : 0;
    const bool has_hi = k > 32;

    // ids are positions in ascending key order; the arrays are uploaded as they are and checked on the device; only
    // an unsorted index (a BoomHashMap2 in slot order) takes the host sort + second upload
    auto key_of = [&](uint64_t i) { return K128{has_hi && key_hi ? key_hi[i] : 0ull, key_lo[i]}; };
    bool sorted = true;
    std::vector<uint64_t> s_hi, s_lo; std::vector<uint8_t> s_exts; std::vector<uint32_t> s_data;
    std::vector<uint32_t> order;        // sorted position -> caller id
    std::vector<uint32_t> rank;         // caller id -> sorted position
    std::vector<uint32_t> link;

    // ---- device: neighbour links, then (when the links are mutual) the whole unitig construction ----
    if (n) {
        DBuf<uint64_t> d_hi, d_lo; DBuf<uint8_t> d_exts; DBuf<uint32_t> d_data, d_link, d_rank, d_flag;
        if (has_hi) ALLOC_OR_FAIL(c, d_hi, n);
        ALLOC_OR_FAIL(c, d_lo, n); ALLOC_OR_FAIL(c, d_exts, n); ALLOC_OR_FAIL(c, d_link, 2 * n); ALLOC_OR_FAIL(c, d_flag, 1);
        if (data) ALLOC_OR_FAIL(c, d_data, n);
        auto upload = [&]() -> int {                               // pinned staging ring, several host threads (hostio.hip)
            std::vector<UploadJob> jobs;
            if (has_hi) {
                if (key_hi) jobs.push_back({d_hi.p, key_hi, (size_t)n * 8});
                else HIP_TRY(c, hipMemsetAsync(d_hi.p, 0, n * 8, c->stream));
            }
            jobs.push_back({d_lo.p, key_lo, (size_t)n * 8});
            jobs.push_back({d_exts.p, exts, (size_t)n});
            if (data) jobs.push_back({d_data.p, data, (size_t)n * 4});
            return staged_upload(c, jobs);
        };
        DBG_TRY(upload());
        KeysDev t{has_hi ? d_hi.p : nullptr, d_lo.p, n};
        {
            uint32_t fl = 0;
            HIP_TRY(c, hipMemsetAsync(d_flag.p, 0, 4, c->stream));
            sorted_check_kernel<<<cdiv(n, 256), 256, 0, c->stream>>>(t, d_flag.p);
            LAUNCH_CHECK(c, "sorted_check");
            HIP_TRY(c, hipMemcpyAsync(&fl, d_flag.p, 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            sorted = fl == 0;
        }
        if (!sorted) {
            order.resize(n); std::iota(order.begin(), order.end(), 0u);
            std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return k128_lt(key_of(a), key_of(b)); });
            for (uint64_t i = 1; i < n; i++)
                if (k128_eq(key_of(order[i - 1]), key_of(order[i]))) return c->fail(43, "duplicate k-mer in index");
            rank.resize(n); s_hi.resize(n); s_lo.resize(n); s_exts.resize(n); if (data) s_data.resize(n);
            for (uint64_t i = 0; i < n; i++) {
                uint32_t o = order[i]; rank[o] = (uint32_t)i;
                s_hi[i] = has_hi && key_hi ? key_hi[o] : 0; s_lo[i] = key_lo[o]; s_exts[i] = exts[o];
                if (data) s_data[i] = data[o];
            }
            key_hi = s_hi.data(); key_lo = s_lo.data(); exts = s_exts.data(); if (data) data = s_data.data();
            DBG_TRY(upload());
        }
        DBuf<uint32_t> d_pidx;
        DBG_TRY(attach_prefix_index(c, &t, k, &d_pidx));
        DBuf<ulonglong2> d_rec;
        DBG_TRY(attach_key_records(c, &t, k, d_exts.p, &d_rec));
        c->t_begin("compress_links", n);
        DBuf<NodeRec> d_nrec;                                      // optional: without it the chain walks read the separate arrays
        if (!c->opt("DBG_NO_NODE_RECORDS")) (void)d_nrec.alloc(c, n);
        const int lr = build_links(c, t, d_exts.p, d_data.p, k, stranded, spec, d_link.p, d_nrec.p);
        c->t_end();
        if (lr) return lr;
        d_rec.release();                                           // back to the pool; later users are ordered behind the kernel on the stream
        const char* mode = c->opt("DBG_COMPRESS");                 // device | host | (default) auto
        const bool want_device = !(mode && !strcmp(mode, "host"));
        if (want_device) {
            // seed rank of every sorted id: position at which the reference's loop (compression.rs:574) visits it
            std::vector<uint32_t> rank_h;
            if (seed_order || !sorted) {
                rank_h.assign(n, 0xFFFFFFFFu);
                for (uint64_t cidx = 0; cidx < n; cidx++) {
                    uint64_t sid = seed_order ? seed_order[cidx] : cidx;
                    if (sid >= n) return c->fail(44, "seed_order entry out of range");
                    uint32_t id = sorted ? (uint32_t)sid : rank[sid];
                    if (rank_h[id] != 0xFFFFFFFFu) return c->fail(47, "seed_order is not a permutation");
                    rank_h[id] = (uint32_t)cidx;
                }
                ALLOC_OR_FAIL(c, d_rank, n);
                HIP_TRY(c, hipMemcpyAsync(d_rank.p, rank_h.data(), n * 4, hipMemcpyHostToDevice, c->stream));
            }
            bool done = false;
            DBG_TRY(compress_links_device(c, k, (uint32_t)n, has_hi ? d_hi.p : nullptr, d_lo.p, d_exts.p, d_data.p, d_link.p,
                                          rank_h.empty() ? nullptr : d_rank.p, spec, stranded, out, &done, nullptr, d_nrec.p));
            if (done) return 0;
            if (mode && !strcmp(mode, "device")) return c->fail(48, "DBG_COMPRESS=device but the neighbour links are not mutual");
            // links were only modified if cycles were cut, which happens after the mutuality check passed
        }
        link.resize(2 * n);
        HIP_TRY(c, hipMemcpyAsync(link.data(), d_link.p, 2 * n * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }

    // ---- host: greedy walk in seed order (compression.rs:545-583) ----
    std::vector<uint8_t> available(n, 1);
    auto first_base = [&](uint32_t id) { return kmer_get(K128{has_hi ? key_hi[id] : 0ull, key_lo[id]}, k, 0); };
    auto last_base = [&](uint32_t id) { return (uint32_t)(key_lo[id] & 3ull); };
    auto single_dir = [&](uint32_t id, int dir) -> uint32_t { return dir ? (exts[id] >> 4) : (exts[id] & 0xfu); };   // lib.rs:719-726
    BitPusher seq;
    std::vector<uint64_t> g_start; std::vector<uint32_t> g_len, g_data; std::vector<uint8_t> g_exts;
    std::vector<std::pair<uint32_t, int>> path;
    std::deque<uint8_t> edge;
    bool spec_panic = false;
    // extend_kmer (compression.rs:450-479) over precomputed links; returns the terminal Exts nibble
    auto extend = [&](uint32_t seed, int start_dir, int* err) -> uint32_t {
        path.clear();
        uint32_t cur = seed; int dir = start_dir;
        available[seed] = 0;                                                           // :458
        for (;;) {
            uint32_t L = link[(uint64_t)dir * n + cur];
            if (L == LINK_TERM) return single_dir(cur, dir);
            uint32_t nid = (L & 0x7FFFFFFFu) >> 1; int ndir = (int)(L & 1u);
            if (!available[nid]) return single_dir(cur, dir);                          // :410-415
            if (L & LINK_PANIC) { *err = 1; return 0; }                                // :428-434
            path.push_back({nid, ndir});
            available[nid] = 0;                                                        // :467
            cur = nid; dir = ndir;
        }
    };
    for (uint64_t cidx = 0; cidx < n; cidx++) {
        uint64_t sid = seed_order ? seed_order[cidx] : cidx;
        if (sid >= n) return c->fail(44, "seed_order entry out of range");
        uint32_t seed = sorted ? (uint32_t)sid : rank[sid];
        if (!available[seed]) continue;                                                // :575
        K128 sk{has_hi ? key_hi[seed] : 0ull, key_lo[seed]};
        edge.clear();
        for (int i = 0; i < k; i++) edge.push_back((uint8_t)kmer_get(sk, k, i));       // :491-493
        uint32_t node_data = data ? data[seed] : 0u;                                   // :495
        int err = 0;
        uint32_t l_ext = extend(seed, 0, &err);                                        // :497
        if (err) return c->fail(45, "unreachable (compression.rs:434)");
        for (auto& pr : path) {                                                        // :500-511
            uint32_t b = pr.second == 0 ? first_base(pr.first) : 3u - last_base(pr.first);
            edge.push_front((uint8_t)b);
            node_data = spec_reduce(spec, node_data, data ? data[pr.first] : 0u, &spec_panic);
        }
        uint32_t left_extend = (!path.empty() && path.back().second == 1) ? exts_complement(l_ext) & 0xfu : l_ext;   // :513-517
        uint32_t r_ext = extend(seed, 1, &err);                                        // :519
        if (err) return c->fail(45, "unreachable (compression.rs:434)");
        for (auto& pr : path) {                                                        // :522-532
            uint32_t b = pr.second == 0 ? 3u - first_base(pr.first) : last_base(pr.first);
            edge.push_back((uint8_t)b);
            node_data = spec_reduce(spec, node_data, data ? data[pr.first] : 0u, &spec_panic);
        }
        uint32_t right_extend = (!path.empty() && path.back().second == 0) ? exts_complement(r_ext) & 0xfu : r_ext;  // :534-538
        if (spec_panic) return c->fail(46, "ScmapCompress::reduce on unequal data: Should not happen (compression.rs:90)");
        g_start.push_back(seq.len);                                                    // graph.rs:104-113, dna_string.rs:811-821
        for (uint8_t b : edge) seq.push(b);
        g_len.push_back((uint32_t)edge.size());
        g_exts.push_back((uint8_t)(((right_extend & 0xfu) << 4) | (left_extend & 0xfu)));   // from_single_dirs (lib.rs:591-595)
        g_data.push_back(node_data);
    }

    auto dup = [](const void* p, size_t bytes) { void* q = malloc(bytes ? bytes : 1); if (bytes) memcpy(q, p, bytes); return q; };
    out->n_nodes = g_start.size();
    out->n_seq_words = seq.words.size();
    out->seq_len_bases = seq.len;
    out->seq_words = (uint64_t*)dup(seq.words.data(), seq.words.size() * 8);
    out->start = (uint64_t*)dup(g_start.data(), g_start.size() * 8);
    out->length = (uint32_t*)dup(g_len.data(), g_len.size() * 4);
    out->exts = (uint8_t*)dup(g_exts.data(), g_exts.size());
    out->data = (uint32_t*)dup(g_data.data(), g_data.size() * 4);
    return 0;
}

// Device-resident index in (the table dbg_filter_kmers_dev returns: ascending keys), host BaseGraph out.  Seed order =
// ascending key (policy B).  data: data_dev (u32) if given, else count16_dev widened, else zeros.
extern "C" int dbg_compress_kmers_with_hash_dev(dbg_ctx* c, uint32_t k_, int stranded, int spec, uint64_t n,
                                                const uint64_t* key_hi_dev, const uint64_t* key_lo_dev, const uint8_t* exts_dev,
                                                const uint32_t* data_dev, const uint16_t* count16_dev, dbg_graph* out) {
    const int k = (int)k_;
    if (k < 1 || k > 64) return c->fail(40, "k must be in 1..=64");
    if (spec < 0 || spec > 4) return c->fail(41, "unknown CompressionSpec");
    if (n >= (1ull << 30)) return c->fail(42, "compress: at most 2^30-1 k-mers per call in this build");
    if (n && (!key_lo_dev || !exts_dev)) return c->fail(10, "null argument");
    HIP_TRY(c, hipSetDevice(c->device));
    c->t_clear();
    memset(out, 0, sizeof(*out));
    out->stranded = stranded ? 1 : 0;
    if (!n) return 0;
    const bool has_hi = k > 32;
    DBuf<uint32_t> d_link, d_flag, d_wide;
    ALLOC_OR_FAIL(c, d_link, 2 * n); ALLOC_OR_FAIL(c, d_flag, 1);
    const uint32_t* d_data = data_dev;
    if (!d_data && count16_dev) {
        ALLOC_OR_FAIL(c, d_wide, n);
        widen_u16_kernel<<<cdiv(n, 256), 256, 0, c->stream>>>(count16_dev, n, d_wide.p);
        LAUNCH_CHECK(c, "widen_u16");
        d_data = d_wide.p;
    }
    KeysDev t{has_hi ? key_hi_dev : nullptr, key_lo_dev, n};
    uint32_t fl = 0;
    HIP_TRY(c, hipMemsetAsync(d_flag.p, 0, 4, c->stream));
    // ascending keys are checked BEFORE the prefix index is built (its fill loops assume them); where the packed records are made
    // the check rides along with the packing, otherwise it is a pass of its own
    DBuf<ulonglong2> d_rec;
    DBG_TRY(attach_key_records(c, &t, k, exts_dev, &d_rec, d_flag.p));
    if (!t.rec) {
        sorted_check_kernel<<<cdiv(n, 256), 256, 0, c->stream>>>(t, d_flag.p);
        LAUNCH_CHECK(c, "sorted_check");
    }
    HIP_TRY(c, hipMemcpyAsync(&fl, d_flag.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (fl) return c->fail(49, "dbg_compress_kmers_with_hash_dev needs strictly ascending keys");
    DBuf<uint32_t> d_pidx;
    DBG_TRY(attach_prefix_index(c, &t, k, &d_pidx));
    c->t_begin("compress_links", n);
    DBuf<NodeRec> d_nrec;
    if (!c->opt("DBG_NO_NODE_RECORDS")) (void)d_nrec.alloc(c, n);
    const int lr = build_links(c, t, exts_dev, d_data, k, stranded, spec, d_link.p, d_nrec.p);
    c->t_end();
    if (lr) return lr;
    d_rec.release();
    bool done = false;
    DBG_TRY(compress_links_device(c, k, (uint32_t)n, t.hi, t.lo, exts_dev, d_data, d_link.p, nullptr, spec, stranded, out, &done, nullptr, d_nrec.p));
    if (done) return 0;
    // inconsistent Exts (non-mutual links): the literal walk needs the index on the host
    std::vector<uint64_t> h_hi(has_hi ? n : 0), h_lo(n); std::vector<uint8_t> h_ex(n); std::vector<uint32_t> h_da(d_data ? n : 0);
    if (has_hi) HIP_TRY(c, hipMemcpyAsync(h_hi.data(), key_hi_dev, n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(h_lo.data(), key_lo_dev, n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(h_ex.data(), exts_dev, n, hipMemcpyDeviceToHost, c->stream));
    if (d_data) HIP_TRY(c, hipMemcpyAsync(h_da.data(), d_data, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    d_link.release(); d_wide.release();
    return dbg_compress_kmers_with_hash(c, k_, stranded, spec, n, has_hi ? h_hi.data() : nullptr, h_lo.data(), h_ex.data(),
                                        d_data ? h_da.data() : nullptr, nullptr, out);
}

// compress_kmers_no_exts (compression.rs:619-659): the Exts come from neighbour probes into the k-mer set, then compress_kmers.
extern "C" int dbg_kmer_set_exts(dbg_ctx* c, uint32_t k_, uint64_t n, const uint64_t* key_hi, const uint64_t* key_lo, uint8_t* exts_out) {
    const int k = (int)k_;
    if (k < 1 || k > 64) return c->fail(40, "k must be in 1..=64");
    if (n >= (1ull << 32)) return c->fail(42, "at most 2^32-1 k-mers per call in this build");
    if (n && (!key_lo || !exts_out)) return c->fail(10, "null argument");
    if (n == 0) return 0;
    HIP_TRY(c, hipSetDevice(c->device));
    const bool has_hi = k > 32;
    auto key_of = [&](uint64_t i) { return K128{has_hi && key_hi ? key_hi[i] : 0ull, key_lo[i]}; };
    DBuf<uint64_t> d_hi, d_lo, s_hi, s_lo; DBuf<uint8_t> d_exts; DBuf<uint32_t> d_flag, d_pidx;
    if (has_hi) ALLOC_OR_FAIL(c, d_hi, n);
    ALLOC_OR_FAIL(c, d_lo, n); ALLOC_OR_FAIL(c, d_exts, n); ALLOC_OR_FAIL(c, d_flag, 1);
    {
        std::vector<UploadJob> jobs;
        if (has_hi) { if (key_hi) jobs.push_back({d_hi.p, key_hi, (size_t)n * 8}); else HIP_TRY(c, hipMemsetAsync(d_hi.p, 0, n * 8, c->stream)); }
        jobs.push_back({d_lo.p, key_lo, (size_t)n * 8});
        DBG_TRY(staged_upload(c, jobs));
    }
    KeysDev probe{has_hi ? d_hi.p : nullptr, d_lo.p, n}, set = probe;
    uint32_t fl = 0;
    HIP_TRY(c, hipMemsetAsync(d_flag.p, 0, 4, c->stream));
    sorted_check_kernel<<<cdiv(n, 256), 256, 0, c->stream>>>(probe, d_flag.p);
    LAUNCH_CHECK(c, "sorted_check");
    HIP_TRY(c, hipMemcpyAsync(&fl, d_flag.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (fl) {                                                       // any key order is allowed: the probes need an ascending copy
        std::vector<uint32_t> order(n);
        std::iota(order.begin(), order.end(), 0u);
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return k128_lt(key_of(a), key_of(b)); });
        for (uint64_t i = 1; i < n; i++)                            // assert_eq!(kmer_set.len(), keys.len()) (compression.rs:655)
            if (k128_eq(key_of(order[i - 1]), key_of(order[i]))) return c->fail(43, "duplicate k-mer in the k-mer set (compression.rs:655)");
        std::vector<uint64_t> h_hi(has_hi ? n : 0), h_lo(n);
        for (uint64_t i = 0; i < n; i++) { h_lo[i] = key_lo[order[i]]; if (has_hi) h_hi[i] = key_hi ? key_hi[order[i]] : 0; }
        ALLOC_OR_FAIL(c, s_lo, n);
        if (has_hi) ALLOC_OR_FAIL(c, s_hi, n);
        std::vector<UploadJob> jobs{{s_lo.p, h_lo.data(), (size_t)n * 8}};
        if (has_hi) jobs.push_back({s_hi.p, h_hi.data(), (size_t)n * 8});
        DBG_TRY(staged_upload(c, jobs));
        set = KeysDev{has_hi ? s_hi.p : nullptr, s_lo.p, n};
    }
    DBG_TRY(attach_prefix_index(c, &set, k, &d_pidx));
    c->t_begin("neighbour_exts", n);
    neighbour_exts_kernel<<<cdiv(n, 256), 256, 0, c->stream>>>(probe, set, d_exts.p, k);
    c->t_end();
    LAUNCH_CHECK(c, "neighbour_exts");
    HIP_TRY(c, hipMemcpyAsync(exts_out, d_exts.p, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int dbg_compress_kmers_no_exts(dbg_ctx* c, uint32_t k, int stranded, int spec, uint64_t n, const uint64_t* key_hi, const uint64_t* key_lo,
                                          const uint32_t* data, const uint64_t* seed_order, dbg_graph* out) {
    if (!out) return c->fail(10, "null argument");
    memset(out, 0, sizeof(*out));
    std::vector<uint8_t> exts((size_t)std::max<uint64_t>(n, 1));
    DBG_TRY(dbg_kmer_set_exts(c, k, n, key_hi, key_lo, exts.data()));
    return dbg_compress_kmers_with_hash(c, k, stranded, spec, n, key_hi, key_lo, exts.data(), data, seed_order, out);
}

// (the arrays are pinned blocks of the ctx's result pool when the graph came off the device route, plain malloc'ed memory
//  otherwise: ctx_hfree tells them apart)
extern "C" void dbg_free_graph(dbg_ctx* c, dbg_graph* g) {
    if (!g) return;
    ctx_hfree(c, g->seq_words); ctx_hfree(c, g->start); ctx_hfree(c, g->length); ctx_hfree(c, g->exts); ctx_hfree(c, g->data);
    memset(g, 0, sizeof(*g));
}

extern "C" int dbg_remove_censored_exts(dbg_ctx* c, uint32_t k_, int stranded, dbg_kmer_table* t, int sharded) {
    const int k = (int)k_;
    if (k < 1 || k > 64) return c->fail(40, "k must be in 1..=64");
    if (!t) return c->fail(10, "null argument");
    HIP_TRY(c, hipSetDevice(c->device));
    const bool has_hi = k > 32;
    uint64_t n = t->n, na = sharded ? t->n_all : 0;
    if (n == 0) return 0;
    DBuf<uint64_t> d_hi, d_lo, a_hi, a_lo; DBuf<uint8_t> d_exts;
    KeysDev valid{nullptr, nullptr, n}, all{nullptr, nullptr, na};
    uint8_t* ex = t->exts;
    if (t->on_device) {
        valid.hi = has_hi ? t->key_hi : nullptr; valid.lo = t->key_lo;
        all.hi = has_hi ? t->all_hi : nullptr; all.lo = t->all_lo;
    } else {
        ALLOC_OR_FAIL(c, d_lo, n); ALLOC_OR_FAIL(c, d_exts, n);
        HIP_TRY(c, hipMemcpyAsync(d_lo.p, t->key_lo, n * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(d_exts.p, t->exts, n, hipMemcpyHostToDevice, c->stream));
        if (has_hi) { ALLOC_OR_FAIL(c, d_hi, n); HIP_TRY(c, hipMemcpyAsync(d_hi.p, t->key_hi, n * 8, hipMemcpyHostToDevice, c->stream)); }
        if (na) {
            ALLOC_OR_FAIL(c, a_lo, na);
            HIP_TRY(c, hipMemcpyAsync(a_lo.p, t->all_lo, na * 8, hipMemcpyHostToDevice, c->stream));
            if (has_hi) { ALLOC_OR_FAIL(c, a_hi, na); HIP_TRY(c, hipMemcpyAsync(a_hi.p, t->all_hi, na * 8, hipMemcpyHostToDevice, c->stream)); }
        }
        valid.hi = has_hi ? d_hi.p : nullptr; valid.lo = d_lo.p;
        all.hi = has_hi ? a_hi.p : nullptr; all.lo = a_lo.p;
        ex = d_exts.p;
    }
    DBuf<uint32_t> pi_valid, pi_all;
    DBuf<ulonglong2> valid_recs;                                       // the link builder's packed {key, Exts} records serve these probes too
    if (n < (1ull << 32)) DBG_TRY(attach_prefix_index(c, &valid, k, &pi_valid));
    DBG_TRY(attach_key_records(c, &valid, k, ex, &valid_recs));
    if (na && na < (1ull << 32)) DBG_TRY(attach_prefix_index(c, &all, k, &pi_all));
    c->t_begin("remove_censored_exts", n);
    censor_kernel<<<cdiv(n, 256), 256, 0, c->stream>>>(valid, all, ex, k, stranded, sharded);
    c->t_end();
    LAUNCH_CHECK(c, "censor_kernel");
    if (!t->on_device) HIP_TRY(c, hipMemcpyAsync(t->exts, d_exts.p, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}
