/* =============================================================================
 * dbg_mi355x.h -- C ABI of the MI355X-native k-mer hot path of the `debruijn`
 * crate (10XGenomics/rust-debruijn): k-mer extraction -> MSP shard ->
 * count/filter -> path compression.
 *
 * The reference has no FFI: its boundary is the pair of generic Rust functions
 * `filter_kmers` (src/filter.rs:139-148) and `compress_kmers_with_hash`
 * (src/compression.rs:588-594) plus the sibling `msp_sequence`
 * (src/msp.rs:279-288).  Traits and closures cannot cross a C ABI, so this
 * header monomorphises over a closed set:
 *   K   : VarIntKmer<u64|u128, K> with run-time k, 1 <= k <= 64 (filter_kmers: 4 <= k, see dbg_filter_params); every key
 *         crosses the ABI as a (hi, lo) u64 pair, right-aligned in 2k bits,
 *         base 0 most significant (src/kmer.rs:429-437).
 *   V   : any Vmer, flattened to PackedDnaStringSet layout
 *         (src/dna_string.rs:762-767): one packed u64 stream (32 bases/word,
 *         base i at bits [63-2(i%32), 62-2(i%32)] of word i/32,
 *         src/dna_string.rs:383-399) + start[] (BASE offsets) + length[].
 *   D1  : none | u8 | u16 | u32.  dbg_filter_kmers[_dev] take every u32 value: a call whose largest label is 2^24 or more runs on
 *         an order-preserving dictionary of its distinct labels (at most 2^24 - 1 of them; round 6); the rank-spanning
 *         dbg_shard_filter_kmers_dev builds that dictionary job-wide (every rank then passes u32 labels).
 *   S   : CountFilter (src/filter.rs:40-63) | CountFilterSet (:68-101).
 *   spec: SimpleCompress with saturating_add / (a+b)%65535 / max / wrapping add,
 *         or ScmapCompress (src/compression.rs:40-98).
 *
 * Conventions: every function returns 0 on success; non-zero means "the
 * reference would have panicked / invalid argument / device error" and
 * dbg_last_error(ctx) holds the message.  The library allocates outputs; the
 * caller releases them with the matching dbg_free_* call.  One ctx per host
 * thread; no hidden globals (the reference functions are re-entrant).
 * Functions suffixed _dev take and return DEVICE pointers (HBM-resident, the
 * configuration the benchmark times); the unsuffixed ones take HOST pointers
 * exactly like the Rust call sites and stage through PCIe.
 * ========================================================================== */
#ifndef DBG_MI355X_H
#define DBG_MI355X_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dbg_ctx dbg_ctx;

/* ---- context ------------------------------------------------------------ */
int         dbg_ctx_create(int device, dbg_ctx** out);
void        dbg_ctx_destroy(dbg_ctx* ctx);
const char* dbg_last_error(dbg_ctx* ctx);        /* ctx may be NULL: last create error */
const char* dbg_version(void);
/* use an externally owned hipStream_t (e.g. torch's current stream); NULL = ctx's own stream */
int         dbg_ctx_set_stream(dbg_ctx* ctx, void* hip_stream);
/* scratch budget in bytes for intermediate k-mer records (0 = 60% of free HBM) */
int         dbg_ctx_set_scratch_budget(dbg_ctx* ctx, uint64_t bytes);
/* Diagnostic knobs (DBG_PATH, DBG_COMPRESS, ...).  The library reads the environment exactly once,
 * in dbg_ctx_create, into the ctx; this call changes one knob of one ctx afterwards (value NULL = unset).  Knobs select
 * among device routes that all produce the same result; unknown names are rejected. */
int         dbg_ctx_set_option(dbg_ctx* ctx, const char* name, const char* value);
/* The ctx keeps the device scratch blocks (and pinned host result blocks) of finished calls for reuse -- tens of GB after a
 * full-size call.  dbg_ctx_trim returns every block that is not in a caller's hands to the driver (after draining the
 * stream), so that another allocator sharing the GPU (torch, a second ctx) can have the memory; *freed_bytes (may be NULL)
 * receives the device bytes released.  The library also trims by itself when one of its own allocations fails. */
int         dbg_ctx_trim(dbg_ctx* ctx, uint64_t* freed_bytes);

/* ---- input: &[(V, Exts, D1)]  (src/filter.rs:140) flattened -------------- */
typedef struct {
    const uint64_t* words;      /* packed bases; n_words >= ceil(max(start+length)/32) */
    uint64_t        n_words;
    const uint64_t* start;      /* [n_seqs] base offset of each sequence (not word aligned) */
    const uint32_t* length;     /* [n_seqs] bases */
    const uint8_t*  exts;       /* [n_seqs] Exts byte of each sequence; NULL = Exts::empty() */
    const void*     data;       /* [n_seqs] D1 values, width data_width bytes; NULL = unit */
    uint32_t        data_width; /* 0, 1, 2 or 4 */
    uint64_t        n_seqs;
} dbg_seqset;

/* The _dev entry points take a dbg_seqset of DEVICE pointers.  A host that does not drive the HIP runtime itself (a Rust shim
 * without HIP bindings) gets one with dbg_seqset_to_device: the checked, staged upload of the host-pointer calls, the device
 * buffers left in the caller's hands; dbg_seqset_free_device returns them to the ctx. */
int  dbg_seqset_to_device(dbg_ctx* ctx, const dbg_seqset* host_seqs, dbg_seqset* dev_out);
void dbg_seqset_free_device(dbg_ctx* ctx, dbg_seqset* dev);

/* ---- filter_kmers (src/filter.rs:139-231) -------------------------------- */
enum { DBG_COUNT_FILTER = 0, DBG_COUNT_FILTER_SET = 1 };

typedef struct {
    uint32_t k;                 /* K::k(); 4 <= k <= 64: filter_kmers' bucket() reads base 3 of every k-mer (filter.rs:18-23), and
                                   Kmer::get(3) of a shorter k-mer underflows `k - 1 - pos` (kmer.rs:254-257, :515-518) -- a panic in
                                   the reference's debug builds, reproduced here as an error */
    int32_t  stranded;          /* filter.rs:142 */
    int32_t  summarizer;        /* DBG_COUNT_FILTER(min) | DBG_COUNT_FILTER_SET(min).  Label sets of any D1 values are served:
                                   up to 64 distinct labels (< 65536) as colour masks in the counting kernel; any larger alphabet as
                                   label lists (one pass: every observation's label appended to its k-mer's segment, segments sorted
                                   and de-duplicated -- about a third of the colour-mask rate, whatever the alphabet); the sort-based
                                   generic path for k < 16 or when the label buffer (4 bytes per k-mer instance) does not fit; labels of
                                   2^24 and more are ranked among the call's distinct labels first and translated back at the end */
    uint64_t min_kmer_obs;      /* filter.rs:41,69 */
    int32_t  report_all_kmers;  /* filter.rs:143 */
    uint64_t memory_size;       /* filter.rs:144, GB; 0 is rejected (reference divides by zero).  Otherwise IGNORED: in the
                                   reference it only sets the number of bucket-range passes (filter.rs:156-168), never the
                                   result; here the device decides its own passes from free HBM */
    uint32_t compact_sets;      /* dbg_filter_kmers (host tables), opt-in bit flags (round 5) -- the table crosses PCIe, so bytes are time:
                                   1: CountFilterSet: return the CSR in the narrowest element types that hold it -- set_off as
                                      uint32_t[n + 1] when n_set_val < 2^32, set_val in the width of D1 (data_width of the input:
                                      uint8_t / uint16_t / uint32_t) -- and say so in dbg_kmer_table.set_off_width / set_val_width
                                      (BASELINE configs[1]: 7.4 GB as u64 / u32, 2.9 GB compact);
                                   2: k <= 32: key_hi comes back NULL instead of n zeros (8 bytes per k-mer: 4 GB at configs[1]'s
                                      size; IntKmer<u64> callers never read it).
                                   0 = the plain form.  Device tables (the _dev calls) are always plain */
} dbg_filter_params;

/* The vectors the reference hands to BoomHashMap2::new (filter.rs:227-230), i.e. ascending
 * k-mer order, as struct-of-arrays.  Host or device pointers depending on the call. */
typedef struct {
    uint64_t  n;                /* valid k-mers */
    uint64_t* key_hi;           /* [n] zero when k <= 32 */
    uint64_t* key_lo;           /* [n] */
    uint8_t*  exts;             /* [n] accumulated Exts */
    uint16_t* count;            /* [n] CountFilter DS = u16 (saturating); NULL for CountFilterSet */
    uint64_t* set_off;          /* [n+1] CountFilterSet DS = Vec<D1> as CSR; NULL for CountFilter */
    uint32_t* set_val;          /* [set_off[n]] sorted, de-duplicated D1 values */
    uint64_t  n_set_val;
    uint64_t  n_all;            /* all_kmers (second tuple element), 0 unless report_all_kmers */
    uint64_t* all_hi;
    uint64_t* all_lo;
    uint64_t  n_kmer_instances; /* input_kmers (filter.rs:152-155) */
    uint32_t  n_passes;         /* passes over the input the device needed (1 unless the input was streamed in several
                                   prefix ranges); unrelated to memory_size */
    int32_t   on_device;        /* 1 when the arrays are device pointers */
    uint32_t  set_off_width;    /* bytes per set_off element: 0 or 8 = uint64_t (the declared type); 4 = the array is uint32_t[n + 1] */
    uint32_t  set_val_width;    /* bytes per set_val element: 0 or 4 = uint32_t; 1 / 2 = the array is uint8_t / uint16_t[n_set_val]
                                   (only tables asked for with dbg_filter_params.compact_sets carry narrow arrays) */
} dbg_kmer_table;

int  dbg_filter_kmers(dbg_ctx* ctx, const dbg_seqset* host_seqs, const dbg_filter_params* p, dbg_kmer_table* out_host);
int  dbg_filter_kmers_dev(dbg_ctx* ctx, const dbg_seqset* dev_seqs, const dbg_filter_params* p, dbg_kmer_table* out_dev);
/* Host tables hold arrays from the ctx's pool of pinned host blocks (kept and reused: pinning is slow): release them with the
 * ctx that produced them, before dbg_ctx_destroy. */
void dbg_free_table(dbg_ctx* ctx, dbg_kmer_table* t);
/* copy a device table to freshly allocated host arrays */
int  dbg_table_to_host(dbg_ctx* ctx, const dbg_kmer_table* dev, dbg_kmer_table* out_host);

/* remove_censored_exts_sharded / remove_censored_exts (src/filter.rs:238-306): rewrites
 * table->exts in place; all_* may be NULL for the non-sharded form. */
int  dbg_remove_censored_exts(dbg_ctx* ctx, uint32_t k, int stranded, dbg_kmer_table* table, int sharded);

/* ---- msp_sequence (src/msp.rs:279-324), batched over many sequences ------- */
typedef struct {
    uint32_t k;
    uint32_t p;                 /* P::k(), 1 <= p <= 16, p <= k */
    const uint32_t* permutation;/* [4^p] or NULL = identity (msp.rs:298-303) */
    int32_t  rc;                /* msp.rs:283 */
    uint32_t lmer_words;        /* V = Lmer<[u64; lmer_words]> (vmer.rs:32-47); 0 = do not emit packed pieces.
                                   The reference asserts V::max_len() = (64*lmer_words-8)/2 >= 2k-p (msp.rs:292) */
} dbg_msp_params;

/* Vec<(u32, Exts, V)> of every input sequence, concatenated in input order; piece_off[i]..piece_off[i+1]
 * are the pieces of sequence i (empty when its length < k, msp.rs:294-296). */
typedef struct {
    uint64_t  n_pieces;
    uint64_t* piece_off;        /* [n_seqs+1] */
    uint32_t* bucket;           /* [n_pieces] MspIntervalP::bucket() = canonical minimizer (msp.rs:115-117) */
    uint8_t*  exts;             /* [n_pieces] Exts::from_slice_bounds (lib.rs:645-660) */
    uint32_t* start;            /* [n_pieces] MspIntervalP.start */
    uint16_t* len;              /* [n_pieces] MspIntervalP.len */
    uint32_t* minimizer_pos;    /* [n_pieces] MspIntervalP.minimizer_pos */
    uint64_t* lmer;             /* [n_pieces * lmer_words] V::from_slice(piece) or NULL */
    int32_t   on_device;
} dbg_msp_pieces;

int  dbg_msp_sequence(dbg_ctx* ctx, const dbg_seqset* host_seqs, const dbg_msp_params* p, dbg_msp_pieces* out_host);
int  dbg_msp_sequence_dev(dbg_ctx* ctx, const dbg_seqset* dev_seqs, const dbg_msp_params* p, dbg_msp_pieces* out_dev);
void dbg_free_pieces(dbg_ctx* ctx, dbg_msp_pieces* pc);

/* ---- compress_kmers_with_hash (src/compression.rs:588-594) ---------------- */
enum {
    DBG_SPEC_SIMPLE_SAT_ADD_U16 = 0,   /* SimpleCompress(|a,b| a.saturating_add(*b))  (test.rs:383) */
    DBG_SPEC_SIMPLE_ADD_MOD_U16 = 1,   /* SimpleCompress(|a,b| (a+b) % 65535)          (test.rs:247) */
    DBG_SPEC_SIMPLE_MAX_U16     = 2,   /* SimpleCompress(|a,b| max(a,*b))              (test.rs:469) */
    DBG_SPEC_SCMAP_EQ           = 3,   /* ScmapCompress                                (compression.rs:68-98) */
    DBG_SPEC_SIMPLE_WRAP_ADD_U16= 4    /* SimpleCompress(|a,b| a+b), release build     (test.rs:265) */
};

/* BaseGraph<K,D> (src/graph.rs:43-50): sequences = PackedDnaStringSet (dna_string.rs:762-767) */
typedef struct {
    uint64_t  n_nodes;
    uint64_t* seq_words;        /* sequences.sequence.storage */
    uint64_t  n_seq_words;
    uint64_t  seq_len_bases;    /* sequences.sequence.len */
    uint64_t* start;            /* [n_nodes] sequences.start */
    uint32_t* length;           /* [n_nodes] sequences.length */
    uint8_t*  exts;             /* [n_nodes] */
    uint32_t* data;             /* [n_nodes] D */
    int32_t   stranded;
} dbg_graph;

/* index: the BoomHashMap2<K,Exts,D> contents as SoA (any key order).  data: u32 per k-mer or NULL
 * (treated as 0).  seed_order: [n] permutation giving the id visited at each step of the reference's
 * `for kmer_counter in 0..n` loop (compression.rs:574), i.e. the MPHF slot order supplied by the Rust
 * shim; NULL = ascending key order.  index arrays are HOST pointers; the graph is returned on the host. */
int  dbg_compress_kmers_with_hash(dbg_ctx* ctx, uint32_t k, int stranded, int spec,
                                  uint64_t n, const uint64_t* key_hi, const uint64_t* key_lo,
                                  const uint8_t* exts, const uint32_t* data,
                                  const uint64_t* seed_order, dbg_graph* out);
/* Same, index resident on the device (e.g. the table of dbg_filter_kmers_dev; keys strictly ascending, which is
 * checked).  data = data_dev (u32) if non-NULL, else count16_dev (u16, widened) if non-NULL, else zeros.  Seed order =
 * ascending key.  The graph is returned on the host. */
int  dbg_compress_kmers_with_hash_dev(dbg_ctx* ctx, uint32_t k, int stranded, int spec, uint64_t n,
                                      const uint64_t* key_hi_dev, const uint64_t* key_lo_dev, const uint8_t* exts_dev,
                                      const uint32_t* data_dev, const uint16_t* count16_dev, dbg_graph* out);
/* compress_kmers_no_exts (src/compression.rs:619-659): a bare k-mer set (any key order; the keys as the caller holds them) ->
 * Exts from eight neighbour probes each (the neighbour is canonicalised with min_rc whether or not the graph is stranded,
 * as the reference does) -> compress_kmers.  A duplicate key is the reference's assert_eq!(kmer_set.len(), keys.len()).
 * dbg_kmer_set_exts returns the Exts alone (exts_out: [n] host bytes). */
int  dbg_kmer_set_exts(dbg_ctx* ctx, uint32_t k, uint64_t n, const uint64_t* key_hi, const uint64_t* key_lo, uint8_t* exts_out);
int  dbg_compress_kmers_no_exts(dbg_ctx* ctx, uint32_t k, int stranded, int spec, uint64_t n,
                                const uint64_t* key_hi, const uint64_t* key_lo, const uint32_t* data,
                                const uint64_t* seed_order, dbg_graph* out);
void dbg_free_graph(dbg_ctx* ctx, dbg_graph* g);

/* ---- CountFilterSet payload as ScmapCompress data (BASELINE config 5) ------------------------
 * After filter_kmers(CountFilterSet) the reference's index is BoomHashMap2<K, Exts, Vec<D1>> (src/filter.rs:68-101) and
 * ScmapCompress<Vec<D1>>::join_test compares the label lists of two k-mers for equality; reduce returns the common list
 * (src/compression.rs:68-98).  Only equality is used, so a dense class id per distinct list carries the same information.
 * dbg_label_classes_dev writes class_dev[i] (u32) for every k-mer of a CSR label table -- equal lists <=> equal ids; ids
 * are ranks in ascending order of the colour bitmask when all labels are < 64 (of a verified 64-bit hash otherwise) -- and
 * returns the class table (id -> sorted label list) on the host, so that a node's Vec<D1> is classes[node data]. */
typedef struct {
    uint64_t  n_classes;
    uint64_t* set_off;          /* [n_classes + 1] */
    uint32_t* set_val;          /* [set_off[n_classes]] */
    uint64_t  n_set_val;
} dbg_label_classes;
int  dbg_label_classes_dev(dbg_ctx* ctx, uint64_t n, const uint64_t* set_off_dev, const uint32_t* set_val_dev, uint64_t n_set_val,
                           uint32_t* class_dev, dbg_label_classes* classes /* may be NULL */);
void dbg_free_label_classes(dbg_label_classes* classes);
/* compress_kmers_with_hash (src/compression.rs:588-594) on the table dbg_filter_kmers_dev left in HBM: D = the count
 * column (CountFilter) or the class id of the label list (CountFilterSet; *classes receives the class table, may be NULL).
 * Seed order = ascending key.  The graph is returned on the host. */
int  dbg_compress_table_dev(dbg_ctx* ctx, uint32_t k, int stranded, int spec, const dbg_kmer_table* table_dev, dbg_graph* out,
                            dbg_label_classes* classes);

/* ---- sharded second stage: BaseGraph::combine (src/graph.rs:71-100) and compress_graph
 *      (src/compression.rs:338-349, which implies finish (graph.rs:116-142) + fix_exts (:337-377)) ---- */
int  dbg_graph_combine(dbg_ctx* ctx, const dbg_graph* graphs, uint32_t n_graphs, dbg_graph* out);
int  dbg_compress_graph(dbg_ctx* ctx, uint32_t k, int stranded, int spec, const dbg_graph* old_graph,
                        const uint64_t* censor_nodes, uint64_t n_censor, dbg_graph* out);

/* ---- far side of the path: edges and GFA text of a graph ------------------------------------
 * dbg_graph_edges = Node::l_edges / r_edges (src/graph.rs:1041-1049 -> find_edges :223-241 -> find_link
 * :252-291) for every node, resolved on the device against sorted node-end indices (finish, :116-142).
 * target[8*i + b] (b < 4: left edge through base b; b >= 4: right edge through base b-4) = target node or
 * 0xFFFFFFFF; info bit0 = incoming side of the target is Right, bit1 = target flipped.  Host arrays. */
typedef struct {
    uint64_t  n_nodes;
    uint32_t* target;           /* [8 * n_nodes] */
    uint8_t*  info;             /* [8 * n_nodes] */
} dbg_edges;
int  dbg_graph_edges(dbg_ctx* ctx, uint32_t k, const dbg_graph* g, dbg_edges* out);
void dbg_free_edges(dbg_edges* e);
/* DebruijnGraph::write_gfa / to_gfa (src/graph.rs:537-611): byte-identical text for the same BaseGraph */
int  dbg_graph_to_gfa(dbg_ctx* ctx, uint32_t k, const dbg_graph* g, char** text, uint64_t* len);
int  dbg_graph_write_gfa(dbg_ctx* ctx, uint32_t k, const dbg_graph* g, const char* path);
void dbg_free_text(char* text);

/* ---- serde forms of BaseGraph<K, D> (derives: src/graph.rs:43-50, src/dna_string.rs:72-76,762-767, src/lib.rs:577-580) ----
 * DBG_SERDE_JSON = serde_json (a dependency of the crate): {"sequences":{"sequence":{"storage":[..],"len":n},"start":[..],
 * "length":[..]},"exts":[{"val":v},..],"data":[..],"stranded":b,"phantom":null}.  DBG_SERDE_BINCODE = bincode 1.x default
 * options: fields in that order, little-endian fixed-width integers, usize as u64, Vec = u64 length + elements.
 * data_width = size_of::<D>() (0 = unit, 1, 2, 4); JSON ignores it when reading.  Host-side formatting: ctx may be NULL
 * (then dbg_serde_last_error() holds the message).  Release with dbg_free_bytes / dbg_free_graph. */
enum { DBG_SERDE_JSON = 0, DBG_SERDE_BINCODE = 1 };
int  dbg_graph_serialize(dbg_ctx* ctx, const dbg_graph* g, int format, uint32_t data_width, uint8_t** bytes, uint64_t* len);
int  dbg_graph_deserialize(dbg_ctx* ctx, const uint8_t* bytes, uint64_t len, int format, uint32_t data_width, dbg_graph* out);
void dbg_free_bytes(uint8_t* bytes);
const char* dbg_serde_last_error(void);

/* ---- near side of the path: ASCII <-> 2-bit packing -----------------------------------------
 * dbg_pack_acgt = DnaString::from_acgt_bytes (src/dna_string.rs:222-250; AVX2 helpers convert_bases /
 * pack_32_bases, src/bitops_avx2.rs:9-132): n ASCII characters -> ceil(n/32) words, first base in the top
 * two bits, anything outside [aAcCgGtT] -> A (base_to_bits, src/lib.rs:65-73).  *n_invalid (host, may be
 * NULL) = how many characters were not ACGT.  A batch of reads concatenated without separators packs
 * straight into PackedDnaStringSet layout with start[i] = byte offset of read i (dna_string.rs:811-821).
 * dbg_unpack_acgt = DnaString::to_ascii_vec (bits_to_ascii, src/lib.rs:53-61) of bases
 * [first_base, first_base + n).  _dev: device pointers in and out. */
int  dbg_pack_acgt(dbg_ctx* ctx, const uint8_t* ascii, uint64_t n, uint64_t* words, uint64_t* n_invalid);
int  dbg_pack_acgt_dev(dbg_ctx* ctx, const uint8_t* ascii_dev, uint64_t n, uint64_t* words_dev, uint64_t* n_invalid);
/* DnaString::from_acgt_bytes_hashn (src/dna_string.rs:255-278) for a batch of reads: read i =
 * ascii[seq_off[i], seq_off[i+1]) (seq_off[0] = 0, n = seq_off[n_seqs] characters), its name =
 * names[name_off[i], name_off[i+1]).  [aAcCgGtT] pack as above; any other character at position pos of read i
 * becomes DefaultHasher::new() fed read_name.hash() then pos.hash(), finish() % 4 -- Rust std's SipHash-1-3 with
 * the zero key over (len(name) as u64 LE, name bytes, pos as u64 LE).  Output layout as dbg_pack_acgt.
 * *n_replaced (host, may be NULL) = number of hashed characters. */
int  dbg_pack_acgt_hashn(dbg_ctx* ctx, const uint8_t* ascii, const uint64_t* seq_off, uint64_t n_seqs,
                         const uint8_t* names, const uint64_t* name_off, uint64_t* words, uint64_t* n_replaced);
int  dbg_pack_acgt_hashn_dev(dbg_ctx* ctx, const uint8_t* ascii_dev, uint64_t n, const uint64_t* seq_off_dev,
                             uint64_t n_seqs, const uint8_t* names_dev, const uint64_t* name_off_dev,
                             uint64_t* words_dev, uint64_t* n_replaced);
int  dbg_unpack_acgt(dbg_ctx* ctx, const uint64_t* words, uint64_t first_base, uint64_t n, uint8_t* ascii);
int  dbg_unpack_acgt_dev(dbg_ctx* ctx, const uint64_t* words_dev, uint64_t first_base, uint64_t n, uint8_t* ascii_dev);

/* ---- sharded counting for multi-GPU runs --------------------------------------------------
 * The reference's own scale-out design is "MSP shard -> independent per-shard filter_kmers"
 * (src/msp.rs:279-324, src/filter.rs:121-124, src/test.rs:433-456).  Here every rank scans its own
 * reads into minimizer bins of ONE global bin space, the caller exchanges the bin-ordered slabs (RCCL
 * all-to-all over xGMI; rank r owns bins [r*n_bins/W, (r+1)*n_bins/W)), and each rank counts the bins
 * it owns.  All pointers are DEVICE pointers. */
typedef struct {
    uint32_t k;
    int32_t  stranded;
    int32_t  summarizer;        /* DBG_COUNT_FILTER | DBG_COUNT_FILTER_SET (labels: see max_label / labels below) */
    uint64_t min_kmer_obs;
    uint64_t total_kmers;       /* k-mer instances over ALL ranks: fixes the bin count */
    uint32_t n_bins;            /* in: 0 = derive from total_kmers, otherwise the bin count to use (every rank must pass the
                                   same value); out of dbg_shard_plan_make: bins */
    uint32_t rec_words;         /* out: u64 words per super-k-mer record */
    uint32_t bin_group;         /* out: ownership boundaries must be multiples of this many bins */
    uint32_t max_label;         /* CountFilterSet: largest D1 label over ALL ranks (dbg_seqset_max_label_dev + a max-reduction);
                                   labels must be < 64 here -- 24..63 select the two-word colour layout on every rank.
                                   0 = "below 24" (what callers of the earlier ABI passed) */
    uint32_t merge_dups;        /* dbg_shard_scan_dev: merge this rank's identical super-k-mer records before the exchange (a
                                   merged record carries its multiplicity; the counting side reads it either way, so ranks may
                                   choose independently).  Fewer bytes on the wire: a rank with 1/W of 18x reads still holds
                                   18/W copies of most records */
    uint32_t n_labels;          /* CountFilterSet with labels >= 64: the sparse alphabet of ALL ranks (at most 64 distinct labels,
                                   each < 65536): labels[0..n_labels) ascending; colour i of the counting kernel stands for
                                   labels[i].  0 = labels are their own colours (max_label < 64).  Every rank passes the same
                                   list (dbg_seqset_label_bitmap_dev + an OR/max-reduction); a rank holding a label outside it
                                   is refused */
    uint32_t labels[64];
} dbg_shard_plan;
/* largest D1 label of a device-resident sequence set (0 when it carries no data) */
int  dbg_seqset_max_label_dev(dbg_ctx* ctx, const dbg_seqset* dev_seqs, uint32_t* max_label_out);
/* which D1 labels occur: bitmap_out[2049] (HOST memory): bit (v & 31) of word v >> 5 for every label v < 65536 present;
 * word 2048 != 0 when a label >= 65536 was seen (such sets take dbg_filter_kmers' generic path) */
int  dbg_seqset_label_bitmap_dev(dbg_ctx* ctx, const dbg_seqset* dev_seqs, uint32_t* bitmap_out);

int  dbg_count_kmer_instances_dev(dbg_ctx* ctx, const dbg_seqset* dev_seqs, uint32_t k, uint64_t* n_out);
int  dbg_shard_plan_make(dbg_ctx* ctx, dbg_shard_plan* plan);
/* stage 1a: scan local reads; writes bin_off_dev[n_bins+1] (record offsets per bin) and *n_recs */
int  dbg_shard_scan_dev(dbg_ctx* ctx, const dbg_seqset* dev_seqs, const dbg_shard_plan* plan, uint64_t* n_recs,
                        uint64_t* bin_off_dev);
/* stage 1b: write the records in bin order into recs_out_dev[n_recs * rec_words] */
int  dbg_shard_scatter_dev(dbg_ctx* ctx, const uint64_t* bin_off_dev, uint64_t* recs_out_dev);
/* stage 2: count n_bins_local bins whose records are n_src bin-ordered segments of recs_dev:
 * segment s of local bin b = records [seg_off[s*(n_bins_local+1)+b], seg_off[s*(n_bins_local+1)+b+1]) */
int  dbg_shard_count_dev(dbg_ctx* ctx, const dbg_shard_plan* plan, const uint64_t* recs_dev, const uint64_t* seg_off_dev,
                         uint32_t n_src, uint32_t n_bins_local, uint64_t n_kmers_hint, dbg_kmer_table* out_dev);
/* The same in pieces, so that the caller can count one range of its bins while the records of the next range are still
 * in flight: begin; then for each contiguous range of owned bins a record buffer + its [n_src][n_bins_chunk + 1] segment
 * table (n_kmers_units is only used for the timing record); finish sorts everything counted since begin. */
int  dbg_shard_count_begin(dbg_ctx* ctx, const dbg_shard_plan* plan, uint64_t n_kmers_hint);
int  dbg_shard_count_bins_dev(dbg_ctx* ctx, const uint64_t* recs_dev, const uint64_t* seg_off_dev, uint32_t n_src,
                              uint32_t n_bins_chunk, uint64_t n_kmers_units);
int  dbg_shard_count_finish(dbg_ctx* ctx, dbg_kmer_table* out);

/* ---- the rank-spanning flow behind the C ABI (round 4) ---------------------------------------------
 * BASELINE north_star: "Shard MSP buckets across the 8 GPUs of one node with RCCL all-to-all over xGMI".  The reference's
 * sharded flow is composed by its caller (src/test.rs:433-470: msp_sequence -> per-shard filter_kmers -> per-shard
 * compress_kmers_with_hash -> BaseGraph::combine -> compress_graph); here the same flow across GPUs runs INSIDE the library,
 * one process per GPU, and what moves bytes between ranks is this small table of operations.
 * dbg_transport_rccl_create fills it with RCCL calls on an ncclComm_t the host created (ncclCommInitRank in a Rust host;
 * dbg_rccl_comm_create below for hosts that do not link RCCL themselves).  A host may also supply its own table (MPI, a test
 * harness that stages through host memory so that N ranks can share one GPU).
 *
 * Contract of every operation: all data buffers are DEVICE pointers; count/offset arrays are host arrays that are only read
 * during the call.  The operation is ordered on `hip_stream` (a hipStream_t): it may return before the data has moved, and it
 * is complete once the work enqueued on that stream up to the call's return is complete (a synchronous implementation drains
 * the stream, moves the data, and returns).  Every rank calls the same operations in the same order.  Return 0 on success;
 * otherwise the calling entry point fails and dbg_last_error names the operation.
 *
 * Failures (round 5).  The reference is one process, where a failure is a panic that unwinds everything (src/filter.rs:167,
 * src/graph.rs:87-91); across processes the two entry points below give the equivalent: ALL RANKS FAIL TOGETHER.  A rank-local
 * failure (out of memory, a bad label, a kernel error) never makes a rank leave a collective phase early: every phase ends in a
 * one-word status all-reduce, and when any rank failed every rank returns the same non-zero code before the phase's data moves
 * (dbg_last_error names the phase and the failing rank).  A failure of the transport itself cannot be agreed on: the rank aborts
 * the communicator (abort below), so that its peers' operations fail or their bounded waits end, and returns.  After an agreed
 * failure ctx and transport remain usable; after an abort the transport is dead. */
typedef struct dbg_transport {
    void*    self;
    int32_t  rank, world;
    /* element-wise reduction over all ranks of n u64 values, in place; op: 0 = sum, 1 = max */
    int (*all_reduce_u64)(void* self, uint64_t* buf_dev, uint64_t n, int32_t op, void* hip_stream);
    /* every rank contributes `bytes` bytes; recv_dev receives world * bytes in rank order */
    int (*all_gather)(void* self, const void* send_dev, void* recv_dev, uint64_t bytes, void* hip_stream);
    /* variable all-to-all: send_bytes[d] bytes at send_dev + send_off[d] go to rank d; recv_bytes[s] bytes from rank s
     * arrive at recv_dev + recv_off[s] (RCCL: one ncclGroupStart/End of ncclSend/ncclRecv pairs) */
    int (*all_to_allv)(void* self, const void* send_dev, const uint64_t* send_off, const uint64_t* send_bytes,
                       void* recv_dev, const uint64_t* recv_off, const uint64_t* recv_bytes, void* hip_stream);
    /* point to point; sends and receives between one pair of ranks match in call order */
    int (*send)(void* self, const void* buf_dev, uint64_t bytes, int32_t peer, void* hip_stream);
    int (*recv)(void* self, void* buf_dev, uint64_t bytes, int32_t peer, void* hip_stream);
    /* Failure handling (round 5; both may be NULL).  poll: 0 while the communicator is healthy, non-zero once it has failed
     * (RCCL: ncclCommGetAsyncError) -- the library calls it while it waits for communication, next to a deadline
     * (ctx option DBG_COMM_TIMEOUT_S, default 300).  abort: tear the communication down so that peers blocked in it fail
     * instead of hanging (RCCL: ncclCommAbort); every later operation of the table fails.  The library calls it when an
     * operation returns non-zero, poll reports a failure, or a wait passes the deadline. */
    int  (*poll)(void* self);
    void (*abort)(void* self);
} dbg_transport;

/* RCCL transport on an existing communicator.  nccl_comm = the host's ncclComm_t (one rank per GPU, rank/world as given to
 * ncclCommInitRank).  librccl_path: the librccl the communicator came from (NULL = the copy already loaded into the process,
 * else "librccl.so.1"); the symbols are resolved at run time, so libdbg_mi355x.so itself does not link RCCL and single-GPU
 * users never load it.  Messages are cut so that no single ncclSend exceeds 1 GiB.  err (may be NULL) receives a message. */
int  dbg_transport_rccl_create(void* nccl_comm, int32_t rank, int32_t world, const char* librccl_path, dbg_transport** out,
                               char* err, uint64_t err_len);
void dbg_transport_destroy(dbg_transport* t);       /* tables made by this library only; the ncclComm_t stays the caller's */
/* 1 once a table made by this library has been aborted or has broken (after a transport failure or a timed-out wait inside
 * dbg_shard_filter_kmers_dev / dbg_shard_compress_dev).  For the RCCL table that means ncclCommAbort has freed the ncclComm_t:
 * the host must not destroy it again, and needs a new communicator to go on. */
int  dbg_transport_aborted(const dbg_transport* t);
/* In-process transport: the ranks are THREADS of one process -- one per GPU, or several sharing a GPU -- each with its own ctx and
 * its own table (out[0..world)).  Device buffers of all ranks live in one address space, so the variable all-to-all is
 * device-to-device copies (peer copies over xGMI between different GPUs) issued by the receiving rank between two barriers of the
 * threads; no RCCL.  Synchronous (an operation drains the caller's stream, moves the data, returns); a rank that does not arrive
 * within 300 s (environment DBG_INPROC_TIMEOUT_S at creation) breaks the group, and so does any failing operation and abort
 * (every pending and later operation fails).  Release each table with dbg_transport_destroy. */
int  dbg_transport_inprocess_create(int32_t world, dbg_transport** out /* [world] */);
/* For hosts that do not bind RCCL themselves: ncclGetUniqueId on one rank (id_out: 128 bytes, to be handed to the other ranks
 * by whatever bootstrap the host has), then ncclCommInitRank on every rank with its GPU current. */
int  dbg_rccl_unique_id(const char* librccl_path, uint8_t* id_out_128, char* err, uint64_t err_len);
int  dbg_rccl_comm_create(const char* librccl_path, const uint8_t* id_128, int32_t world, int32_t rank, int32_t device,
                          void** nccl_comm_out, char* err, uint64_t err_len);
int  dbg_rccl_comm_destroy(const char* librccl_path, void* nccl_comm);

/* Ownership of the global bin space: rank r owns bins [bounds[r], bounds[r+1]), boundaries multiples of bin_group.
 * group_records[n_bins / bin_group] = records of every bin group summed over ALL ranks -> contiguous ranges of nearly equal
 * record count (greedy cut of the cumulative histogram: low-complexity minimizers do not pile up on one owner);
 * NULL = equal numbers of bins.  Pure host arithmetic, the same on every rank (no GPU needed). */
int  dbg_shard_owner_bounds(const uint64_t* group_records, uint32_t n_bins, uint32_t bin_group, uint32_t world,
                            uint32_t* bounds_out /* [world + 1] */);
/* Exchange rounds: every rank's owned range is cut into n_rounds ranges (cuts_out[d * stride + c], stride = the value of
 * *n_rounds_io at entry + 1; relative to bounds[d], multiples of bin_group); round c moves range c of every destination while
 * range c - 1 is counted.  n_rounds is clamped to what the smallest owner can be cut into; the value used is returned in
 * *n_rounds_io (the stride stays what it was). */
int  dbg_shard_round_cuts(const uint32_t* bounds, uint32_t world, uint32_t bin_group, uint32_t* n_rounds_io,
                          uint32_t* cuts_out /* [world * (*n_rounds_io + 1)], sized for the value passed in */);

typedef struct {
    uint32_t k;
    int32_t  stranded;
    int32_t  summarizer;        /* DBG_COUNT_FILTER | DBG_COUNT_FILTER_SET (any labels < 2^24: up to 64 distinct ones < 65536 over all
                                   ranks as colour masks, larger alphabets as label lists -- both on the super-k-mer route) */
    uint64_t min_kmer_obs;
    uint32_t n_rounds;          /* exchange rounds; 0 = 4 (8 from 4 ranks on), more when a round's receive buffer would pass 8 GiB */
    int32_t  merge_dups;        /* sender-side duplicate merge: 1 on, 0 off, -1 = the library decides (on at 2 ranks, where one link
                                   carries everything; otherwise by a vote of all ranks -- max-reduced, so every rank uses the same
                                   setting -- from the exposed exchange and wire time each ctx measured in its last call,
                                   re-evaluated every call) */
    int32_t  balance;           /* 1 = ownership from the all-reduced record histogram of the scan (default), 0 = equal bin ranges */
    int32_t  force_exchange;    /* world == 1: run the collective route anyway (a functional check of the transport) */
} dbg_shard_params;

typedef struct {
    uint64_t total_kmers;       /* k-mer instances over all ranks */
    uint64_t local_kmers;       /* ... of this rank's reads */
    uint64_t records_scanned;   /* super-k-mer records this rank produced (after the sender-side merge) */
    uint64_t records_owned;     /* records of the bins this rank owns, from all ranks */
    uint64_t bytes_sent;        /* record bytes that left this rank */
    uint32_t n_bins, owned_lo, owned_hi;
    uint32_t n_rounds;
    int32_t  merge_dups;        /* what was used */
    int32_t  balanced;
    double   exposed_ms;        /* time the counting kernels waited for exchange rounds (HIP events on the ctx's stream) */
    double   exposed_ms_round[64];
    double   setup_ms;          /* host time from entry to the first round on the wire (plan, scan, histogram, layout) */
} dbg_shard_stats;

/* filter_kmers over the reads of ALL ranks (src/filter.rs:139-231 applied to the union; the reference's scale-out is
 * msp_sequence shards + per-shard filter_kmers, src/test.rs:433-456).  Every rank passes its own device-resident reads and
 * receives the ascending table of the valid k-mers of the bins it owns; the tables of the ranks are disjoint and their union
 * is the table dbg_filter_kmers_dev returns for the concatenated reads.  Collective: every rank calls it with the same
 * parameters.  Scan -> ownership -> slab compaction in (round, destination, bin) order -> pipelined all-to-all rounds on a
 * communication stream ordered against the ctx's stream with events (no host synchronisation per round) -> per-bin counting
 * -> one order-restoring sort.  Label alphabets beyond 64 colours travel as label lists: the records carry their read's label in an
 * extra word, the owners append, sort and de-duplicate per k-mer (round 5; DBG_LABEL_LISTS=0: label groups up to 1024 labels, the
 * key-range route beyond).  Shapes that exchange cannot carry -- k < 16 -- take the key-range route instead (round 5): ownership by
 * ranges of the canonical k-mer's top byte from the all-reduced
 * byte histogram, k-mer records exchanged per round, sort + segmented reduce on the owner (the rank-spanning form of the generic
 * path; SURVEY.md section 8(e) "non-MSP variant"), so the call covers every shape dbg_filter_kmers_dev covers.  stats may be NULL. */
int  dbg_shard_filter_kmers_dev(dbg_ctx* ctx, const dbg_transport* tr, const dbg_seqset* dev_seqs, const dbg_shard_params* p,
                                dbg_kmer_table* out_dev, dbg_shard_stats* stats);

/* The rank-spanning end of the flow (src/test.rs:459-470): every rank runs compress_kmers_with_hash on the device-resident table
 * of the bins it owns, then the shard graphs are merged with BaseGraph::combine (src/graph.rs:71-100) + compress_graph
 * (src/compression.rs:291-349).  Graphs stay in HBM between the steps and travel as device buffers.
 * reduce = DBG_REDUCE_GATHER: all shard graphs go to `root`, which combines them in rank order and runs one compress_graph --
 *   the reference's flow literally (the result equals the oracle's node for node).
 * reduce = DBG_REDUCE_TREE: pairwise combine + compress_graph up a binary tree (ranks r and r + 2^l at level l), so that the
 *   early merges run in parallel and the root sees the last pair only; the same unitigs, in a different node order / strand
 *   (compare with tools/compare_gfa.py's canonical form).  Inner levels hold a part of the shards: their compress_graph
 *   treats an extension whose k-mer is in no node yet as the end of the path FOR NOW and skips both fix_exts
 *   (compression.rs:309, :331 would strip the Exts that point into shards still to come); the root's last merge is the
 *   reference's compress_graph proper.
 * CountFilterSet tables: D = class id of the label list, made global first (the ranks' class tables are all-gathered and
 * renumbered in sorted order of the lists; *classes receives the global table on every rank).
 * final_out: host BaseGraph on `root` (n_nodes = 0 elsewhere); local_out (may be NULL): this rank's own shard graph. */
enum { DBG_REDUCE_GATHER = 0, DBG_REDUCE_TREE = 1 };
int  dbg_shard_compress_dev(dbg_ctx* ctx, const dbg_transport* tr, uint32_t k, int stranded, int spec, int second_spec,
                            const dbg_kmer_table* table_dev, int32_t reduce, int32_t root, dbg_graph* final_out,
                            dbg_graph* local_out, dbg_label_classes* classes);

/* ---- synthetic reads: splitmix64, deterministic ----- */
typedef struct {
    uint64_t n_reads;
    uint32_t read_len;          /* L */
    uint64_t genome_len;        /* G; 0 = n_reads*read_len/30 */
    uint64_t genome_seed;       /* 0xDB60001 */
    uint64_t read_seed;         /* 0xDB60002 */
    double   error_rate;        /* substitution probability per base */
    int32_t  stranded;          /* 0: reverse-complement each read with prob 1/2 */
    uint32_t n_colours;         /* D1 = read_index % n_colours as u8; 0 = no data */
    uint64_t first_read;        /* index of the first read to generate (shards a stream across ranks) */
} dbg_synth_params;

/* n_words needed for n_reads reads back-to-back + 2 words of tail padding */
uint64_t dbg_synth_words(const dbg_synth_params* p);
/* fill caller-allocated device buffers: words[dbg_synth_words], start[n], length[n], data[n] (u8, may be NULL) */
int  dbg_synth_reads_dev(dbg_ctx* ctx, const dbg_synth_params* p, uint64_t* words, uint64_t* start,
                         uint32_t* length, uint8_t* data);
/* same stream generated on the host (no GPU needed) */
int  dbg_synth_reads_host(const dbg_synth_params* p, uint64_t* words, uint64_t* start,
                          uint32_t* length, uint8_t* data);

/* ---- timing hook for bench.py: per-kernel HIP-event time of the last call --- */
typedef struct {
    char     name[48];
    double   ms;                /* summed over launches in the last API call */
    uint32_t launches;
    uint64_t units;             /* k-mer instances (or records) the launches processed */
} dbg_kernel_time;
int  dbg_ctx_enable_timing(dbg_ctx* ctx, int on);
int  dbg_ctx_get_timings(dbg_ctx* ctx, dbg_kernel_time* out, uint32_t cap, uint32_t* n_out);

/* ---- allocation account of a ctx + the slab the scan writes (round 6) ---------------------------------------------
 * What a cold call pays before its first kernel (driver allocations, pinned host blocks), and the slab tournament: a block of
 * tens of GB takes the scan's random record writes 10 % faster or slower depending on where the driver placed it (DESIGN.md
 * section 4.1), so the first DBG_SLAB_TRIALS (default 4) fast-path calls of a shape each scan into a fresh block, the scan is
 * timed, and the fastest block stays.  Nothing degrades silently: slab_backing says what the last scan ran on.
 * The caller sets struct_size = sizeof(dbg_ctx_stats); the library fills at most that many bytes (fields are only ever
 * appended). */
#define DBG_SLAB_NONE         0   /* no slab taken yet (dense / generic path, or no call so far) */
#define DBG_SLAB_PLAIN        1   /* a plain device block of 4 GB or more: takes part in the tournament */
#define DBG_SLAB_PLAIN_SMALL  2   /* below 4 GB: a pool block (placement kinds only show on blocks of many GB) */
#define DBG_SLAB_RECORD_ORDER 5   /* no memory for slabs: every record takes the read-order buffer and the scatter pass */
typedef struct {
    uint32_t struct_size;
    uint32_t slab_backing;          /* DBG_SLAB_* of the last fast-path scan */
    uint64_t slab_bytes;
    uint32_t slab_rec_words;        /* 64-bit words per record of that scan */
    uint32_t slab_pooled;           /* 1: the block is in the ctx pool right now (dbg_ctx_probe_slab can run on it) */
    uint64_t pooled_bytes;          /* device bytes the ctx holds (live + kept for reuse) */
    uint64_t pooled_high_water;
    uint64_t n_hipmalloc, n_fresh_blocks, n_pool_hits, n_trims, n_oom_retries, n_raw_free, n_pinned_alloc;
    double   s_hipmalloc;           /* host seconds inside hipMalloc */
    double   s_free;                /* ... giving blocks back (hipFree) */
    double   s_pinned_alloc;        /* ... hipHostMalloc of staging / result blocks */
    char     slab_note[96];
    /* slab tournament: slab_trial_ms[i] = scan time of candidate i (milliseconds, first scan into the block) */
    uint32_t slab_trials_done;
    uint32_t slab_candidates_pooled;    /* measured candidates still held (1 once the tournament has settled) */
    float    slab_trial_ms[8];
} dbg_ctx_stats;
int  dbg_ctx_get_stats(dbg_ctx* ctx, dbg_ctx_stats* out);
/* Random-write probe of the slab block the last scan used, while it sits in the pool between calls: n_writes (0 = 2^27)
 * record-sized writes at random record slots, best of three launches, milliseconds in *ms_out (*writes_out: the writes of one
 * launch).  On MI355X 2^27 24-byte writes into 26 GB take 5.4 ms in a well-placed block and up to 7.4 ms in a badly placed one.
 * Returns non-zero when there is no pooled slab to probe. */
int  dbg_ctx_probe_slab(dbg_ctx* ctx, uint64_t n_writes, float* ms_out, uint64_t* writes_out);
/* Pays the cold costs of a first call ahead of time: allocates a device block of slab_bytes (0 = none) and pinned_bytes of pinned
 * host result blocks (0 = none) into the ctx pools.  A later call of matching size then starts like a warm one. */
int  dbg_ctx_warm(dbg_ctx* ctx, uint64_t slab_bytes, uint64_t pinned_bytes);
/* ABI revision of this header: structs only grow at their end from one revision to the next, and a host may check
 * dbg_abi_version() >= the revision it was compiled against before it passes the newer fields (round 6 = 6). */
#define DBG_ABI_VERSION 6
uint32_t dbg_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DBG_MI355X_H */
