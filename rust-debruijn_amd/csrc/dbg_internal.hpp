// Internal host-side interfaces between the translation units of libdbg_mi355x.so.
#pragma once
#include "dbg_ctx.hpp"
#include "dbg_device.hpp"
#include <cstdlib>
#include <sys/mman.h>
// Host arrays handed to the caller (released with free()).  Large ones are 2 MB-aligned and advised for transparent huge
// pages: a fresh multi-hundred-MB malloc otherwise takes ~10^5 page faults while the copy from the device fills it.
static inline void* dbg_host_alloc(size_t bytes) {
    if (bytes < (8u << 20)) return malloc(bytes ? bytes : 1);
    const size_t al = 2u << 20, sz = (bytes + al - 1) / al * al;
    void* q = aligned_alloc(al, sz);
    if (q) (void)madvise(q, sz, MADV_HUGEPAGE);
    return q;
}


// ---- hostio.hip : fast host <-> device transfers for the host-pointer entry points ----------------
#include <functional>
#include <vector>
struct UploadJob { void* dst; const void* src; size_t bytes; };
int staged_upload(dbg_ctx* c, const std::vector<UploadJob>& jobs);            // complete on return
void staged_upload_prepare(dbg_ctx* c);                                       // (see hostio.hip: uploads from a helper thread)
int staged_upload_quiet(dbg_ctx* c, const std::vector<UploadJob>& jobs, std::string* err);
void host_parallel_ranges(uint64_t n, const std::function<void(uint64_t, uint64_t, unsigned)>& fn);
unsigned host_parallel_width();
void* ctx_halloc(dbg_ctx* c, size_t bytes);                                  // result array: pinned block from the ctx pool (malloc when small)
void ctx_hfree(dbg_ctx* c, void* p);                                         // c may be null
void ctx_hrelease_all(dbg_ctx* c);
void ctx_hrelease_free(dbg_ctx* c);                                         // pooled pinned blocks only (dbg_ctx_trim)

// device view of &[(V, Exts, D1)] in PackedDnaStringSet layout
struct SeqDev {
    const uint64_t* words;
    const uint64_t* start;
    const uint32_t* length;
    const uint8_t* exts;        // may be null
    const void* data;           // may be null
    uint32_t data_width;        // 0,1,2,4
    uint64_t n;
    uint64_t n_words;           // words in `words` (loads are clamped to it where they are issued unconditionally)
};

// k-mer records in HBM: struct-of-arrays, key (hi, lo) + payload (Exts | D1 << 8)
struct RecArrays {
    uint64_t* hi;               // null when k <= 32
    uint64_t* lo;
    uint32_t* pay;
};

// ---- scan.hip : exclusive prefix sums, out has n+1 entries (out[n] = total) ----------------
int scan_exclusive_u32(dbg_ctx* ctx, const uint32_t* in, uint32_t* out, uint64_t n);
int scan_exclusive_u32_u64(dbg_ctx* ctx, const uint32_t* in, uint64_t* out, uint64_t n);
int scan_exclusive_u64(dbg_ctx* ctx, const uint64_t* in, uint64_t* out, uint64_t n);

// ---- extract.hip : iter_kmer_exts + min_rc_flip + Exts::rc fused (lib.rs:812-841, filter.rs:190-196)
// kcount[i] = len_i.saturating_sub(k-1) (filter.rs:154)
int kmer_counts(dbg_ctx* ctx, const SeqDev& s, int k, uint32_t* kcount);
int kmer_total(dbg_ctx* ctx, const SeqDev& s, int k, uint64_t* n_out);      // their sum alone
// writes records of sequence i at koff[i] .. koff[i+1]
int extract_kmers(dbg_ctx* ctx, const SeqDev& s, const uint64_t* koff, uint64_t n_kmers, int k, bool stranded,
                  RecArrays out);

// passes over key ranges: histogram of the canonical k-mers' top bytes; per-sequence counts and records of the k-mers whose
// top byte lies in [b_lo, b_hi)
int kmer_top_byte_hist(dbg_ctx* ctx, const SeqDev& s, int k, bool stranded, unsigned long long* hist_dev);
int kmer_counts_range(dbg_ctx* ctx, const SeqDev& s, int k, bool stranded, uint32_t b_lo, uint32_t b_hi, uint32_t* kcount);
int extract_kmers_range(dbg_ctx* ctx, const SeqDev& s, const uint64_t* koff, uint64_t n_kmers, int k, bool stranded, uint32_t b_lo, uint32_t b_hi,
                        RecArrays out);

// ---- radix.hip : stable LSD radix sort of records by (key, selected payload bits) -----------
// Sorts n (< 2^32) records.  key_bits = 2k significant key bits; pay_shift/pay_bits select payload
// bits that act as the least-significant sort digits (CountFilterSet needs (key, D1) order).
// a = input (clobbered), b = scratch; *result_in_b tells where the sorted records ended up.
int radix_sort_records(dbg_ctx* ctx, uint64_t n, RecArrays a, RecArrays b, int key_bits, int pay_shift,
                       int pay_bits, bool* result_in_b);

// hybrid sort of the output table: LSD passes over the top key bits + LDS finisher that also decodes the payload
// (Exts | count << 8, or Exts | colour mask << 8) into the table columns
int sort_table_hybrid(dbg_ctx* ctx, uint64_t n, RecArrays a, RecArrays b, int key_bits, bool is_set, bool allow_hybrid,
                      uint64_t* o_hi, uint64_t* o_lo, uint8_t* o_exts, uint16_t* o_count, uint32_t* o_setn, uint32_t* o_msk);

// the same for keys of at most 96 bits held as 16-byte records {key 0..31, 32..63, 64..95, payload}
int sort_table_hybrid16(dbg_ctx* ctx, uint64_t n, uint4* a, uint4* b, int key_bits, bool is_set, bool allow_hybrid,
                        uint64_t* o_hi, uint64_t* o_lo, uint8_t* o_exts, uint16_t* o_count, uint32_t* o_setn, uint32_t* o_msk);

// ---- reduce.hip : group_by key + KmerSummarizer::summarize (filter.rs:53-62, :85-100) -------
struct ReduceOut {
    uint64_t n_valid = 0, n_all = 0, n_set_val = 0;
    uint64_t *key_hi = nullptr, *key_lo = nullptr;
    uint8_t* exts = nullptr;
    uint16_t* count = nullptr;
    uint64_t* set_off = nullptr;
    uint32_t* set_val = nullptr;
    uint64_t *all_hi = nullptr, *all_lo = nullptr;
};
int reduce_sorted_records(dbg_ctx* ctx, uint64_t n, RecArrays sorted, bool has_hi, int summarizer, uint64_t min_obs,
                          bool report_all, ReduceOut* out);
// api.hip: the tables of consecutive key ranges, one after the other (the parts' arrays are released, also on failure)
int join_reduce_parts(dbg_ctx* ctx, std::vector<ReduceOut>& parts, bool is_set, bool report_all, ReduceOut* out);
// shard_generic.hip: the rank-spanning form of the generic path -- ownership by ranges of the canonical k-mer's top byte; what
// dbg_shard_filter_kmers_dev runs for shapes the super-k-mer exchange does not take (k < 16, more than 64 distinct labels, labels
// >= 65536).  X: the call's failure-agreement state (shard_comm.hpp), passed as an opaque pointer to keep that header out of here.
int shard_filter_generic(dbg_ctx* ctx, void* shard_comm, const dbg_transport* tr, const dbg_seqset* ds, const dbg_shard_params* p,
                         uint64_t total_kmers, dbg_kmer_table* out, dbg_shard_stats* stats);

// ---- fastpath.hip : super-k-mer bins + per-bin LDS hash tables --------------------------------
int filter_kmers_fast(dbg_ctx* ctx, const SeqDev& s, const dbg_filter_params* prm, uint64_t n_kmers, dbg_kmer_table* out,
                      bool* used);

// ---- densepath.hip : directly addressed tables for 4 <= k <= 15 ------------------------------------
int filter_kmers_dense(dbg_ctx* ctx, const SeqDev& s, const dbg_filter_params* prm, uint64_t n_kmers, dbg_kmer_table* out,
                       bool* used);

// ---- classes.hip : label lists of a CountFilterSet table -> dense class ids ---------------------
int label_classes_device(dbg_ctx* ctx, uint64_t n, const uint64_t* set_off_dev, const uint32_t* set_val_dev, uint64_t n_set_val,
                         uint32_t* class_dev, dbg_label_classes* classes);

// ---- synth.hip ----------------------------------------------------------------------------
int synth_reads_dev(dbg_ctx* ctx, const dbg_synth_params* p, uint64_t* words, uint64_t* start, uint32_t* length,
                    uint8_t* data);

// ---- unitig.hip : chains of mutually linked elements -> BaseGraph (k-mers, or graph nodes for compress_graph) ----
struct UnitigNodes {            // device arrays describing graph nodes as elements
    const uint32_t* weight;     // k-mers per node (length - k + 1)
    const uint64_t* words;      // packed node sequences
    const uint64_t* start;
    const uint32_t* length;
    const uint8_t* avail;       // 0 = censored node (compression.rs:297-307); null = all available
};

// One k-mer of the index with everything the chain walks of unitig.hip read per step, in half a cache line (the separate
// key / link / data arrays cost three lines per step).  Written by link_kernel next to the plain link array.
struct __attribute__((aligned(32))) NodeRec {
    uint64_t lo, hi;
    uint32_t link[2];      // as link[side * n + i]
    uint32_t data;
    uint32_t exts;
};

// A BaseGraph (graph.rs:43-50) resident in HBM: what the rank-spanning second stage passes from step to step and sends
// between ranks (graph.hip).  words carries at least two zero words of slack behind n_words (k-mer fetches need no clamp).
struct GraphDev {
    DBuf<uint64_t> words, start;
    DBuf<uint32_t> length, data;
    DBuf<uint8_t> exts;
    DBuf<uint32_t> seed;            // (first level of unitig.hip's segment route only) the seed k-mer of every node
    uint64_t n_nodes = 0, n_words = 0, n_bases = 0;
    int stranded = 0;
    bool filled = false;
};

int graph_dev_join_segments(struct dbg_ctx* c, int k, int stranded, int spec, GraphDev* in, GraphDev* out);   // graph.hip

// ---- launch helpers -----------------------------------------------------------------------
#define LAUNCH_CHECK(ctx, name)                                                                 \
    do {                                                                                        \
        hipError_t _e = hipGetLastError();                                                      \
        if (_e != hipSuccess) {                                                                 \
            char _b[256];                                                                       \
            snprintf(_b, sizeof(_b), "kernel launch %s failed: %s", name, hipGetErrorString(_e)); \
            return (ctx)->fail(102, _b);                                                        \
        }                                                                                       \
    } while (0)

// order-preserving dictionary of the distinct u32 labels of a call (labeldict.hip): labels of 2^24 and more run as their ranks
struct LabelDict {
    DBuf<unsigned long long> bm;        // presence bitmap over the 2^32 values
    DBuf<uint32_t> base;                // rank of the first value of every block of 1024 values
    DBuf<uint32_t> dict;                // rank -> label, ascending
    uint32_t n = 0;
};
int label_dict_make(dbg_ctx* c, const uint32_t* labels, uint64_t n, LabelDict* d);
int label_dict_rank(dbg_ctx* c, const LabelDict& d, const uint32_t* labels, uint64_t n, DBuf<uint32_t>* rank_out);
int label_dict_build(dbg_ctx* c, const uint32_t* labels, uint64_t n, DBuf<uint32_t>* rank_out, DBuf<uint32_t>* dict_out, uint32_t* n_distinct);
int label_dict_translate(dbg_ctx* c, uint32_t* set_val, uint64_t n, const uint32_t* dict, uint32_t n_dict);

