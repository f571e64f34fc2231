// extern "C" boundary: context management and filter_kmers (src/filter.rs:139-231).
#include "dbg_internal.hpp"
#include <thread>
#include <chrono>
#include <algorithm>
#include <cstdlib>
#include <atomic>

static thread_local std::string g_create_err;

extern "C" const char* dbg_version(void) { return "dbg_mi355x 0.1 (gfx950)"; }

extern "C" int dbg_ctx_create(int device, dbg_ctx** out) {
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        (void)hipGetLastError();
        g_create_err = "no HIP device available (the MI355X path has no CPU fallback)";
        return 1;
    }
    if (device < 0 || device >= n) { g_create_err = "device index out of range"; return 2; }
    if (hipSetDevice(device) != hipSuccess) { g_create_err = "hipSetDevice failed"; return 3; }
    dbg_ctx* c = new dbg_ctx();
    c->device = device;
    { int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cu > 0) c->n_cu = cu; else (void)hipGetLastError(); }
    for (const char* name : DBG_OPTION_NAMES)               // the only place the library reads the environment
        if (const char* v = getenv(name)) c->opts[name] = v;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c; g_create_err = "hipStreamCreate failed"; return 4;
    }
    *out = c;
    return 0;
}

extern "C" void dbg_ctx_destroy(dbg_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    c->shard_scan.reset(); c->shard_count.reset();          // they hold pool blocks of this ctx
    c->stager.reset();
    ctx_hrelease_all(c);
    (void)hipStreamSynchronize(c->stream);
    c->t_clear();
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    for (auto& kv : c->free_blocks) c->raw_free(kv.second);
    for (auto& kv : c->live_blocks) c->raw_free(kv.first);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" const char* dbg_last_error(dbg_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

extern "C" int dbg_ctx_set_stream(dbg_ctx* c, void* s) {
    (void)hipStreamSynchronize(c->stream);
    if (s) {
        if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
        c->stream = (hipStream_t)s; c->own_stream = false;
    } else if (!c->own_stream) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    return 0;
}

extern "C" int dbg_ctx_set_option(dbg_ctx* c, const char* name, const char* value) {
    if (!name) return c->fail(10, "null option name");
    bool known = false;
    for (const char* n : DBG_OPTION_NAMES) known = known || !strcmp(n, name);
    if (!known) return c->fail(18, std::string("unknown option ") + name);
    if (value) c->opts[name] = value; else c->opts.erase(name);
    return 0;
}

extern "C" int dbg_ctx_trim(dbg_ctx* c, uint64_t* freed_bytes) {
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t before = c->pooled_bytes;
    c->trim();                                       // drains the stream, frees the pooled device blocks
    ctx_hrelease_free(c);                            // pinned result blocks not in a caller's hands
    if (freed_bytes) *freed_bytes = before - c->pooled_bytes;
    return 0;
}

extern "C" uint32_t dbg_abi_version(void) { return DBG_ABI_VERSION; }

extern "C" int dbg_ctx_get_stats(dbg_ctx* c, dbg_ctx_stats* out) {
    if (!out || out->struct_size < 8) return c->fail(10, "dbg_ctx_get_stats: struct_size not set");
    dbg_ctx_stats st;
    memset(&st, 0, sizeof(st));
    st.slab_backing = c->slab_backing; st.slab_bytes = c->slab_bytes; st.slab_rec_words = c->slab_rec_words;
    if (c->slab_ptr) for (auto& kv : c->free_blocks) if (kv.second == c->slab_ptr) st.slab_pooled = 1;
    st.pooled_bytes = c->pooled_bytes; st.pooled_high_water = c->pooled_high_water;
    st.n_hipmalloc = c->n_hipmalloc; st.n_fresh_blocks = c->n_fresh_blocks; st.n_pool_hits = c->n_pool_hits; st.n_trims = c->n_trims;
    st.n_oom_retries = c->n_oom_retries; st.n_raw_free = c->n_raw_free; st.n_pinned_alloc = c->n_pinned_alloc;
    st.s_hipmalloc = c->s_hipmalloc; st.s_free = c->s_free; st.s_pinned_alloc = c->s_pinned_alloc;
    memcpy(st.slab_note, c->slab_note, sizeof(st.slab_note));
    st.slab_trials_done = c->slab_trials_done; st.slab_candidates_pooled = (uint32_t)c->slab_cands.size();
    memcpy(st.slab_trial_ms, c->slab_trial_ms, sizeof(st.slab_trial_ms));
    const uint32_t n = std::min<uint32_t>(out->struct_size, (uint32_t)sizeof(st));
    st.struct_size = n;
    memcpy(out, &st, n);
    return 0;
}

// record-sized writes at random record slots of a block (tools/micro/slab_probe5.hip's pattern: what the scan does to its slab)
template <int RW>
__global__ void __launch_bounds__(256) slab_write_probe_kernel(uint64_t* __restrict__ slab, uint64_t n_rec, uint32_t per_thread) {
    auto mix = [](uint32_t x) { x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16; return x; };
    uint32_t h = mix(blockIdx.x * 1024u + threadIdx.x);
    for (uint32_t i = 0; i < per_thread; i++) {
        h = mix(h + i);
        const uint32_t h2 = mix(h ^ 0x9E3779B9u);
        const uint64_t r = (uint64_t)(((unsigned __int128)(((uint64_t)h << 32) | h2) * n_rec) >> 64);
        uint64_t* o = slab + r * RW;
#pragma unroll
        for (int q = 0; q < RW; q++) o[q] = h + q;
    }
}

extern "C" int dbg_ctx_probe_slab(dbg_ctx* c, uint64_t n_writes, float* ms_out, uint64_t* writes_out) {
    HIP_TRY(c, hipSetDevice(c->device));
    bool pooled = false;
    size_t bytes = 0;
    if (c->slab_ptr) for (auto& kv : c->free_blocks) if (kv.second == c->slab_ptr) { pooled = true; bytes = std::min(kv.first, c->slab_bytes); }
    if (!pooled || bytes < 4096) return c->fail(19, "dbg_ctx_probe_slab: no pooled slab (no fast-path call yet, the block is in use, or the pool was trimmed)");
    const int rw = c->slab_rec_words >= 2 && c->slab_rec_words <= 5 ? (int)c->slab_rec_words : 3;
    if (!n_writes) n_writes = 1ull << 27;
    const uint32_t per_thread = 16;
    const uint32_t blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_writes / (256ull * per_thread), 1u << 20));
    const uint64_t n_rec = bytes / (8ull * rw);
    hipEvent_t a = c->get_event(), b = c->get_event();
    float best = 1e30f;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(a, c->stream);
        uint64_t* sp = (uint64_t*)c->slab_ptr;
        if (rw == 2) slab_write_probe_kernel<2><<<blocks, 256, 0, c->stream>>>(sp, n_rec, per_thread);
        else if (rw == 3) slab_write_probe_kernel<3><<<blocks, 256, 0, c->stream>>>(sp, n_rec, per_thread);
        else if (rw == 4) slab_write_probe_kernel<4><<<blocks, 256, 0, c->stream>>>(sp, n_rec, per_thread);
        else slab_write_probe_kernel<5><<<blocks, 256, 0, c->stream>>>(sp, n_rec, per_thread);
        (void)hipEventRecord(b, c->stream);
        HIP_TRY(c, hipEventSynchronize(b));
        float ms = 0;
        HIP_TRY(c, hipEventElapsedTime(&ms, a, b));
        best = std::min(best, ms);
    }
    c->event_pool.push_back(a); c->event_pool.push_back(b);
    if (ms_out) *ms_out = best;
    if (writes_out) *writes_out = (uint64_t)blocks * 256ull * per_thread;
    return 0;
}

extern "C" int dbg_ctx_warm(dbg_ctx* c, uint64_t slab_bytes, uint64_t pinned_bytes) {
    HIP_TRY(c, hipSetDevice(c->device));
    if (slab_bytes) {
        void* p = c->dalloc(slab_bytes);
        if (!p) return c->fail(101, "dbg_ctx_warm: device allocation failed");
        c->dfree(p);
    }
    if (pinned_bytes) {
        // result arrays come in several blocks (keys, Exts, counts / label sets): four quarters serve the usual split
        void* q[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int i = 0; i < 4; i++) q[i] = ctx_halloc(c, (size_t)(pinned_bytes / 4));
        for (int i = 0; i < 4; i++) ctx_hfree(c, q[i]);
    }
    return 0;
}

extern "C" int dbg_ctx_set_scratch_budget(dbg_ctx* c, uint64_t bytes) { c->scratch_budget = bytes; return 0; }

extern "C" int dbg_ctx_enable_timing(dbg_ctx* c, int on) { c->timing = on != 0; c->t_clear(); return 0; }

extern "C" int dbg_ctx_get_timings(dbg_ctx* c, dbg_kernel_time* out, uint32_t cap, uint32_t* n_out) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::vector<dbg_kernel_time> agg;
    for (auto& r : c->trecs) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) { (void)hipGetLastError(); continue; }
        size_t i = 0;
        for (; i < agg.size(); i++) if (!strcmp(agg[i].name, r.name)) break;
        if (i == agg.size()) {
            dbg_kernel_time t; memset(&t, 0, sizeof(t));
            strncpy(t.name, r.name, sizeof(t.name) - 1);
            agg.push_back(t);
        }
        agg[i].ms += ms; agg[i].launches += 1; agg[i].units += r.units;
    }
    uint32_t n = (uint32_t)std::min<size_t>(agg.size(), cap);
    for (uint32_t i = 0; i < n; i++) out[i] = agg[i];
    *n_out = (uint32_t)agg.size();
    return 0;
}

// ------------------------------------------------------------------------------------------
static int validate_filter(dbg_ctx* c, const dbg_seqset* s, const dbg_filter_params* p) {
    if (!s || !p) return c->fail(10, "null argument");
    if (p->k < 4 || p->k > 64) return c->fail(11, "k must be in 4..=64 (filter.rs:18-23 reads the first 4 bases)");
    if (p->memory_size == 0) return c->fail(12, "attempt to divide by zero: memory_size = 0 (filter.rs:158)");
    if (p->summarizer != DBG_COUNT_FILTER && p->summarizer != DBG_COUNT_FILTER_SET) return c->fail(13, "unknown summarizer");
    if (s->data && !(s->data_width == 1 || s->data_width == 2 || s->data_width == 4)) return c->fail(14, "data_width must be 1, 2 or 4");
    if (s->n_seqs && (!s->words || !s->start || !s->length)) return c->fail(15, "null sequence arrays");
    return 0;
}

static void table_from_reduce(const ReduceOut& r, uint64_t n_inst, dbg_kmer_table* t) {
    memset(t, 0, sizeof(*t));
    t->n = r.n_valid; t->key_hi = r.key_hi; t->key_lo = r.key_lo; t->exts = r.exts; t->count = r.count;
    t->set_off = r.set_off; t->set_val = r.set_val; t->n_set_val = r.n_set_val;
    t->n_all = r.n_all; t->all_hi = r.all_hi; t->all_lo = r.all_lo;
    t->n_kmer_instances = n_inst; t->n_passes = 1; t->on_device = 1;
}

__global__ void shift_u64_kernel(const uint64_t* __restrict__ in, uint64_t n, uint64_t add, uint64_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + add;
}

// The tables of consecutive key ranges (passes of the generic path, rounds of its rank-spanning form), one after the other: set_off
// entries of later parts are shifted by the labels before them.  The parts' arrays are released (also on failure).
int join_reduce_parts(dbg_ctx* c, std::vector<ReduceOut>& parts, bool is_set, bool report_all, ReduceOut* out) {
    auto free_parts = [&]() {
        for (auto& r : parts) for (void* q : {(void*)r.key_hi, (void*)r.key_lo, (void*)r.exts, (void*)r.count, (void*)r.set_off, (void*)r.set_val, (void*)r.all_hi, (void*)r.all_lo}) c->dfree(q);
        parts.clear();
    };
    ReduceOut r;
    if (parts.size() == 1) { *out = parts[0]; parts.clear(); return 0; }
    for (auto& q : parts) { r.n_valid += q.n_valid; r.n_all += q.n_all; r.n_set_val += q.n_set_val; }
    DBuf<uint64_t> o_hi, o_lo, o_set_off, all_hi, all_lo;
    DBuf<uint8_t> o_exts; DBuf<uint16_t> o_count; DBuf<uint32_t> o_set_val;
    bool ok = o_hi.alloc(c, std::max<uint64_t>(r.n_valid, 1)) && o_lo.alloc(c, std::max<uint64_t>(r.n_valid, 1)) && o_exts.alloc(c, std::max<uint64_t>(r.n_valid, 1));
    if (is_set) ok = ok && o_set_off.alloc(c, r.n_valid + 1) && o_set_val.alloc(c, std::max<uint64_t>(r.n_set_val, 1));
    else ok = ok && o_count.alloc(c, std::max<uint64_t>(r.n_valid, 1));
    if (report_all) ok = ok && all_hi.alloc(c, std::max<uint64_t>(r.n_all, 1)) && all_lo.alloc(c, std::max<uint64_t>(r.n_all, 1));
    if (!ok) { free_parts(); return c->fail(101, "device allocation failed in the generic path"); }
    uint64_t ov = 0, oa = 0, os = 0;
    hipError_t e = hipSuccess;
    auto cp = [&](void* d, const void* sp, size_t bytes) { if (bytes && e == hipSuccess) e = hipMemcpyAsync(d, sp, bytes, hipMemcpyDeviceToDevice, c->stream); };
    for (auto& q : parts) {
        cp(o_hi.p + ov, q.key_hi, q.n_valid * 8); cp(o_lo.p + ov, q.key_lo, q.n_valid * 8); cp(o_exts.p + ov, q.exts, q.n_valid);
        if (is_set) {
            cp(o_set_val.p + os, q.set_val, q.n_set_val * 4);
            if (e == hipSuccess && q.n_valid) { shift_u64_kernel<<<cdiv(q.n_valid, 256), 256, 0, c->stream>>>(q.set_off, q.n_valid, os, o_set_off.p + ov); e = hipGetLastError(); }
        } else cp(o_count.p + ov, q.count, q.n_valid * 2);
        if (report_all) { cp(all_hi.p + oa, q.all_hi, q.n_all * 8); cp(all_lo.p + oa, q.all_lo, q.n_all * 8); }
        ov += q.n_valid; oa += q.n_all; os += q.n_set_val;
    }
    if (is_set && e == hipSuccess) e = hipMemcpyAsync(o_set_off.p + r.n_valid, &r.n_set_val, 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    free_parts();
    if (e != hipSuccess) { (void)hipGetLastError(); return c->fail(100, std::string("HIP error while joining the passes: ") + hipGetErrorString(e)); }
    r.key_hi = o_hi.take(); r.key_lo = o_lo.take(); r.exts = o_exts.take(); r.count = o_count.take();
    r.set_off = o_set_off.take(); r.set_val = o_set_val.take(); r.all_hi = all_hi.take(); r.all_lo = all_lo.take();
    *out = r;
    return 0;
}

int seq_max_label(dbg_ctx* c, const SeqDev& s, uint32_t* out);                                       // fastpath.hip
static int filter_kmers_dev_impl(dbg_ctx* c, const dbg_seqset* ds, const dbg_filter_params* p, dbg_kmer_table* out);

extern "C" int dbg_filter_kmers_dev(dbg_ctx* c, const dbg_seqset* ds, const dbg_filter_params* p, dbg_kmer_table* out) {
    DBG_TRY(validate_filter(c, ds, p));
    HIP_TRY(c, hipSetDevice(c->device));
    c->t_clear();
    memset(out, 0, sizeof(*out));
    // Full-width D1 (CountFilterSet<D: Ord> takes any u32, filter.rs:68-101): the device routes hold labels in 24 bits, so a call whose
    // largest label does not fit runs on the labels' ranks among the call's distinct labels and gets its set_val translated back
    // (labeldict.hip).  Smaller labels take the routes as they are.
    if (p->summarizer == DBG_COUNT_FILTER_SET && ds->data && ds->data_width == 4 && ds->n_seqs) {
        SeqDev s{ds->words, ds->start, ds->length, ds->exts, ds->data, 4u, ds->n_seqs, ds->n_words};
        uint32_t mx = 0;
        DBG_TRY(seq_max_label(c, s, &mx));
        if (mx >= (1u << 24)) {
            DBuf<uint32_t> rank, dict;
            uint32_t nd = 0;
            DBG_TRY(label_dict_build(c, (const uint32_t*)ds->data, ds->n_seqs, &rank, &dict, &nd));
            dbg_seqset ranked = *ds;
            ranked.data = rank.p;
            auto kept = std::move(c->trecs);                             // (the impl clears the timing records: keep the dictionary's)
            c->trecs.clear();
            const int r = filter_kmers_dev_impl(c, &ranked, p, out);
            c->trecs.insert(c->trecs.begin(), kept.begin(), kept.end());
            if (r) return r;
            const int t = label_dict_translate(c, out->set_val, out->n_set_val, dict.p, nd);
            if (t) { dbg_free_table(c, out); return t; }
            return 0;
        }
    }
    return filter_kmers_dev_impl(c, ds, p, out);
}

static int filter_kmers_dev_impl(dbg_ctx* c, const dbg_seqset* ds, const dbg_filter_params* p, dbg_kmer_table* out) {
    c->t_clear();
    memset(out, 0, sizeof(*out));
    const int k = (int)p->k;
    const bool has_hi = k > 32;
    const bool is_set = p->summarizer == DBG_COUNT_FILTER_SET;
    SeqDev s{ds->words, ds->start, ds->length, ds->exts, ds->data, ds->data ? ds->data_width : 0u, ds->n_seqs, ds->n_words};

    uint64_t n_kmers = 0;
    DBG_TRY(kmer_total(c, s, k, &n_kmers));                  // the fast path needs the total only
    if (k < 16) DBG_TRY(c->wait_all_reads());                // (host-boundary calls: only the fast path's scan takes the reads as they arrive)
    {   // short k-mers: directly addressed tables (densepath.hip); DBG_PATH=dense insists on it
        const char* force = c->opt("DBG_PATH");
        const bool want_dense = !force || !strcmp(force, "auto") || !strcmp(force, "dense");
        if (want_dense) {
            bool used = false;
            DBG_TRY(filter_kmers_dense(c, s, p, n_kmers, out, &used));
            if (used) return 0;
            if (force && !strcmp(force, "dense")) return c->fail(21, "DBG_PATH=dense but the dense path does not support this call shape (4 <= k <= 15, labels < 64)");
        }
    }
    {   // fast path: super-k-mer bins + LDS hash tables (fastpath.hip); DBG_PATH=generic|fast|auto overrides
        const char* force = c->opt("DBG_PATH");
        bool want_fast = !(force && (!strcmp(force, "generic") || !strcmp(force, "dense")));
        if (want_fast) {
            bool used = false;
            DBG_TRY(filter_kmers_fast(c, s, p, n_kmers, out, &used));
            if (used) return 0;
            if (force && !strcmp(force, "fast") && n_kmers) return c->fail(21, "DBG_PATH=fast but the fast path does not support this call shape");
        }
    }
    DBG_TRY(c->wait_all_reads());
    DBuf<uint32_t> kcount;                                   // per-sequence counts and offsets: the generic path writes one record per k-mer
    DBuf<uint64_t> koff;
    ALLOC_OR_FAIL(c, kcount, std::max<uint64_t>(s.n, 1));
    ALLOC_OR_FAIL(c, koff, s.n + 1);
    DBG_TRY(kmer_counts(c, s, k, kcount.p));
    DBG_TRY(scan_exclusive_u32_u64(c, kcount.p, koff.p, s.n));
    // ---- generic path: extract every k-mer instance -> global radix sort -> segmented reduce, in as many passes over key
    //      ranges as the records need (one pass = at most `pass_max` records: 2^32 - 1, less when device memory is short; the
    //      reference does the same with its bucket ranges, filter.rs:156-168).  Ranges ascend, so the passes' tables simply
    //      follow one another. ----
    uint64_t pass_max = (1ull << 32) - 1;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const uint64_t budget = c->scratch_budget ? c->scratch_budget : (uint64_t)((free_b + c->pooled_bytes) * 0.6);
            const uint64_t per_rec = 2ull * ((has_hi ? 8 : 0) + 8 + 4) + 24;       // two record buffers + sort/reduce scratch
            pass_max = std::min<uint64_t>(pass_max, std::max<uint64_t>(budget / per_rec, 1u << 20));
        } else (void)hipGetLastError();
        if (const char* e = c->opt("DBG_GENERIC_PASS_MAX")) pass_max = std::max<uint64_t>(1, strtoull(e, nullptr, 10));   // tests: force several passes
    }
    std::vector<std::pair<uint32_t, uint32_t>> ranges;                             // top-byte ranges [lo, hi)
    std::vector<uint64_t> range_n;
    if (n_kmers <= pass_max) { ranges.push_back({0u, 256u}); range_n.push_back(n_kmers); }
    else {
        DBuf<unsigned long long> d_hist;
        ALLOC_OR_FAIL(c, d_hist, 256);
        DBG_TRY(kmer_top_byte_hist(c, s, k, p->stranded != 0, d_hist.p));
        unsigned long long hist[256];
        HIP_TRY(c, hipMemcpyAsync(hist, d_hist.p, sizeof(hist), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        uint32_t lo = 0; uint64_t acc = 0;
        for (uint32_t b = 0; b < 256; b++) {
            if (hist[b] > pass_max) return c->fail(20, "generic path: one top-byte range of k-mers alone exceeds what a pass can hold");
            if (acc + hist[b] > pass_max) { ranges.push_back({lo, b}); range_n.push_back(acc); lo = b; acc = 0; }
            acc += hist[b];
        }
        ranges.push_back({lo, 256u}); range_n.push_back(acc);
    }
    std::vector<ReduceOut> parts(ranges.size());
    auto free_parts = [&]() {
        for (auto& r : parts) for (void* q : {(void*)r.key_hi, (void*)r.key_lo, (void*)r.exts, (void*)r.count, (void*)r.set_off, (void*)r.set_val, (void*)r.all_hi, (void*)r.all_lo}) c->dfree(q);
    };
    int rc = 0;
    for (size_t pi = 0; pi < ranges.size() && !rc; pi++) {
        const uint64_t np = range_n[pi];
        DBuf<uint64_t> a_hi, a_lo, b_hi, b_lo;
        DBuf<uint32_t> a_pay, b_pay;
        const size_t nalloc = std::max<uint64_t>(np, 1);
        bool ok = (!has_hi || (a_hi.alloc(c, nalloc) && b_hi.alloc(c, nalloc))) && a_lo.alloc(c, nalloc) && b_lo.alloc(c, nalloc) &&
                  a_pay.alloc(c, nalloc) && b_pay.alloc(c, nalloc);
        if (!ok) { rc = c->fail(101, "device allocation failed in the generic path"); break; }
        RecArrays A{a_hi.p, a_lo.p, a_pay.p}, B{b_hi.p, b_lo.p, b_pay.p};
        if (ranges.size() == 1) rc = extract_kmers(c, s, koff.p, np, k, p->stranded != 0, A);
        else {
            rc = kmer_counts_range(c, s, k, p->stranded != 0, ranges[pi].first, ranges[pi].second, kcount.p);
            if (!rc) rc = scan_exclusive_u32_u64(c, kcount.p, koff.p, s.n);
            if (!rc) rc = extract_kmers_range(c, s, koff.p, np, k, p->stranded != 0, ranges[pi].first, ranges[pi].second, A);
        }
        bool in_b = false;
        // CountFilterSet needs (key, D1) order so distinct labels are adjacent; D1 < 2^24
        const int pay_bits = is_set && s.data ? (s.data_width == 1 ? 8 : (s.data_width == 2 ? 16 : 24)) : 0;
        if (!rc) rc = radix_sort_records(c, np, A, B, 2 * k, 8, pay_bits, &in_b);
        if (!rc) rc = reduce_sorted_records(c, np, in_b ? B : A, has_hi, p->summarizer, p->min_kmer_obs, p->report_all_kmers != 0, &parts[pi]);
    }
    if (rc) { free_parts(); return rc; }
    ReduceOut r;
    DBG_TRY(join_reduce_parts(c, parts, is_set, p->report_all_kmers != 0, &r));
    if (!has_hi && r.n_valid) HIP_TRY(c, hipMemsetAsync(r.key_hi, 0, r.n_valid * 8, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    table_from_reduce(r, n_kmers, out);
    out->n_passes = (uint32_t)ranges.size();
    return 0;
}

extern "C" int dbg_table_to_host(dbg_ctx* c, const dbg_kmer_table* d, dbg_kmer_table* h) {
    *h = *d;
    h->on_device = 0;
    h->key_hi = h->key_lo = h->set_off = h->all_hi = h->all_lo = nullptr;
    h->exts = nullptr; h->count = nullptr; h->set_val = nullptr;
    // the arrays leave one after the other on ONE stream: a single device-to-host copy runs at 57 GB/s, two in flight on two
    // streams share the link at 47 GB/s in all (tools/micro/pcie_bw.hip, profiles/r05_pcie_bw.txt; rounds 2-4 used two streams);
    // destinations are pinned blocks of the ctx pool
    hipStream_t st[2] = {c->stream, c->stream};
    int which = 0;
    hipError_t e = hipSuccess;
#define CP(field, T, cnt)                                                                             \
    if (d->field && e == hipSuccess) {                                                                \
        h->field = (T*)ctx_halloc(c, (size_t)(cnt) * sizeof(T));                                      \
        if (!h->field) e = hipErrorOutOfMemory;                                                       \
        else if ((cnt)) e = hipMemcpyAsync(h->field, d->field, (size_t)(cnt) * sizeof(T), hipMemcpyDeviceToHost, st[which++ & 1]); \
    }
    CP(key_hi, uint64_t, d->n) CP(key_lo, uint64_t, d->n) CP(exts, uint8_t, d->n) CP(count, uint16_t, d->n)
    CP(set_off, uint64_t, d->n + 1) CP(set_val, uint32_t, d->n_set_val) CP(all_hi, uint64_t, d->n_all) CP(all_lo, uint64_t, d->n_all)
#undef CP
    if (e == hipSuccess) e = hipStreamSynchronize(st[0]);
    if (e == hipSuccess && st[1] != st[0]) e = hipStreamSynchronize(st[1]);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        dbg_free_table(c, h);
        return c->fail(100, std::string("HIP error while copying the table to the host: ") + hipGetErrorString(e));
    }
    return 0;
}

extern "C" void dbg_free_table(dbg_ctx* c, dbg_kmer_table* t) {
    if (!t) return;
    void* ptrs[] = {t->key_hi, t->key_lo, t->exts, t->count, t->set_off, t->set_val, t->all_hi, t->all_lo};
    for (void* p : ptrs) {
        if (!p) continue;
        if (t->on_device) { if (c) c->dfree(p); } else ctx_hfree(c, p);
    }
    memset(t, 0, sizeof(*t));
}

// host-pointer form: the Rust call site's shape (filter.rs:139-148); stages through PCIe
struct DevSeqSet {
    DBuf<uint64_t> words, start;
    DBuf<uint32_t> length;
    DBuf<uint8_t> exts, data;
    dbg_seqset view;
};
// words_too = false: everything but the packed words (the caller streams those in chunks of reads, dbg_filter_kmers)
static int upload_seqset_parts(dbg_ctx* c, const dbg_seqset* hs, DevSeqSet* d, bool words_too) {
    uint64_t n = hs->n_seqs;
    // the device kernels never read past the last word a sequence touches, so n_words suffices
    ALLOC_OR_FAIL(c, d->words, std::max<uint64_t>(hs->n_words, 1));
    ALLOC_OR_FAIL(c, d->start, std::max<uint64_t>(n, 1));
    ALLOC_OR_FAIL(c, d->length, std::max<uint64_t>(n, 1));
    std::vector<UploadJob> jobs;
    if (words_too) jobs.push_back({d->words.p, hs->words, (size_t)hs->n_words * 8});
    jobs.push_back({d->start.p, hs->start, (size_t)n * 8});
    jobs.push_back({d->length.p, hs->length, (size_t)n * 4});
    d->view = *hs;
    d->view.words = d->words.p; d->view.start = d->start.p; d->view.length = d->length.p;
    d->view.exts = nullptr; d->view.data = nullptr;
    if (hs->exts) {
        ALLOC_OR_FAIL(c, d->exts, std::max<uint64_t>(n, 1));
        jobs.push_back({d->exts.p, hs->exts, (size_t)n});
        d->view.exts = d->exts.p;
    }
    if (hs->data && hs->data_width) {
        ALLOC_OR_FAIL(c, d->data, std::max<uint64_t>(n * hs->data_width, 1));
        jobs.push_back({d->data.p, hs->data, (size_t)n * hs->data_width});
        d->view.data = d->data.p;
    }
    return staged_upload(c, jobs);
}
int upload_seqset(dbg_ctx* c, const dbg_seqset* hs, DevSeqSet* d) { return upload_seqset_parts(c, hs, d, true); }

// chunk_end (may be null): [n_chunks] receives, for the reads [n * g / n_chunks, n * (g + 1) / n_chunks), the largest end base of a
// read of that or an earlier chunk; *monotone = the reads' start offsets never decrease (PackedDnaStringSet::add appends, so they do
// not in practice): then the words a chunk of reads needs are a prefix of the array
static int check_host_seqset(dbg_ctx* c, const dbg_seqset* hs, uint32_t n_chunks = 0, uint64_t* chunk_end = nullptr, bool* monotone = nullptr) {
    // bounds checks that would be undefined behaviour on the device (on the host threads the container is granted)
    std::atomic<int> bad{0};
    std::vector<std::atomic<uint64_t>> cend(n_chunks);
    for (auto& v : cend) v = 0;
    std::vector<uint64_t> upto(n_chunks);                            // chunk g = reads [upto[g - 1], upto[g])
    for (uint32_t g = 0; g < n_chunks; g++) upto[g] = (uint64_t)(((unsigned __int128)hs->n_seqs * (g + 1)) / n_chunks);
    host_parallel_ranges(hs->n_seqs, [&](uint64_t a, uint64_t b, unsigned) {
        int f = 0;
        uint64_t prev = a ? hs->start[a - 1] : 0, far = 0;
        uint32_t g = 0;
        while (n_chunks && g + 1 < n_chunks && a >= upto[g]) g++;
        auto flush = [&]() {
            if (!n_chunks || !far) return;
            uint64_t cur = cend[g].load(std::memory_order_relaxed);
            while (far > cur && !cend[g].compare_exchange_weak(cur, far, std::memory_order_relaxed)) {}
            far = 0;
        };
        for (uint64_t i = a; i < b; i++) {
            const uint64_t st = hs->start[i], end = st + hs->length[i];
            if ((end + 31) / 32 > hs->n_words && hs->length[i]) f |= 1;
            if (n_chunks) {
                if (i >= upto[g]) { flush(); while (g + 1 < n_chunks && i >= upto[g]) g++; }
                if (st < prev) f |= 4;
                prev = st;
                far = end > far ? end : far;
            }
        }
        flush();
        if (f) bad |= f;
    });
    if (n_chunks) {
        uint64_t run = 0;
        for (uint32_t g = 0; g < n_chunks; g++) { run = std::max<uint64_t>(run, cend[g]); chunk_end[g] = run; }
        *monotone = !(bad & 4);
    }
    if (bad & 1) return c->fail(16, "sequence runs past n_words");
    return 0;
}

// A sequence set resident in HBM for hosts that do not drive the HIP runtime themselves (a Rust shim without HIP bindings): the
// same checked, staged upload the host-pointer calls perform, with the device buffers handed to the caller (pool blocks of the ctx;
// dbg_seqset_free_device returns them).
extern "C" int dbg_seqset_to_device(dbg_ctx* c, const dbg_seqset* hs, dbg_seqset* dev_out) {
    if (!c) return 1;
    if (!hs || !dev_out) return c->fail(10, "null argument");
    if (hs->n_seqs && (!hs->words || !hs->start || !hs->length)) return c->fail(10, "null argument");
    if (hs->data && hs->data_width != 1 && hs->data_width != 2 && hs->data_width != 4) return c->fail(15, "data_width must be 1, 2 or 4");
    DBG_TRY(check_host_seqset(c, hs));
    HIP_TRY(c, hipSetDevice(c->device));
    DevSeqSet d;
    DBG_TRY(upload_seqset(c, hs, &d));
    *dev_out = d.view;
    d.words.take(); d.start.take(); d.length.take(); d.exts.take(); d.data.take();      // the caller's now (live pool blocks)
    return 0;
}
extern "C" void dbg_seqset_free_device(dbg_ctx* c, dbg_seqset* dev) {
    if (!c || !dev) return;
    c->dfree((void*)dev->words); c->dfree((void*)dev->start); c->dfree((void*)dev->length); c->dfree((void*)dev->exts); c->dfree((void*)dev->data);
    memset(dev, 0, sizeof(*dev));
}

// compact CSR of a host table (dbg_filter_params.compact_sets): offsets and labels narrowed on the device, before they cross PCIe
template <class T>
__global__ void __launch_bounds__(256) narrow_kernel(const uint64_t* __restrict__ in64, const uint32_t* __restrict__ in32, uint64_t n, T* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (T)(in64 ? in64[i] : (uint64_t)in32[i]);
}

extern "C" int dbg_filter_kmers(dbg_ctx* c, const dbg_seqset* hs, const dbg_filter_params* p, dbg_kmer_table* out) {
    DBG_TRY(validate_filter(c, hs, p));
    HIP_TRY(c, hipSetDevice(c->device));
    DevSeqSet d;
    dbg_kmer_table dev;
    bool counted = false;
    const bool dbg_t = c->opt("DBG_DEBUG") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!dbg_t) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[host boundary] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
        t_last = t;
    };
    // Large inputs on the fast path: the packed words (three quarters of the input bytes) are uploaded in chunks of reads by a
    // helper thread while the scan runs over the chunks that have arrived (dbg_ctx::read_gates; fast_scan launches once per
    // chunk) -- the scan's ~30 ms per 10^8 reads disappear under the upload.  Everything else waits for the last chunk.
    constexpr uint32_t N_GATES = 8;
    const char* hmode = c->opt("DBG_HOST_STAGING");
    const bool chunked = (size_t)hs->n_words * 8 >= (256u << 20) && hs->n_seqs >= 4096 && p->k >= 16 && !(hmode && (!strcmp(hmode, "off") || !strcmp(hmode, "whole"))) &&
                         !(c->opt("DBG_PATH") && strcmp(c->opt("DBG_PATH"), "auto") && strcmp(c->opt("DBG_PATH"), "fast"));
    {
        // the bounds / label checks of the caller's arrays (a pass over start[] and length[] on the host threads) run next to the
        // upload -- copying unchecked arrays is harmless, running kernels on them is not: the check is joined before anything is launched
        int check_rc = 0;
        std::string check_err;
        uint64_t chunk_end[N_GATES];
        bool monotone = false;
        std::thread checker([&] { dbg_ctx tmp; check_rc = check_host_seqset(&tmp, hs, chunked ? N_GATES : 0, chunk_end, &monotone); check_err = tmp.err; });
        const int up = upload_seqset_parts(c, hs, &d, !chunked);
        lap(chunked ? "upload (all but the words)" : "upload");
        checker.join();
        lap("input checks (joined)");
        if (check_rc) return c->fail(check_rc, check_err);
        DBG_TRY(up);
        if (chunked && !monotone) {                                  // (reads out of order: the whole array first, as for small inputs)
            std::vector<UploadJob> jobs{{d.words.p, hs->words, (size_t)hs->n_words * 8}};
            DBG_TRY(staged_upload(c, jobs));
        } else if (chunked) {
            dbg_read_gates gates;
            for (uint32_t g = 0; g < N_GATES; g++) gates.upto.push_back((uint64_t)(((unsigned __int128)hs->n_seqs * (g + 1)) / N_GATES));
            staged_upload_prepare(c);
            HIP_TRY(c, hipStreamSynchronize(c->stream));             // (the words block may still be in use by queued work of its previous owner)
            uint64_t* dw = d.words.p;
            std::thread uploader([&, dw] {
                (void)hipSetDevice(c->device);
                uint64_t done_words = 0;
                for (uint32_t g = 0; g < N_GATES; g++) {
                    // chunk g's reads end at base chunk_end[g]: the words up to there (the last chunk: everything, padding included)
                    const uint64_t upto_w = g + 1 == N_GATES ? hs->n_words : std::min<uint64_t>(hs->n_words, (chunk_end[g] + 31) / 32 + 1);
                    if (upto_w > done_words) {
                        std::string err;
                        std::vector<UploadJob> jobs{{dw + done_words, hs->words + done_words, (size_t)(upto_w - done_words) * 8}};
                        if (int e = staged_upload_quiet(c, jobs, &err)) { gates.publish(g, e, err); return; }
                        done_words = upto_w;
                    }
                    gates.publish(g + 1);
                }
            });
            c->read_gates = &gates;
            const int r = dbg_filter_kmers_dev(c, &d.view, p, &dev);
            c->read_gates = nullptr;
            lap("filter (gated on the upload)");
            uploader.join();
            lap("uploader joined");
            if (r) return r;
            if (gates.err) { dbg_free_table(c, &dev); return c->fail(gates.err, gates.msg); }
            counted = true;
        }
    }
    if (!counted) { DBG_TRY(dbg_filter_kmers_dev(c, &d.view, p, &dev)); lap("filter"); }
    d.words.release(); d.start.release(); d.length.release(); d.exts.release(); d.data.release();     // (the reads are no longer needed: room for the copies below)
    int r = 0;
    if ((p->compact_sets & 2u) && p->k <= 32 && dev.key_hi) {        // k <= 32: key_hi is all zeros -- 8 bytes per k-mer that need not cross PCIe
        c->dfree(dev.key_hi);
        dev.key_hi = nullptr;
    }
    if ((p->compact_sets & 1u) && dev.set_off) {
        const uint32_t ow = dev.n_set_val < (1ull << 32) ? 4u : 8u;
        const uint32_t vw = hs->data && (hs->data_width == 1 || hs->data_width == 2) ? hs->data_width : 4u;
        DBuf<uint32_t> off32, val32;
        DBuf<uint16_t> val16;
        DBuf<uint8_t> val8;
        r = [&]() -> int {
            const uint64_t no = dev.n + 1, nv = dev.n_set_val;
            if (ow == 4) {
                ALLOC_OR_FAIL(c, off32, no);
                narrow_kernel<uint32_t><<<cdiv(no, 256), 256, 0, c->stream>>>(dev.set_off, nullptr, no, off32.p);
                LAUNCH_CHECK(c, "narrow_set_off");
            }
            if (vw == 1 && nv) { ALLOC_OR_FAIL(c, val8, nv); narrow_kernel<uint8_t><<<cdiv(nv, 256), 256, 0, c->stream>>>(nullptr, dev.set_val, nv, val8.p); LAUNCH_CHECK(c, "narrow_set_val"); }
            if (vw == 2 && nv) { ALLOC_OR_FAIL(c, val16, nv); narrow_kernel<uint16_t><<<cdiv(nv, 256), 256, 0, c->stream>>>(nullptr, dev.set_val, nv, val16.p); LAUNCH_CHECK(c, "narrow_set_val"); }
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            // the plain columns leave as they are; the CSR leaves narrow
            dbg_kmer_table plain = dev;
            plain.set_off = nullptr; plain.set_val = nullptr;
            DBG_TRY(dbg_table_to_host(c, &plain, out));
            out->n_set_val = nv;
            const size_t ob = (size_t)no * ow, vb = (size_t)std::max<uint64_t>(nv, 1) * vw;
            out->set_off = (uint64_t*)ctx_halloc(c, ob);
            out->set_val = (uint32_t*)ctx_halloc(c, vb);
            if (!out->set_off || !out->set_val) { dbg_free_table(c, out); return c->fail(101, "host allocation of the compact CSR failed"); }
            hipStream_t s2 = c->stream;                                  // (one copy at a time: see dbg_table_to_host)
            HIP_TRY(c, hipMemcpyAsync(out->set_off, ow == 4 ? (const void*)off32.p : (const void*)dev.set_off, ob, hipMemcpyDeviceToHost, c->stream));
            if (nv) HIP_TRY(c, hipMemcpyAsync(out->set_val, vw == 1 ? (const void*)val8.p : (vw == 2 ? (const void*)val16.p : (const void*)dev.set_val), (size_t)nv * vw, hipMemcpyDeviceToHost, s2));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (s2 != c->stream) HIP_TRY(c, hipStreamSynchronize(s2));
            out->set_off_width = ow; out->set_val_width = vw;
            return 0;
        }();
        if (r) dbg_free_table(c, out);
    } else r = dbg_table_to_host(c, &dev, out);
    lap("table to the host");
    dbg_free_table(c, &dev);
    return r;
}

extern "C" int dbg_synth_reads_dev(dbg_ctx* c, const dbg_synth_params* p, uint64_t* words, uint64_t* start,
                                   uint32_t* length, uint8_t* data) {
    HIP_TRY(c, hipSetDevice(c->device));
    DBG_TRY(synth_reads_dev(c, p, words, start, length, data));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}
