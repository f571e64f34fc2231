// Host-side context: device, stream, pooled device memory, error string, per-kernel HIP-event timing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <map>
#include <set>
#include <algorithm>
#include <mutex>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <chrono>
#include "../../include/dbg_mi355x.h"

// host seconds spent inside the driver's allocation calls (dbg_ctx_get_stats: the cold-call account)
struct HostTimer {
    double* acc;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit HostTimer(double* a) : acc(a) {}
    ~HostTimer() { *acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

struct dbg_timing_rec {
    const char* name;
    hipEvent_t a, b;
    uint64_t units;
};

// State that one API call leaves for the next one of the same ctx (dbg_shard_scan_dev -> dbg_shard_scatter_dev,
// dbg_shard_count_begin -> _bins_dev -> _finish).  Owned by the ctx: the library has no globals.
struct dbg_state_slot {
    void* p = nullptr;
    void (*del)(void*) = nullptr;
    void reset(void* q = nullptr, void (*d)(void*) = nullptr) { if (p && del) del(p); p = q; del = d; }
    void* release() { void* q = p; p = nullptr; del = nullptr; return q; }
    ~dbg_state_slot() { reset(); }
};

// Diagnostic knobs.  The environment is read ONCE, in dbg_ctx_create; afterwards
// dbg_ctx_set_option changes a knob of one ctx.  Names are those of the environment variables.
static const char* const DBG_OPTION_NAMES[] = {
    "DBG_PATH", "DBG_COMPRESS", "DBG_FAST_TARGET", "DBG_FAST_NT", "DBG_FAST_TABLE", "DBG_NO_HYBRID_SORT", "DBG_NO_REC16",
    "DBG_FAST_NO_SLAB", "DBG_DEBUG", "DBG_UNITIG_NO_WALK", "DBG_UNITIG_NO_CHAINS", "DBG_NO_KEY_RECORDS", "DBG_NO_NODE_RECORDS",
    "DBG_PIDX_BITS", "DBG_SORT", "DBG_DYN_LDS", "DBG_GENERIC_PASS_MAX", "DBG_FAST_P", "DBG_HOST_STAGING", "DBG_SCAN", "DBG_MSP", "DBG_SLAB_CAP", "DBG_ONESWEEP", "DBG_NO_LABEL_GROUPS", "DBG_NO_STRAND_NORM", "DBG_SHARD_MERGE", "DBG_LINKS", "DBG_DENSE_RANGES", "DBG_CHAIN_WALKS", "DBG_DENSE_PART", "DBG_DENSE_BATCH", "DBG_DENSE_L1", "DBG_DENSE_RAW",
    "DBG_FAIL_AT", "DBG_COMM_TIMEOUT_S", "DBG_SHARD_MERGE_COST_MS", "DBG_LABEL_LISTS", "DBG_SLAB_TRIALS", "DBG_UNITIG_FULL_DOUBLING", "DBG_WALK_CAP", "DBG_CHAIN_CAP", "DBG_FAST_PERSIST", "DBG_SEGMENTS", "DBG_SEGMENTS_FORCE"};   // (the last two: fault injection and the bound on communication waits of the rank-spanning calls, shard_comm.hpp)   // (DBG_MSP: wave | twopass)

// Reads whose packed words are still on their way to the device (dbg_filter_kmers: the upload of the caller's words runs in chunks
// of reads next to the scan of the chunks that have arrived, api.hip).  Chunk g is complete once reads [0, upto[g]) have their words
// in HBM.  Everything that reads the words of all sequences waits for the last gate first (dbg_ctx::wait_all_reads).
struct dbg_read_gates {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<uint64_t> upto;
    size_t ready = 0;
    int err = 0;
    std::string msg;
    void publish(size_t n_ready, int e = 0, const std::string& m = std::string()) {
        std::lock_guard<std::mutex> g(mu);
        if (n_ready > ready) ready = n_ready;
        if (e && !err) { err = e; msg = m; }
        cv.notify_all();
    }
    int wait(size_t g) {                                   // until chunk g is complete (or the upload failed)
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return ready > g || err; });
        return err;
    }
};

struct dbg_ctx {
    int device = 0;
    int n_cu = 256;                                          // compute units of the device (dbg_ctx_create)
    dbg_read_gates* read_gates = nullptr;
    int wait_all_reads() {
        if (!read_gates || read_gates->upto.empty()) return 0;
        const int e = read_gates->wait(read_gates->upto.size() - 1);
        return e ? fail(e, read_gates->msg) : 0;
    }
    std::map<std::string, std::string> opts;
    const char* opt(const char* name) const {
        auto it = opts.find(name);
        return it == opts.end() ? nullptr : it->second.c_str();
    }
    dbg_state_slot shard_scan, shard_count;
    // what the last dbg_shard_filter_kmers_dev of this ctx measured: input of the next call's sender-side-merge decision
    bool shard_last_valid = false, shard_last_merge = false;
    double shard_last_exposed_ms = 0.0, shard_last_merge_cost_ms = 0.0, shard_last_wire_ms = 0.0, shard_last_merge_ratio = 1.0;
    uint32_t* cycle_seed_list = nullptr;    // when set (segment route, second level): cut_cycles_kernel appends the seed element of every cycle it cuts
    uint32_t* cycle_seed_count = nullptr;
    uint32_t cycle_seed_cap = 0;
    uint32_t* segment_outer_links = nullptr;   // first level of the segment route: the call's own link array (cycles no sampled cut fell into are cut there, at their seeds)
    uint32_t segment_new_cuts = 0;
    int segment_depth = 0;                  // > 0 inside the first level of the segment route (unitig.hip): no further nesting, no general route
    struct GraphDev* graph_sink = nullptr;   // when set, compress_links_device leaves its BaseGraph in HBM there instead of copying it to the host (graph.hip: second stage)
    dbg_state_slot stager;                  // pinned staging ring of the host-boundary uploads (hostio.hip)
    std::multimap<size_t, void*> hfree_blocks;   // pinned host blocks kept for result arrays
    std::map<void*, size_t> hlive_blocks;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    hipStream_t copy_stream = nullptr;     // created on first use: result copies that overlap kernels of `stream`
    hipStream_t get_copy_stream() {
        if (!copy_stream && hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking) != hipSuccess) copy_stream = nullptr;
        return copy_stream;
    }
    // Communication stream of the rank-spanning flow (exchange rounds, per-round compaction).  Created with the HIGHEST stream priority:
    // the runtime maps streams of one priority onto a small pool of hardware queues, and two streams that share a queue do not
    // overlap at all (measured, round 4: with an ordinary second stream the compaction queued for round r + 1 ran strictly BEFORE the
    // counting kernel of round r, same queue id in the trace); a stream of another priority gets a queue of its own, and its kernels
    // -- a copy, RCCL's send / receive -- are dispatched ahead of the counting kernel's 10^5 queued workgroups instead of behind them.
    hipStream_t comm_stream = nullptr;
    hipStream_t get_comm_stream() {
        if (!comm_stream) {
            int lo = 0, hi = 0;                                  // (numerically lower = higher priority)
            if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = hi = 0; }
            if (hipStreamCreateWithPriority(&comm_stream, hipStreamNonBlocking, hi) != hipSuccess) {
                (void)hipGetLastError();
                if (hipStreamCreateWithFlags(&comm_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); comm_stream = nullptr; }
            }
        }
        return comm_stream;
    }
    std::string err;
    uint64_t scratch_budget = 0;
    bool timing = false;
    std::vector<dbg_timing_rec> trecs;
    std::vector<hipEvent_t> event_pool;
    // pooled device allocations: free blocks by size; live blocks by pointer
    // (size, address): among blocks of one size the lowest address serves first, so the same request of the same call sequence gets
    //  the same block every time -- with insertion order two equal-sized buffers swapped blocks from call to call, and the scan's time
    //  alternated by 0.9 ms with the block its cursor array sat in, round 6)
    std::set<std::pair<size_t, void*>> free_blocks;
    std::set<std::pair<size_t, void*>>::iterator pool_find(size_t bytes) { return free_blocks.lower_bound({bytes, nullptr}); }
    std::map<void*, size_t> live_blocks;
    size_t pooled_bytes = 0;
    // ---- allocation account (dbg_ctx_get_stats): what a cold call pays, and how the scan's slab is backed ----
    size_t pooled_high_water = 0;
    uint64_t n_hipmalloc = 0, n_fresh_blocks = 0, n_pool_hits = 0, n_trims = 0, n_oom_retries = 0, n_raw_free = 0, n_pinned_alloc = 0;
    double s_hipmalloc = 0, s_free = 0, s_pinned_alloc = 0;
    uint32_t slab_backing = 0;                     // DBG_SLAB_* of the last slab the scan took
    void* slab_ptr = nullptr;                      // ... the block (it sits in free_blocks between calls)
    size_t slab_bytes = 0;
    uint32_t slab_rec_words = 0;                   // words per record of the scan that used it
    char slab_note[96] = {0};
    // Slab tournament (fastpath.hip, slab_alloc): the first calls of a shape each run their scan on a fresh block, the scan is timed,
    // and the fastest block stays -- a 26 GB block takes the same random 24-byte writes at 31 or 34-36 ms depending on where the driver
    // placed it, reproducibly per block and differently in every process.
    struct SlabCand { void* p; size_t bytes; double ns_per_kmer; };
    std::vector<SlabCand> slab_cands;
    uint64_t slab_trial_key = 0;
    uint32_t slab_trials_done = 0;
    float slab_trial_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    void note_pool() { if (pooled_bytes > pooled_high_water) pooled_high_water = pooled_bytes; }

    int fail(int code, const std::string& msg) { err = msg; return code; }

    void* dalloc(size_t bytes) {
        if (bytes == 0) bytes = 256;
        bytes = (bytes + 255) & ~(size_t)255;
        auto it = pool_find(bytes);
        if (it != free_blocks.end() && it->first <= bytes + bytes / 4 + 4096) {
            void* p = it->second;
            live_blocks[p] = it->first;
            free_blocks.erase(it);
            n_pool_hits++;
            return p;
        }
        void* p = nullptr;
        hipError_t e;
        { HostTimer t_(&s_hipmalloc); e = hipMalloc(&p, bytes); n_hipmalloc++; }
        if (e != hipSuccess) {
            n_oom_retries++;
            (void)hipGetLastError();            // (the failure is sticky: the next launch check would report it as its own)
            // the device is full: a pooled block of up to twice the size serves before everything pooled is given back (trim() also
            // unmaps the piecewise-mapped slabs, which the next call then maps again: seconds at 100 GB -- label lists at k = 20 and
            // 10^8 reads went from 0.57 s of kernels to 5.3 s of wall time that way)
            if (it != free_blocks.end() && it->first <= 2 * bytes) {
                void* q = it->second;
                live_blocks[q] = it->first;
                free_blocks.erase(it);
                return q;
            }
            trim();
            { HostTimer t_(&s_hipmalloc); e = hipMalloc(&p, bytes); n_hipmalloc++; }
            if (e != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        }
        live_blocks[p] = bytes;
        pooled_bytes += bytes;
        note_pool();
        return p;
    }
    void dfree(void* p) {
        if (!p) return;
        auto it = live_blocks.find(p);
        if (it == live_blocks.end()) { raw_free(p); return; }
        free_blocks.insert({it->second, p});
        live_blocks.erase(it);
    }
    // A block of the asked size straight from the driver, not from the pool (a candidate of the slab tournament, fastpath.hip).
    // Round 5 mapped slabs piecewise from physical handles (hipMemCreate / hipMemMap) in the belief that this fixed their placement
    // kind; round 6 measured that it does not (any handle size, 2 MB .. 2 GB, probes at 5.4 .. 7.4 ms like a plain block:
    // tools/micro/slab_probe6.hip, slab_probe7.hip) and that RELEASING such a block is unsafe on this driver: physical pages that
    // return to the driver and are mapped again are zero-filled asynchronously, without a fence the user queues wait for, so the clear
    // lands on top of data kernels have already written into the new mapping (tools/micro/vmm_reuse.hip: 2e7 zeroed words per 26 GB
    // round; the bench failed with "corrupt super-k-mer record" in 2 of 5 processes).  hipMalloc / hipFree do not show it.
    void* dalloc_fresh(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        void* p = nullptr;
        hipError_t e;
        { HostTimer t_(&s_hipmalloc); e = hipMalloc(&p, bytes); n_hipmalloc++; }
        if (e != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        live_blocks[p] = bytes;
        pooled_bytes += bytes;
        note_pool();
        return p;
    }
    // give a block back to the driver, whichever way it was obtained
    void raw_free(void* p) {
        HostTimer t_(&s_free);
        n_raw_free++;
        if (p == slab_ptr) slab_ptr = nullptr;
        for (size_t i = 0; i < slab_cands.size(); i++) if (slab_cands[i].p == p) { slab_cands.erase(slab_cands.begin() + i); break; }
        (void)hipFree(p);
    }
    void trim() {
        n_trims++;
        (void)hipStreamSynchronize(stream);
        for (auto& kv : free_blocks) { raw_free(kv.second); pooled_bytes -= kv.first; }
        free_blocks.clear();
    }
    hipEvent_t get_event() {
        if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; }
        hipEvent_t e; (void)hipEventCreate(&e); return e;
    }
    void t_begin(const char* name, uint64_t units) {
        if (!timing) return;
        dbg_timing_rec r{name, get_event(), get_event(), units};
        (void)hipEventRecord(r.a, stream);
        trecs.push_back(r);
    }
    void t_end() {
        if (!timing) return;
        (void)hipEventRecord(trecs.back().b, stream);
    }
    void t_clear() {
        for (auto& r : trecs) { event_pool.push_back(r.a); event_pool.push_back(r.b); }
        trecs.clear();
    }
};

// RAII device buffer drawn from the ctx pool
template <class T>
struct DBuf {
    dbg_ctx* ctx = nullptr;
    T* p = nullptr;
    size_t n = 0;
    DBuf() {}
    DBuf(dbg_ctx* c, size_t count) { alloc(c, count); }
    DBuf(const DBuf&) = delete;
    DBuf& operator=(const DBuf&) = delete;
    DBuf(DBuf&& o) noexcept : ctx(o.ctx), p(o.p), n(o.n) { o.p = nullptr; }
    DBuf& operator=(DBuf&& o) noexcept { release(); ctx = o.ctx; p = o.p; n = o.n; o.p = nullptr; return *this; }
    bool alloc(dbg_ctx* c, size_t count) {
        release();
        ctx = c; n = count;
        p = (T*)c->dalloc(count * sizeof(T));
        return p != nullptr;
    }
    void release() { if (p && ctx) ctx->dfree(p); p = nullptr; }
    T* take() { T* q = p; p = nullptr; return q; }      // ownership leaves the RAII wrapper (stays a live pool block)
    ~DBuf() { release(); }
};

#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            char _b[512];                                                                       \
            snprintf(_b, sizeof(_b), "HIP error %s at %s:%d (%s)", hipGetErrorString(_e), __FILE__, __LINE__, #expr); \
            return (ctx)->fail(100, _b);                                                        \
        }                                                                                       \
    } while (0)

#define DBG_TRY(expr)                 \
    do {                              \
        int _r = (expr);              \
        if (_r) return _r;            \
    } while (0)

#define ALLOC_OR_FAIL(ctx, buf, count)                                                          \
    do {                                                                                        \
        if (!(buf).alloc((ctx), (count))) {                                                     \
            char _b[256];                                                                       \
            snprintf(_b, sizeof(_b), "device allocation of %zu bytes failed at %s:%d",          \
                     (size_t)(count) * sizeof(*(buf).p), __FILE__, __LINE__);                   \
            return (ctx)->fail(101, _b);                                                        \
        }                                                                                       \
    } while (0)

static inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }
