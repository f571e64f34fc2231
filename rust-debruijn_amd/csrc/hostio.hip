// The reference's own boundary takes and returns HOST memory (filter_kmers(&[(V, Exts, D1)]) -> BoomHashMap2, src/filter.rs:139-148;
// compress_kmers_with_hash(&BoomHashMap2) -> BaseGraph, src/compression.rs:588-594), so the unsuffixed entry points move GBs over
// PCIe around ~0.15 s of kernels.  Two things make that fast:
//   * uploads of the caller's (pageable) arrays go through a ring of pinned staging buffers filled by several host threads, each
//     with its own stream: the CPU copies of chunk i+1.. overlap the DMA of chunk i, and PCIe, not one memcpy thread, is the limit
//     (a plain hipMemcpy from pageable memory stages through one thread: ~12 GB/s measured);
//   * result arrays are handed out from a per-ctx pool of pinned host blocks (hipHostMalloc is slow, ~GB/s, so blocks are kept
//     and reused): device-to-host copies then run at PCIe speed straight into the caller-visible arrays, on two streams.
#include "dbg_internal.hpp"
#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {
// every pinned result block -> the ctx whose pool it belongs to, so that a table released without its ctx (dbg_free_table(NULL, ..))
// still finds its way back, and a ctx that is destroyed while tables are alive can disown them instead of freeing them
std::mutex g_pin_mu;
std::unordered_map<void*, dbg_ctx*> g_pin_owner;

constexpr size_t STAGE_CHUNK = 16u << 20;       // bytes per staging buffer
constexpr size_t STAGE_MIN = 8u << 20;          // smaller transfers take the plain copy

struct UploadLane {
    void* buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
};
struct Stager {
    std::vector<UploadLane> lanes;
    // ONE stream carries all the DMA: two host-to-device copies in flight on two streams share the link at 47 GB/s in all, one
    // after the other they run at 57 (tools/micro/pcie_bw.hip, profiles/r05_pcie_bw.txt); the lanes' threads only fill the pinned
    // buffers (8 threads copy pageable memory at ~97 GB/s) and queue their copies here, one at a time
    hipStream_t dma = nullptr;
    std::mutex mu;
    ~Stager() {
        if (dma) (void)hipStreamDestroy(dma);
        for (auto& l : lanes) {
            for (int b = 0; b < 2; b++) { if (l.buf[b]) (void)hipHostFree(l.buf[b]); if (l.ev[b]) (void)hipEventDestroy(l.ev[b]); }
        }
    }
};
void stager_delete(void* p) { delete static_cast<Stager*>(p); }

unsigned host_threads() {
    unsigned n = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min<unsigned>(n ? n : 64, (unsigned)CPU_COUNT(&set));
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {       // container CPU quota (cgroup v2)
        char q[32]; long long per = 0;
        if (fscanf(f, "%31s %lld", q, &per) == 2 && strcmp(q, "max") && per > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, (atoll(q) + per - 1) / per));
        fclose(f);
    }
    return std::max(1u, std::min(n, 8u));
}

Stager* get_stager(dbg_ctx* c) {
    if (c->stager.p) return static_cast<Stager*>(c->stager.p);
    Stager* s = new Stager();
    const unsigned nt = host_threads();
    s->lanes.resize(nt);
    bool ok = hipStreamCreateWithFlags(&s->dma, hipStreamNonBlocking) == hipSuccess;
    HostTimer t_(&c->s_pinned_alloc);
    for (auto& l : s->lanes) {
        for (int b = 0; b < 2 && ok; b++)
            ok = hipHostMalloc(&l.buf[b], STAGE_CHUNK, hipHostMallocDefault) == hipSuccess && hipEventCreateWithFlags(&l.ev[b], hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) { (void)hipGetLastError(); delete s; return nullptr; }
    c->stager.reset(s, stager_delete);
    return s;
}
}  // namespace

// host -> device copies of several arrays, complete on return.  drain = false: the destinations are known to be idle (a buffer
// nothing has touched yet) and the ctx stream may be busy with other work (the scan of reads that have already arrived): it is not
// waited for, and errors are reported through *err only -- the caller is a helper thread that must not write the ctx's error slot.
static int staged_upload_impl(dbg_ctx* c, const std::vector<UploadJob>& jobs, bool drain, std::string* err) {
    size_t total = 0;
    for (auto& j : jobs) total += j.bytes;
    const char* mode = c->opt("DBG_HOST_STAGING");            // "off": plain copies (for A/B measurements)
    Stager* s = (total >= STAGE_MIN && !(mode && !strcmp(mode, "off"))) ? (drain ? get_stager(c) : static_cast<Stager*>(c->stager.p)) : nullptr;
    auto bad = [&](const char* what) { (void)hipGetLastError(); *err = what; return 100; };
    if (!s) {
        hipStream_t st = drain ? c->stream : (c->copy_stream ? c->copy_stream : c->stream);
        for (auto& j : jobs) if (j.bytes && hipMemcpyAsync(j.dst, j.src, j.bytes, hipMemcpyHostToDevice, st) != hipSuccess) return bad("HIP error in a host-to-device copy");
        if (hipStreamSynchronize(st) != hipSuccess) return bad("HIP error in a host-to-device copy");
        return 0;
    }
    struct Chunk { char* dst; const char* src; size_t n; };
    std::vector<Chunk> chunks;
    for (auto& j : jobs)
        for (size_t o = 0; o < j.bytes; o += STAGE_CHUNK) chunks.push_back({(char*)j.dst + o, (const char*)j.src + o, std::min(STAGE_CHUNK, j.bytes - o)});
    // the destinations are pool blocks that c->stream may still be using under their previous owner (the pool hands blocks out
    // assuming same-stream ordering): nothing is written from the DMA stream before c->stream has drained
    if (drain && hipStreamSynchronize(c->stream) != hipSuccess) return bad("HIP error while draining the stream");
    std::atomic<size_t> next{0};
    std::atomic<int> e{0};
    const int device = c->device;
    auto work = [&](UploadLane* l) {
        if (hipSetDevice(device) != hipSuccess) { e = 1; return; }
        for (unsigned it = 0;; it++) {
            const size_t i = next.fetch_add(1);
            if (i >= chunks.size() || e) break;
            const int b = it & 1;
            if (it >= 2 && hipEventSynchronize(l->ev[b]) != hipSuccess) { e = 1; break; }       // the buffer's previous DMA is done
            memcpy(l->buf[b], chunks[i].src, chunks[i].n);
            std::lock_guard<std::mutex> g(s->mu);
            if (hipMemcpyAsync(chunks[i].dst, l->buf[b], chunks[i].n, hipMemcpyHostToDevice, s->dma) != hipSuccess ||
                hipEventRecord(l->ev[b], s->dma) != hipSuccess) { e = 1; break; }
        }
    };
    std::vector<std::thread> th;
    for (size_t t = 1; t < s->lanes.size(); t++) th.emplace_back(work, &s->lanes[t]);
    work(&s->lanes[0]);
    for (auto& t : th) t.join();
    if (hipStreamSynchronize(s->dma) != hipSuccess) e = 1;
    if (e) return bad("HIP error in the staged host-to-device upload");
    return 0;
}
int staged_upload(dbg_ctx* c, const std::vector<UploadJob>& jobs) {
    std::string err;
    const int r = staged_upload_impl(c, jobs, true, &err);
    return r ? c->fail(r, err) : 0;
}
// for a helper thread next to a call that is using the ctx: see staged_upload_impl.  staged_upload_prepare (on the ctx's own
// thread, before the helper starts) makes sure the staging ring exists, so that the helper only reads the ctx
void staged_upload_prepare(dbg_ctx* c) { (void)get_stager(c); (void)c->get_copy_stream(); }
int staged_upload_quiet(dbg_ctx* c, const std::vector<UploadJob>& jobs, std::string* err) { return staged_upload_impl(c, jobs, false, err); }

// parallel for over [0, n) in contiguous ranges on the host threads the container is granted
void host_parallel_ranges(uint64_t n, const std::function<void(uint64_t, uint64_t, unsigned)>& fn) {
    const unsigned nt = n < (1u << 20) ? 1u : host_threads();
    if (nt == 1) { fn(0, n, 0); return; }
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(fn, n * t / nt, n * (t + 1) / nt, t);
    fn(0, n / nt, 0);
    for (auto& t : th) t.join();
}
unsigned host_parallel_width() { return host_threads(); }

// ---- pinned result arrays ----------------------------------------------------------------------------------------------
void* ctx_halloc(dbg_ctx* c, size_t bytes) {
    if (bytes < STAGE_MIN) return malloc(bytes ? bytes : 1);
    bytes = (bytes + 4095) & ~(size_t)4095;
    auto it = c->hfree_blocks.lower_bound(bytes);
    if (it != c->hfree_blocks.end() && it->first <= bytes + bytes / 4 + (1u << 20)) {
        void* p = it->second;
        c->hlive_blocks[p] = it->first;
        c->hfree_blocks.erase(it);
        return p;
    }
    void* p = nullptr;
    HostTimer t_(&c->s_pinned_alloc);
    c->n_pinned_alloc++;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        { std::lock_guard<std::mutex> g(g_pin_mu); for (auto& kv : c->hfree_blocks) g_pin_owner.erase(kv.second); }
        for (auto& kv : c->hfree_blocks) (void)hipHostFree(kv.second);        // give the kept blocks back and retry once
        c->hfree_blocks.clear();
        if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return dbg_host_alloc(bytes); }
    }
    c->hlive_blocks[p] = bytes;
    { std::lock_guard<std::mutex> g(g_pin_mu); g_pin_owner[p] = c; }
    return p;
}

void ctx_hfree(dbg_ctx* c, void* p) {
    if (!p) return;
    if (!c) { std::lock_guard<std::mutex> g(g_pin_mu); auto o = g_pin_owner.find(p); if (o != g_pin_owner.end()) c = o->second; }
    if (c) {
        auto it = c->hlive_blocks.find(p);
        if (it != c->hlive_blocks.end()) {
            c->hfree_blocks.insert({it->second, p});
            c->hlive_blocks.erase(it);
            return;
        }
    }
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeHost) { (void)hipHostFree(p); return; }   // a pinned block released without its ctx
    (void)hipGetLastError();
    free(p);
}

void ctx_hrelease_free(dbg_ctx* c) {
    std::lock_guard<std::mutex> g(g_pin_mu);
    for (auto& kv : c->hfree_blocks) { g_pin_owner.erase(kv.second); (void)hipHostFree(kv.second); }
    c->hfree_blocks.clear();
}

// ctx teardown: pooled blocks are freed; blocks still in a caller's hands are disowned, not freed -- a host table may outlive its
// ctx (as plain malloc'ed tables always could) and is then released by dbg_free_table(NULL, ..) through the pointer-attribute route
void ctx_hrelease_all(dbg_ctx* c) {
    std::lock_guard<std::mutex> g(g_pin_mu);
    for (auto& kv : c->hfree_blocks) { g_pin_owner.erase(kv.second); (void)hipHostFree(kv.second); }
    for (auto& kv : c->hlive_blocks) g_pin_owner.erase(kv.first);
    c->hfree_blocks.clear(); c->hlive_blocks.clear();
}
