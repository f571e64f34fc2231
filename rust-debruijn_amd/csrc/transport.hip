// Transport of the rank-spanning flow (include/dbg_mi355x.h, "the rank-spanning flow behind the C ABI"): the RCCL
// implementation of dbg_transport, resolved from librccl at run time, and the host arithmetic of the exchange geometry.
//
// The reference has no transport of its own: its sharded flow is composed by the caller out of independent per-shard calls
// (src/test.rs:433-470).  One process per GPU, RCCL over xGMI: xGMI is point to point, so the variable all-to-all is a
// group of ncclSend / ncclRecv pairs (one per peer) and every link carries only its own pair's bytes.
#include "dbg_ctx.hpp"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <algorithm>
#include <chrono>
#include <mutex>
#include <atomic>

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    // failure handling (optional symbols: a librccl without them simply offers no poll / abort)
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
};

void set_err(char* err, uint64_t n, const std::string& m) {
    if (err && n) { snprintf(err, (size_t)n, "%s", m.c_str()); }
}

// librccl is looked for (1) at the caller's path, (2) among the objects already loaded into the process (a Rust host that
// links RCCL; a Python host whose torch loaded its own copy), (3) by soname.  One table per path for the life of the process:
// the handle is never closed (RCCL keeps threads and device state of its own).
bool load_rccl(const char* path, RcclApi* api, std::string* why) {
    static std::mutex mu;
    static std::map<std::string, RcclApi> cache;
    std::lock_guard<std::mutex> lk(mu);
    const std::string key = path ? path : "";
    auto it = cache.find(key);
    if (it != cache.end()) { *api = it->second; return true; }
    void* h = nullptr;
    if (path && *path) {
        h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!h) { *why = std::string("dlopen(") + path + ") failed: " + dlerror(); return false; }
    } else {
        if (dlsym(RTLD_DEFAULT, "ncclSend")) h = dlopen(nullptr, RTLD_NOW);
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (int i = 0; i < 2 && !h; i++) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        for (int i = 0; i < 2 && !h; i++) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (!h) { *why = "librccl.so.1 not found (pass the path of the librccl the communicator came from)"; return false; }
    }
    RcclApi a;
    a.handle = h;
#define SYM(field, name) do { *(void**)(&a.field) = dlsym(h, name); if (!a.field) { *why = std::string("librccl lacks ") + name; return false; } } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
    SYM(GetErrorString, "ncclGetErrorString"); SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd");
    SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv"); SYM(AllReduce, "ncclAllReduce"); SYM(AllGather, "ncclAllGather");
#undef SYM
    *(void**)(&a.CommGetAsyncError) = dlsym(h, "ncclCommGetAsyncError");
    *(void**)(&a.CommAbort) = dlsym(h, "ncclCommAbort");
    cache[key] = a;
    *api = a;
    return true;
}

struct RcclTransport {
    dbg_transport tab;          // first member: dbg_transport* <-> RcclTransport*
    RcclApi api;
    ncclComm_t comm;
    std::string last;
    std::atomic<bool> dead{false};   // aborted (or seen failing): the communicator is gone, every operation fails at once
};

// No single message above 1 GiB: RCCL transfers of 2 GiB and more were seen to arrive incomplete (round 2), and a message is
// a message whatever the size of the buffer it is cut from.
constexpr uint64_t MSG_MAX = 1ull << 30;

int rc(RcclTransport* t, ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return 0;
    if (r == ncclInProgress) return 0;          // (non-blocking communicators: the operation is queued)
    t->last = std::string(what) + ": " + t->api.GetErrorString(r);
    fprintf(stderr, "[dbg transport rccl] %s\n", t->last.c_str());
    return 1;
}

int rccl_all_reduce(void* self, uint64_t* buf, uint64_t n, int32_t op, void* stream) {
    RcclTransport* t = (RcclTransport*)self;
    if (t->dead.load()) return 1;
    if (!n) return 0;
    return rc(t, t->api.AllReduce(buf, buf, (size_t)n, ncclUint64, op == 1 ? ncclMax : ncclSum, t->comm, (hipStream_t)stream), "ncclAllReduce");
}
int rccl_all_gather(void* self, const void* send, void* recv, uint64_t bytes, void* stream) {
    RcclTransport* t = (RcclTransport*)self;
    if (t->dead.load()) return 1;
    if (!bytes) return 0;
    return rc(t, t->api.AllGather(send, recv, (size_t)bytes, ncclUint8, t->comm, (hipStream_t)stream), "ncclAllGather");
}
int rccl_all_to_allv(void* self, const void* send, const uint64_t* soff, const uint64_t* sbytes, void* recv, const uint64_t* roff,
                     const uint64_t* rbytes, void* stream) {
    RcclTransport* t = (RcclTransport*)self;
    if (t->dead.load()) return 1;
    const int W = t->tab.world, me = t->tab.rank;
    // Peers in rotated order (rank + i), every pair's message cut into <= 1 GiB pieces: piece j of all pairs forms one group,
    // so that sends and receives of a group can progress together and both ends cut a pair's bytes the same way.
    uint64_t mx = 0;
    for (int d = 0; d < W; d++) mx = std::max(mx, std::max(sbytes[d], rbytes[d]));
    // every rank must run the same number of groups: the largest message of the whole job is not known here, but a pair's
    // two ends agree on that pair's size, and a group may be empty on one rank while another still has pieces to move
    const uint64_t pieces = std::max<uint64_t>(1, (mx + MSG_MAX - 1) / MSG_MAX);
    for (uint64_t j = 0; j < pieces; j++) {
        if (rc(t, t->api.GroupStart(), "ncclGroupStart")) return 1;
        int bad = 0;
        for (int i = 0; i < W && !bad; i++) {
            const int to = (me + i) % W, from = (me - i + W) % W;
            const uint64_t so = j * MSG_MAX, ro = j * MSG_MAX;
            if (sbytes[to] > so)
                bad |= rc(t, t->api.Send((const char*)send + soff[to] + so, (size_t)std::min(MSG_MAX, sbytes[to] - so), ncclUint8, to, t->comm, (hipStream_t)stream), "ncclSend");
            if (!bad && rbytes[from] > ro)
                bad |= rc(t, t->api.Recv((char*)recv + roff[from] + ro, (size_t)std::min(MSG_MAX, rbytes[from] - ro), ncclUint8, from, t->comm, (hipStream_t)stream), "ncclRecv");
        }
        if (rc(t, t->api.GroupEnd(), "ncclGroupEnd") || bad) return 1;
    }
    return 0;
}
int rccl_send(void* self, const void* buf, uint64_t bytes, int32_t peer, void* stream) {
    RcclTransport* t = (RcclTransport*)self;
    if (t->dead.load()) return 1;
    for (uint64_t o = 0; o < bytes; o += MSG_MAX)
        if (rc(t, t->api.Send((const char*)buf + o, (size_t)std::min(MSG_MAX, bytes - o), ncclUint8, peer, t->comm, (hipStream_t)stream), "ncclSend")) return 1;
    return 0;
}
int rccl_recv(void* self, void* buf, uint64_t bytes, int32_t peer, void* stream) {
    RcclTransport* t = (RcclTransport*)self;
    if (t->dead.load()) return 1;
    for (uint64_t o = 0; o < bytes; o += MSG_MAX)
        if (rc(t, t->api.Recv((char*)buf + o, (size_t)std::min(MSG_MAX, bytes - o), ncclUint8, peer, t->comm, (hipStream_t)stream), "ncclRecv")) return 1;
    return 0;
}

// 0 = healthy.  ncclCommGetAsyncError reports errors RCCL's proxy / network threads met after an operation was queued (a peer
// that died, a link that failed): the library polls it while it waits for communication.
int rccl_poll(void* self) {
    RcclTransport* t = (RcclTransport*)self;
    if (t->dead.load()) return 1;
    if (!t->api.CommGetAsyncError) return 0;
    ncclResult_t st = ncclSuccess;
    const ncclResult_t r = t->api.CommGetAsyncError(t->comm, &st);
    if (r != ncclSuccess || (st != ncclSuccess && st != ncclInProgress)) {
        t->last = std::string("ncclCommGetAsyncError: ") + t->api.GetErrorString(r != ncclSuccess ? r : st);
        fprintf(stderr, "[dbg transport rccl] %s\n", t->last.c_str());
        return 1;
    }
    return 0;
}
// ncclCommAbort: frees the communicator and makes the kernels RCCL has in flight on this rank give up, so that a host wait on
// them returns; the peers' operations with this rank then fail (their poll reports it) instead of waiting for ever.  The
// ncclComm_t is gone afterwards: the host must not destroy it again (dbg_rccl_comm_destroy of an aborted communicator is the
// caller's to skip -- dbg_transport_aborted tells).
void rccl_abort(void* self) {
    RcclTransport* t = (RcclTransport*)self;
    if (t->dead.exchange(true)) return;
    if (t->api.CommAbort) (void)t->api.CommAbort(t->comm);
}

}  // namespace

extern "C" int dbg_transport_rccl_create(void* nccl_comm, int32_t rank, int32_t world, const char* librccl_path, dbg_transport** out,
                                         char* err, uint64_t err_len) {
    if (!out) return 1;
    *out = nullptr;
    if (!nccl_comm || world < 1 || rank < 0 || rank >= world) { set_err(err, err_len, "dbg_transport_rccl_create: bad communicator / rank / world"); return 1; }
    RcclApi api;
    std::string why;
    if (!load_rccl(librccl_path, &api, &why)) { set_err(err, err_len, why); return 2; }
    RcclTransport* t = new RcclTransport();
    t->api = api;
    t->comm = (ncclComm_t)nccl_comm;
    t->tab.self = t; t->tab.rank = rank; t->tab.world = world;
    t->tab.all_reduce_u64 = rccl_all_reduce; t->tab.all_gather = rccl_all_gather; t->tab.all_to_allv = rccl_all_to_allv;
    t->tab.send = rccl_send; t->tab.recv = rccl_recv;
    t->tab.poll = rccl_poll; t->tab.abort = rccl_abort;
    *out = &t->tab;
    return 0;
}



extern "C" int dbg_rccl_unique_id(const char* librccl_path, uint8_t* id_out, char* err, uint64_t err_len) {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes in the header this file documents");
    RcclApi api;
    std::string why;
    if (!id_out) return 1;
    if (!load_rccl(librccl_path, &api, &why)) { set_err(err, err_len, why); return 2; }
    ncclUniqueId id;
    ncclResult_t r = api.GetUniqueId(&id);
    if (r != ncclSuccess) { set_err(err, err_len, std::string("ncclGetUniqueId: ") + api.GetErrorString(r)); return 3; }
    memcpy(id_out, &id, 128);
    return 0;
}

extern "C" int dbg_rccl_comm_create(const char* librccl_path, const uint8_t* id_128, int32_t world, int32_t rank, int32_t device,
                                    void** comm_out, char* err, uint64_t err_len) {
    RcclApi api;
    std::string why;
    if (!id_128 || !comm_out) return 1;
    *comm_out = nullptr;
    if (!load_rccl(librccl_path, &api, &why)) { set_err(err, err_len, why); return 2; }
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); set_err(err, err_len, "hipSetDevice failed"); return 3; }
    ncclUniqueId id;
    memcpy(&id, id_128, 128);
    ncclComm_t comm = nullptr;
    ncclResult_t r = api.CommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess) { set_err(err, err_len, std::string("ncclCommInitRank: ") + api.GetErrorString(r)); return 4; }
    *comm_out = (void*)comm;
    return 0;
}

extern "C" int dbg_rccl_comm_destroy(const char* librccl_path, void* comm) {
    RcclApi api;
    std::string why;
    if (!comm) return 0;
    if (!load_rccl(librccl_path, &api, &why)) return 2;
    return api.CommDestroy((ncclComm_t)comm) == ncclSuccess ? 0 : 3;
}

// ---- in-process transport: the ranks are THREADS of one process, one per GPU (or several sharing a GPU) -----------------------
// A host that drives all GPUs of a node from one process needs no RCCL: device buffers of all ranks live in one address space, so
// the variable all-to-all is a set of hipMemcpyAsync device-to-device copies (peer copies over xGMI between different GPUs) issued by
// the RECEIVING rank on its own stream, between two barriers of the ranks' threads.  Small reductions go through a shared host
// array.  Synchronous in the sense of the contract (an operation drains the caller's stream, moves the data, returns).  Also what
// tests/cpp/test_shard_threads.cpp runs the rank-spanning entry points with: N ranks on the one GPU of the test box, no Python.
#include <condition_variable>
#include <thread>
namespace {
struct InprocShared {
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    bool broken = false;
    // what every rank publishes for the operation in flight
    std::vector<const void*> ptr;                  // send buffer / reduction input (device)
    std::vector<std::vector<uint64_t>> off, bytes; // all_to_allv: per destination
    std::vector<std::vector<uint64_t>> host;       // all_reduce / all_gather staging (host)
    // point to point: mailbox[to][from]
    struct Mail { const void* p = nullptr; uint64_t bytes = 0; bool full = false, taken = false; };
    std::vector<std::vector<Mail>> mail;
    int refs = 0;
    int timeout_s = 300;
    void fail() { std::lock_guard<std::mutex> lk(mu); broken = true; cv.notify_all(); }
    bool barrier() {
        std::unique_lock<std::mutex> lk(mu);
        if (broken) return false;
        const uint64_t gen = generation;
        if (++arrived == world) { arrived = 0; generation++; cv.notify_all(); return true; }
        // a rank that never arrives (its entry point failed) must not hang the others for ever
        if (!cv.wait_for(lk, std::chrono::seconds(timeout_s), [&] { return generation != gen || broken; })) { broken = true; cv.notify_all(); return false; }
        return !broken;
    }
};
struct InprocTransport {
    dbg_transport tab;
    InprocShared* sh;
};

int ip_sync(void* stream) { return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? 0 : 1; }

int ip_all_reduce(void* self, uint64_t* buf, uint64_t n, int32_t op, void* stream) {
    InprocTransport* t = (InprocTransport*)self;
    InprocShared* sh = t->sh;
    const int me = t->tab.rank, W = sh->world;
    std::vector<uint64_t>& mine = sh->host[me];
    mine.resize(n);
    if (n && hipMemcpyAsync(mine.data(), buf, n * 8, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) { (void)hipGetLastError(); sh->fail(); return 1; }
    if (ip_sync(stream)) { sh->fail(); return 1; }
    if (!sh->barrier()) return 1;
    std::vector<uint64_t> acc(n, 0);
    for (int r = 0; r < W; r++) {
        const std::vector<uint64_t>& v = sh->host[r];
        if (v.size() != n) { sh->fail(); return 1; }                    // the ranks disagree about the operation
        for (uint64_t i = 0; i < n; i++) acc[i] = op == 1 ? std::max(acc[i], v[i]) : acc[i] + v[i];
    }
    if (!sh->barrier()) return 1;                                       // everybody has read the slots
    if (n && hipMemcpyAsync(buf, acc.data(), n * 8, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) { (void)hipGetLastError(); sh->fail(); return 1; }
    if (ip_sync(stream)) { sh->fail(); return 1; }                      // `acc` leaves scope
    return 0;
}
int ip_all_gather(void* self, const void* send, void* recv, uint64_t bytes, void* stream) {
    InprocTransport* t = (InprocTransport*)self;
    InprocShared* sh = t->sh;
    const int me = t->tab.rank, W = sh->world;
    sh->ptr[me] = send;
    if (ip_sync(stream)) { sh->fail(); return 1; }
    if (!sh->barrier()) return 1;
    for (int r = 0; r < W; r++)
        if (bytes && hipMemcpyAsync((char*)recv + (uint64_t)r * bytes, sh->ptr[r], bytes, hipMemcpyDefault, (hipStream_t)stream) != hipSuccess) { (void)hipGetLastError(); sh->fail(); return 1; }
    if (ip_sync(stream)) { sh->fail(); return 1; }
    if (!sh->barrier()) return 1;                                       // the senders' buffers may change again
    return 0;
}
int ip_all_to_allv(void* self, const void* send, const uint64_t* soff, const uint64_t* sbytes, void* recv, const uint64_t* roff,
                   const uint64_t* rbytes, void* stream) {
    InprocTransport* t = (InprocTransport*)self;
    InprocShared* sh = t->sh;
    const int me = t->tab.rank, W = sh->world;
    sh->ptr[me] = send;
    sh->off[me].assign(soff, soff + W);
    sh->bytes[me].assign(sbytes, sbytes + W);
    if (ip_sync(stream)) { sh->fail(); return 1; }
    if (!sh->barrier()) return 1;
    for (int i = 0; i < W; i++) {
        const int s = (me + i) % W;
        if (sh->bytes[s][me] != rbytes[s]) { sh->fail(); return 1; }    // sender and receiver disagree about the message
        if (rbytes[s] && hipMemcpyAsync((char*)recv + roff[s], (const char*)sh->ptr[s] + sh->off[s][me], rbytes[s], hipMemcpyDefault, (hipStream_t)stream) != hipSuccess) {
            (void)hipGetLastError(); sh->fail(); return 1;
        }
    }
    if (ip_sync(stream)) { sh->fail(); return 1; }
    if (!sh->barrier()) return 1;
    return 0;
}
int ip_send(void* self, const void* buf, uint64_t bytes, int32_t peer, void* stream) {
    InprocTransport* t = (InprocTransport*)self;
    InprocShared* sh = t->sh;
    if (peer < 0 || peer >= sh->world) return 1;
    if (ip_sync(stream)) { sh->fail(); return 1; }
    InprocShared::Mail& m = sh->mail[peer][t->tab.rank];
    std::unique_lock<std::mutex> lk(sh->mu);
    if (!sh->cv.wait_for(lk, std::chrono::seconds(sh->timeout_s), [&] { return !m.full || sh->broken; }) || sh->broken) { sh->broken = true; sh->cv.notify_all(); return 1; }
    m.p = buf; m.bytes = bytes; m.full = true; m.taken = false;
    sh->cv.notify_all();
    if (!sh->cv.wait_for(lk, std::chrono::seconds(sh->timeout_s), [&] { return m.taken || sh->broken; }) || sh->broken) { sh->broken = true; sh->cv.notify_all(); return 1; }   // the receiver has copied
    m.full = false;
    sh->cv.notify_all();
    return 0;
}
int ip_recv(void* self, void* buf, uint64_t bytes, int32_t peer, void* stream) {
    InprocTransport* t = (InprocTransport*)self;
    InprocShared* sh = t->sh;
    if (peer < 0 || peer >= sh->world) return 1;
    InprocShared::Mail& m = sh->mail[t->tab.rank][peer];
    const void* src = nullptr;
    {
        std::unique_lock<std::mutex> lk(sh->mu);
        if (!sh->cv.wait_for(lk, std::chrono::seconds(sh->timeout_s), [&] { return (m.full && !m.taken) || sh->broken; }) || sh->broken) { sh->broken = true; sh->cv.notify_all(); return 1; }
        if (m.bytes != bytes) { sh->broken = true; sh->cv.notify_all(); return 1; }
        src = m.p;
    }
    int rc = 0;
    if (bytes && hipMemcpyAsync(buf, src, bytes, hipMemcpyDefault, (hipStream_t)stream) != hipSuccess) rc = 1;
    if (!rc) rc = ip_sync(stream);
    std::unique_lock<std::mutex> lk(sh->mu);
    m.taken = true;
    if (rc) sh->broken = true;
    sh->cv.notify_all();
    return rc;
}
int ip_poll(void* self) {
    InprocShared* sh = ((InprocTransport*)self)->sh;
    std::lock_guard<std::mutex> lk(sh->mu);
    return sh->broken ? 1 : 0;
}
void ip_abort(void* self) { ((InprocTransport*)self)->sh->fail(); }
std::mutex g_inproc_mu;
}  // namespace

extern "C" int dbg_transport_inprocess_create(int32_t world, dbg_transport** out /* [world] */) {
    if (world < 1 || world > 64 || !out) return 1;
    InprocShared* sh = new InprocShared();
    sh->world = world;
    sh->ptr.assign(world, nullptr);
    sh->off.assign(world, {}); sh->bytes.assign(world, {}); sh->host.assign(world, {});
    sh->mail.assign(world, std::vector<InprocShared::Mail>(world));
    sh->refs = world;
    if (const char* e = getenv("DBG_INPROC_TIMEOUT_S")) { const int v = atoi(e); if (v > 0) sh->timeout_s = v; }    // (read once, at creation)
    for (int r = 0; r < world; r++) {
        InprocTransport* t = new InprocTransport();
        t->sh = sh;
        t->tab.self = t; t->tab.rank = r; t->tab.world = world;
        t->tab.all_reduce_u64 = ip_all_reduce; t->tab.all_gather = ip_all_gather; t->tab.all_to_allv = ip_all_to_allv;
        t->tab.send = ip_send; t->tab.recv = ip_recv;
        t->tab.poll = ip_poll; t->tab.abort = ip_abort;
        out[r] = &t->tab;
    }
    return 0;
}

// 1 once a table of this library has been aborted / broken (RCCL: the ncclComm_t no longer exists -- do not destroy it again)
extern "C" int dbg_transport_aborted(const dbg_transport* t) {
    if (!t || t->self != (const void*)t) return 0;
    if (t->all_reduce_u64 == rccl_all_reduce) return ((RcclTransport*)t->self)->dead.load() ? 1 : 0;
    if (t->all_reduce_u64 == ip_all_reduce) return ip_poll(t->self);
    return 0;
}

extern "C" void dbg_transport_destroy(dbg_transport* t) {
    if (!t || t->self != (void*)t) return;                  // not a table this library made
    if (t->all_reduce_u64 == rccl_all_reduce) { delete (RcclTransport*)t->self; return; }
    if (t->all_reduce_u64 == ip_all_reduce) {
        InprocTransport* it = (InprocTransport*)t->self;
        InprocShared* sh = it->sh;
        bool last;
        { std::lock_guard<std::mutex> g(g_inproc_mu); last = --sh->refs == 0; }
        delete it;
        if (last) delete sh;
    }
}

// ---- exchange geometry: pure host arithmetic, identical on every rank ---------------------------------------------------
extern "C" int dbg_shard_owner_bounds(const uint64_t* group_records, uint32_t n_bins, uint32_t bin_group, uint32_t world, uint32_t* bounds) {
    if (!bounds || !world || !bin_group || n_bins % bin_group) return 1;
    const uint32_t ng = n_bins / bin_group;
    bounds[0] = 0;
    if (!group_records) {
        for (uint32_t r = 1; r <= world; r++) bounds[r] = (uint32_t)((uint64_t)r * ng / world) * bin_group;
        return 0;
    }
    // Greedy cut of the cumulative record histogram: owner r ends at the first group boundary where the running total reaches
    // r/world of all records (nearest of the two boundaries around the target), never before its predecessor's end.  Every owner
    // keeps at least one group while groups remain, so that a rank is never handed an empty range by rounding alone.
    unsigned __int128 total = 0;
    for (uint32_t g = 0; g < ng; g++) total += group_records[g];
    uint64_t run = 0;
    uint32_t g = 0;
    for (uint32_t r = 1; r < world; r++) {
        const uint64_t target = (uint64_t)(total * r / world);
        while (g < ng && run + group_records[g] <= target) run += group_records[g++];
        // g is the first group whose end lies beyond the target: cut before or after it, whichever is nearer
        uint32_t cut = g;
        if (g < ng && target - run > run + group_records[g] - target) { run += group_records[g++]; cut = g; }
        const uint32_t lo = bounds[r - 1] / bin_group;
        const uint32_t left_for_rest = world - r;                      // owners after r, each wants a group if there is one
        if (cut <= lo && lo < ng) cut = lo + 1;
        if (ng >= world && cut + left_for_rest > ng) cut = ng - left_for_rest;
        if (cut < lo) cut = lo;
        if (cut > ng) cut = ng;
        while (g < cut) run += group_records[g++];
        while (g > cut) run -= group_records[--g];
        bounds[r] = cut * bin_group;
    }
    bounds[world] = n_bins;
    return 0;
}

extern "C" int dbg_shard_round_cuts(const uint32_t* bounds, uint32_t world, uint32_t bin_group, uint32_t* n_rounds_io, uint32_t* cuts) {
    if (!bounds || !n_rounds_io || !cuts || !world || !bin_group) return 1;
    const uint32_t stride = *n_rounds_io + 1;
    uint32_t smallest = ~0u;
    for (uint32_t d = 0; d < world; d++) smallest = std::min(smallest, (bounds[d + 1] - bounds[d]) / bin_group);
    uint32_t nr = std::max<uint32_t>(1, std::min<uint32_t>(*n_rounds_io, std::max<uint32_t>(smallest, 1)));
    for (uint32_t d = 0; d < world; d++) {
        const uint64_t ng = (bounds[d + 1] - bounds[d]) / bin_group;
        for (uint32_t c = 0; c <= nr; c++) cuts[(size_t)d * stride + c] = (uint32_t)(c * ng / nr) * bin_group;
        cuts[(size_t)d * stride + nr] = bounds[d + 1] - bounds[d];
    }
    *n_rounds_io = nr;
    return 0;
}
