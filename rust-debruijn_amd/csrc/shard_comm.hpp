// Failure agreement and bounded waits of the rank-spanning entry points (dbg_shard_filter_kmers_dev, dbg_shard_compress_dev).
//
// The reference is one process: a failure anywhere is a panic that unwinds all of it (src/filter.rs:167, src/graph.rs:87-91).
// Across processes the equivalent is "every rank fails together", and nothing gives that for free: a rank that runs out of
// memory and returns while its peers are inside ncclRecv leaves them there for ever.  The rules here:
//   * a rank-local failure (allocation, kernel, bad label) never makes a rank skip a collective: the phase's local work runs to
//     its end with the status kept, every phase ends in agree() -- a one-word MAX all-reduce of the status -- and on a non-zero
//     result EVERY rank returns the same error code before the phase's data moves;
//   * a failure of the transport itself (an operation returns non-zero, the communicator reports an asynchronous error, a wait
//     exceeds DBG_COMM_TIMEOUT_S) cannot be agreed on: the rank calls the transport's abort (RCCL: ncclCommAbort) so that peers
//     blocked in it fail instead of hang, and returns;
//   * the host never waits unboundedly on a stream that holds communication: waits poll the stream / event, the transport's
//     health and a deadline.
// DBG_FAIL_AT=<site>[:<rank>] (ctx option) injects a local failure at a named site: tests/test_gpu_shard_faults.py.
#pragma once
#include "dbg_ctx.hpp"
#include <chrono>
#include <thread>

namespace {

struct ShardComm {
    dbg_ctx* c;
    const dbg_transport* tr;
    uint32_t W, me;
    bool live;                          // a transport with peers (or a forced one-rank exchange): agreement and bounded waits apply
    double timeout_s = 300.0;
    std::string fail_site;
    int fail_rank = -1;
    DBuf<uint64_t> word;                // device words of the status all-reduce, reserved before anything can fail
    static constexpr uint32_t WORDS = 8;

    ShardComm(dbg_ctx* c_, const dbg_transport* tr_, bool forced = false)
        : c(c_), tr(tr_), W(tr_ ? (uint32_t)tr_->world : 1u), me(tr_ ? (uint32_t)tr_->rank : 0u), live(tr_ && (tr_->world > 1 || forced)) {
        if (const char* e = c->opt("DBG_COMM_TIMEOUT_S")) { const double v = atof(e); if (v > 0.0) timeout_s = v; }
        if (const char* e = c->opt("DBG_FAIL_AT")) {
            std::string s(e);
            const size_t colon = s.find(':');
            fail_site = s.substr(0, colon);
            if (colon != std::string::npos) fail_rank = atoi(s.c_str() + colon + 1);
        }
    }
    // what every later agree() needs; a failure here cannot be communicated (abort, so that the peers do not wait for it)
    int prepare() {
        if (!live) return 0;
        if (!word.alloc(c, WORDS)) { abort(); return c->fail(101, "rank-spanning call: no device memory for the status word"); }
        return 0;
    }
    // fault injection: true when this rank is to fail at `site`
    bool inject(const char* site) const { return !fail_site.empty() && fail_site == site && (fail_rank < 0 || (uint32_t)fail_rank == me); }
    int injected(const char* site) { return c->fail(169, std::string("injected failure at '") + site + "' (DBG_FAIL_AT)"); }

    void abort() { if (tr && tr->abort) tr->abort(tr->self); }
    bool unhealthy() const { return tr && tr->poll && tr->poll(tr->self) != 0; }
    int op_failed(const char* op) {
        abort();
        return c->fail(160, std::string("rank-spanning call: transport operation ") + op + " failed (communicator aborted)");
    }

    template <class Q>
    int wait_ready(Q query, const char* what) {
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t spins = 0;; spins++) {
            const hipError_t e = query();
            if (e == hipSuccess) return 0;
            (void)hipGetLastError();
            if (e != hipErrorNotReady) { abort(); return c->fail(100, std::string("HIP error ") + hipGetErrorString(e) + " while waiting for " + what); }
            if ((spins & 15u) == 15u) {
                if (unhealthy()) { abort(); return c->fail(167, std::string("the communicator reported an error while this rank waited for ") + what + " (communicator aborted)"); }
                const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (el > timeout_s) {
                    abort();
                    char b[256];
                    snprintf(b, sizeof(b), "timed out after %.0f s waiting for %s: a peer rank has probably failed (communicator aborted; DBG_COMM_TIMEOUT_S)", el, what);
                    return c->fail(168, b);
                }
            }
            if (spins > 256) std::this_thread::sleep_for(std::chrono::microseconds(spins > 4096 ? 200 : 20));
        }
    }
    // hipStreamSynchronize / hipEventSynchronize with a deadline and a health check (plain waits when there is nothing to watch)
    int wait_stream(hipStream_t s, const char* what) {
        if (!live) { HIP_TRY(c, hipStreamSynchronize(s)); return 0; }
        return wait_ready([&] { return hipStreamQuery(s); }, what);
    }
    int wait_event(hipEvent_t ev, const char* what) {
        if (!live) { HIP_TRY(c, hipEventSynchronize(ev)); return 0; }
        return wait_ready([&] { return hipEventQuery(ev); }, what);
    }

    // MAX all-reduce of a few host values over the ranks, ordered on the ctx stream
    int reduce(uint64_t* vals, uint32_t n, int op, const char* what) {
        if (!live) return 0;                // (a forced one-rank exchange goes through the transport like any other: its calls are the point)
        DBuf<uint64_t> big;
        uint64_t* d = word.p;
        if (n > WORDS) { if (!big.alloc(c, n)) { abort(); return c->fail(101, std::string("rank-spanning call: no device memory for ") + what); } d = big.p; }
        if (hipMemcpyAsync(d, vals, (size_t)n * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess) { (void)hipGetLastError(); return op_failed(what); }
        if (tr->all_reduce_u64(tr->self, d, n, op, c->stream)) return op_failed(what);
        if (hipMemcpyAsync(vals, d, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { (void)hipGetLastError(); return op_failed(what); }
        return wait_stream(c->stream, what);
    }

    // End of a phase: every rank brings its local status; all leave with the same verdict.  `extra` (may be NULL): n_extra more
    // values max-reduced in the same message (what the phase has to agree on anyway).  Returns 0, or the largest error code of
    // any rank -- on every rank -- with dbg_last_error naming the phase and the rank.
    int agree(int local_rc, const char* phase, uint64_t* extra = nullptr, uint32_t n_extra = 0) {
        if (!live) return local_rc;
        uint64_t v[WORDS] = {0};
        if (n_extra + 1 > WORDS) return c->fail(10, "agree: too many values");
        v[0] = local_rc ? (((uint64_t)(uint32_t)local_rc << 16) | (uint64_t)(me + 1)) : 0;
        for (uint32_t i = 0; i < n_extra; i++) v[1 + i] = extra[i];
        const std::string mine = local_rc ? c->err : std::string();
        DBG_TRY(reduce(v, 1 + n_extra, 1, "status agreement"));
        for (uint32_t i = 0; i < n_extra; i++) extra[i] = v[1 + i];
        if (!v[0]) return 0;
        const int code = (int)(v[0] >> 16);
        const uint32_t who = (uint32_t)(v[0] & 0xffffu) - 1u;
        char b[200];
        if (local_rc) {
            snprintf(b, sizeof(b), " [phase '%s', rank %u (local error %d); all %u ranks return error %d]", phase, me, local_rc, W, code);
            c->err = mine + b;
        } else {
            snprintf(b, sizeof(b), "rank %u failed in phase '%s' of the rank-spanning call with error %d; all %u ranks return it (this rank, %u, had no error of its own)",
                     who, phase, code, W, me);
            c->err = b;
        }
        return code;
    }
};

}  // namespace
