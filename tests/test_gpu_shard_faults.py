"""GPU: a rank-local failure inside the rank-spanning entry points makes ALL ranks fail together, quickly, and leaves ctx and
transport usable (round 5; include/dbg_mi355x.h "Failures", rust-debruijn_amd/csrc/shard_comm.hpp).

The reference is one process: a failure is a panic that unwinds all of it (src/filter.rs:167, src/graph.rs:87-91).  Across ranks
the equivalent is "every rank returns the same error"; what it must never be is one rank returning while its peers wait in a
collective for ever.  The ranks here are threads of this process on the one GPU (in-process transport); DBG_FAIL_AT=<site>:<rank>
makes one rank fail at a named site of the flow -- before the first collective, between collectives, inside the exchange rounds,
after the last collective -- and every rank must come back with the same non-zero code within seconds; the same group and the same
contexts must then run the call again and produce the oracle's table / graph.  A transport that breaks (abort) is the other case:
every rank fails, nobody hangs, the group stays dead."""
import ctypes as C
import threading
import time

import numpy as np
import pytest

import oracle_lib as O
import refgen as R
from graph_canon import canonical_nodes, graphs_equal
from pkg import dbg, capi, D

pytestmark = pytest.mark.gpu

FILTER_SITES = ["count", "scan", "hist", "layout", "counts", "tables", "recs", "rbuf", "count_begin", "round", "finish"]
COMPRESS_SITES_GATHER = ["shard", "gather", "merge"]
COMPRESS_SITES_TREE = ["shard", "tree"]


def make_group(world):
    lib = capi.load()
    arr = (C.POINTER(capi.Transport) * world)()
    assert lib.dbg_transport_inprocess_create(world, arr) == 0
    return arr


class _Tr:
    def __init__(self, ptr, rank):
        self.ptr, self.rank = ptr, rank


def run_ranks(world, fn, timeout=120.0):
    out, err = [None] * world, [None] * world

    def body(r):
        try:
            out[r] = fn(r)
        except BaseException as e:                                # noqa: BLE001 -- reported below
            err[r] = e
    th = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(world)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join(max(0.1, timeout - (time.perf_counter() - t0)))
    hung = [r for r, t in enumerate(th) if t.is_alive()]
    assert not hung, "ranks %s are still inside the call after %.0f s: a peer's failure left them waiting" % (hung, timeout)
    for r, e in enumerate(err):
        if e is not None:
            raise AssertionError("rank %d: %r" % (r, e))
    return out, time.perf_counter() - t0


def make_reads(seed, k, n_reads=900, kind=0, n_labels=5):
    rng = np.random.default_rng(seed)
    genome = R.random_dna(rng, 6000)
    seqs = []
    for _ in range(n_reads):
        ln = int(rng.integers(k, 200))
        st = int(rng.integers(0, len(genome) - ln + 1))
        s = genome[st:st + ln].copy()
        m = rng.random(ln) < 0.004
        s[m] = (s[m] + rng.integers(1, 4, size=int(m.sum()))) % 4
        if rng.random() < 0.5:
            s = R.revcomp_bytes(s)
        seqs.append(s.astype(np.uint8))
    data = rng.integers(0, n_labels, size=n_reads) if kind else None
    return seqs, data


class Job:
    """world contexts + one in-process group, kept across calls (the point: they stay usable after an agreed failure)"""

    def __init__(self, world, k, kind, seed, n_labels=5, n_reads=900, lists=True, wide=False):
        self.world, self.k, self.kind = world, k, kind
        self.lib = capi.load()
        self.keep = make_group(world)
        self.trs = [_Tr(self.keep[r], r) for r in range(world)]
        # DBG_TEST_PER_DEVICE=1 (tools/first_contact_8gpu.sh on a multi-GPU node): thread-rank r drives device r, so that the in-process
        # transport's copies are real peer copies
        import os
        import torch
        nd = torch.cuda.device_count() if os.environ.get("DBG_TEST_PER_DEVICE") else 1
        self.ctxs = [dbg.Context(r % max(nd, 1)) for r in range(world)]
        for c in self.ctxs:
            c.set_option("DBG_COMPRESS", "device")
            if not lists:                                             # alphabets beyond 64 colours: label groups / the key-range route instead of label lists
                c.set_option("DBG_LABEL_LISTS", "0")
        self.seqs, self.data = make_reads(seed, k, n_reads=n_reads, kind=kind, n_labels=n_labels)
        w = self.width = 4 if wide else 2
        if wide and kind:                                           # full-width u32 labels: the job-wide label dictionary (round 6)
            self.data = (self.data.astype(np.uint64) * 1000003 + (1 << 24)) % (1 << 32)
        n = len(self.seqs)
        self.bounds = [n * r // world for r in range(world + 1)]
        self.devs = []
        for r in range(world):
            lo, hi = self.bounds[r], self.bounds[r + 1]
            ss_o = O.SeqSet.from_byte_seqs(self.seqs[lo:hi], data=(self.data[lo:hi] if kind else None), sizeof_d1=w if kind else 0)
            hs = dbg.HostSeqs(ss_o.words, ss_o.start, ss_o.length, None, ss_o.data if kind else None, w if kind else 0)
            dev, hc = capi.SeqSet(), hs.c_struct()
            self.ctxs[r].check(self.lib.dbg_seqset_to_device(self.ctxs[r].h, C.byref(hc), C.byref(dev)))
            self.devs.append(dev)

    def set_fault(self, spec):
        for c in self.ctxs:
            c.set_option("DBG_FAIL_AT", spec)

    def filter_raw(self, r, rounds=3):
        """-> (return code, table, error text): the C call without the mirror's exception"""
        ctx = self.ctxs[r]
        p = capi.ShardParams(self.k, 0, self.kind, 2, rounds, 0, 1, 0)
        tab, st = capi.KmerTable(), capi.ShardStats()
        rc = self.lib.dbg_shard_filter_kmers_dev(ctx.h, self.trs[r].ptr, C.byref(self.devs[r]), C.byref(p), C.byref(tab), C.byref(st))
        return rc, tab, self.lib.dbg_last_error(ctx.h).decode()

    def compress_raw(self, r, tab, reduce, root=0):
        ctx = self.ctxs[r]
        spec = dbg.ScmapCompress() if self.kind else dbg.SimpleCompress("saturating_add")
        fin, loc, cl = capi.Graph(), capi.Graph(), capi.LabelClasses()
        rc = self.lib.dbg_shard_compress_dev(ctx.h, self.trs[r].ptr, self.k, 0, spec.kind, spec.kind, C.byref(tab), reduce, root,
                                             C.byref(fin), C.byref(loc), C.byref(cl))
        err = self.lib.dbg_last_error(ctx.h).decode()
        g = None
        if rc == 0:
            if r == root:
                g = dbg._graph_from_c(ctx, fin, self.k)
            else:
                self.lib.dbg_free_graph(ctx.h, C.byref(fin))
            self.lib.dbg_free_graph(ctx.h, C.byref(loc))
            self.lib.dbg_free_label_classes(C.byref(cl))
        return rc, g, err

    def oracle_table(self):
        ss = O.SeqSet.from_byte_seqs(self.seqs, data=self.data if self.kind else None, sizeof_d1=self.width if self.kind else 0)
        return O.filter_kmers(ss, self.k, O.COUNT_FILTER_SET if self.kind else O.COUNT_FILTER, 2, stranded=False)

    def check_tables(self, tabs):
        want = self.oracle_table()
        hi, lo, cnt, sets = [], [], [], []
        for r, t in enumerate(tabs):
            h = capi.KmerTable()
            self.ctxs[r].check(self.lib.dbg_table_to_host(self.ctxs[r].h, C.byref(t), C.byref(h)))
            th = dbg._table_from_c(h, self.k)
            self.lib.dbg_free_table(self.ctxs[r].h, C.byref(h))
            hi.append(th.key_hi); lo.append(th.key_lo)
            if not self.kind:
                cnt.append(th.count)
            else:
                sets += [tuple(int(x) for x in th.set_val[int(th.set_off[i]):int(th.set_off[i + 1])]) for i in range(len(th.key_lo))]
        hi, lo = np.concatenate(hi), np.concatenate(lo)
        o = np.lexsort((lo, hi))
        assert len(hi) == want.n and np.array_equal(hi[o], want.key_hi) and np.array_equal(lo[o], want.key_lo)
        if not self.kind:
            assert np.array_equal(np.concatenate(cnt)[o], want.count)
        else:
            assert [sets[i] for i in o] == [tuple(int(x) for x in want.set_val[int(want.set_off[i]):int(want.set_off[i + 1])]) for i in range(want.n)]

    def close(self):
        for r, c in enumerate(self.ctxs):
            self.lib.dbg_seqset_free_device(c.h, C.byref(self.devs[r]))
            c.close()
        for r in range(self.world):
            self.lib.dbg_transport_destroy(self.keep[r])


@pytest.mark.parametrize("world,k,kind,n_labels,lists", [(3, 31, 0, 5, True), (2, 47, 1, 5, True), (3, 33, 1, 150, False), (2, 12, 1, 5, True), (3, 40, 1, 60000, False),
                                                         (3, 33, 1, 150, True), (2, 40, 1, 60000, True), (3, 47, 1, 900, "wide")])
def test_injected_failure_in_shard_filter_fails_all_ranks_together(world, k, kind, n_labels, lists):
    """(shapes 3-5: label groups -- 150 distinct labels with DBG_LABEL_LISTS=0, one CountFilter run + three group runs, failures in the
    rank-local steps between the runs included -- and the key-range route for k < 16 and, with DBG_LABEL_LISTS=0, for > 1024 distinct
    labels; the last two: label lists, the default beyond 64 colours -- the sites of the super-k-mer exchange, with bin_labels_kernel
    counting the rounds)"""
    job = Job(world, k, kind, seed=5100 + world, n_labels=n_labels, n_reads=2600 if n_labels > 1024 else 900, lists=bool(lists), wide=lists == "wide")
    sites = FILTER_SITES
    if lists == "wide":                                              # labels of 2^24 and more: the dictionary phase, then the label-list run on the ranks
        sites = ["count", "labeldict", "labels", "scan", "round", "finish"]
    elif n_labels > 64 and lists:
        sites = FILTER_SITES + ["labels"]
    elif n_labels == 150:
        sites = ["count", "labels", "scan", "groups", "round", "join", "finish"]
    elif k < 16 or n_labels > 1024:
        sites = ["count", "histogram", "extract", "reduce"]
    try:
        for i, site in enumerate(sites):
            victim = i % world
            job.set_fault("%s:%d" % (site, victim))
            res, secs = run_ranks(world, lambda r: job.filter_raw(r), timeout=60.0)
            codes = [x[0] for x in res]
            assert all(cd != 0 for cd in codes), "site %s: return codes %s -- a rank succeeded although rank %d failed" % (site, codes, victim)
            assert len(set(codes)) == 1, "site %s: the ranks return different codes %s" % (site, codes)
            assert secs < 30.0, "site %s: %.1f s -- the ranks waited for something" % (site, secs)
            assert "injected failure" in res[victim][2] and site in res[victim][2], res[victim][2]
            for r in range(world):
                if r != victim:
                    assert "rank %d failed" % victim in res[r][2], (site, r, res[r][2])
                assert res[r][1].n == 0 and not res[r][1].key_lo, "site %s: rank %d got a table out of a failed call" % (site, r)
            # ... and the same contexts, the same group, no fault: the oracle's table
            job.set_fault("")
            res, _ = run_ranks(world, lambda r: job.filter_raw(r), timeout=60.0)
            assert [x[0] for x in res] == [0] * world, (site, [x[2] for x in res])
            job.check_tables([x[1] for x in res])
            for r in range(world):
                job.lib.dbg_free_table(job.ctxs[r].h, C.byref(res[r][1]))
    finally:
        job.close()


@pytest.mark.parametrize("reduce,kind", [(0, 0), (1, 0), (0, 1)])
def test_injected_failure_in_shard_compress_fails_all_ranks_together(reduce, kind):
    world, k = 3, 31
    job = Job(world, k, kind, seed=5200 + reduce)
    try:
        job.set_fault("")
        res, _ = run_ranks(world, lambda r: job.filter_raw(r), timeout=60.0)
        assert [x[0] for x in res] == [0] * world, [x[2] for x in res]
        tabs = [x[1] for x in res]
        # the reference flow on the oracle: per-shard compress -> combine -> compress_graph (src/test.rs:459-470)
        want = None
        sites = (COMPRESS_SITES_TREE if reduce else COMPRESS_SITES_GATHER) + (["classes"] if kind else [])
        for i, site in enumerate(sites):
            # (the merge of the gather mode runs on root only; the tree's first-level receivers are the even positions)
            victim = 0 if site in ("gather", "merge", "tree") else (i + 1) % world
            job.set_fault("%s:%d" % (site, victim))
            out, secs = run_ranks(world, lambda r: job.compress_raw(r, tabs[r], reduce), timeout=60.0)
            codes = [x[0] for x in out]
            assert all(cd != 0 for cd in codes) and len(set(codes)) == 1, "site %s: return codes %s" % (site, codes)
            assert secs < 30.0, "site %s: %.1f s" % (site, secs)
            assert "injected failure" in out[victim][2], out[victim][2]
            job.set_fault("")
            out, _ = run_ranks(world, lambda r: job.compress_raw(r, tabs[r], reduce), timeout=60.0)
            assert [x[0] for x in out] == [0] * world, (site, [x[2] for x in out])
            g = out[0][1]
            if want is None:
                want = g                                       # every repetition must give the same graph as the first clean run
            elif reduce:                                       # (tree: the same unitigs; node order and strand may differ)
                assert canonical_nodes(g.arrays(), k, False) == canonical_nodes(want.arrays(), k, False)
            else:
                assert graphs_equal(g.arrays(), want.arrays())
        for r in range(world):
            job.lib.dbg_free_table(job.ctxs[r].h, C.byref(tabs[r]))
    finally:
        job.close()


def test_broken_transport_fails_every_rank_and_nobody_hangs():
    """one rank aborts the group from outside the call (what a host does when it learns that a peer process died): the ranks that
    are inside the collective flow come back with an error instead of waiting out the timeout, and the group stays dead"""
    world, k = 3, 31
    job = Job(world, k, 0, seed=5300)
    try:
        job.set_fault("")
        tbl = job.keep[1].contents
        started = threading.Event()

        def body(r):
            if r == 1:                                            # the "dead" rank never enters; it breaks the group a moment later
                started.wait(10.0)
                time.sleep(0.3)
                tbl.abort(tbl.self)
                return (1, None, "aborted from outside")
            started.set()
            return job.filter_raw(r)
        res, secs = run_ranks(world, body, timeout=60.0)
        assert all(x[0] != 0 for x in res), [x[0] for x in res]
        assert secs < 30.0
        for r in range(world):
            assert job.lib.dbg_transport_aborted(job.keep[r]) == 1
        # a dead group fails fast
        res, secs = run_ranks(world, lambda r: job.filter_raw(r), timeout=60.0)
        assert all(x[0] != 0 for x in res) and secs < 10.0
    finally:
        job.close()


def test_merge_vote_is_collective_and_not_sticky():
    """merge_dups = -1 with three ranks: the decision is a vote of all ranks -- max-reduced, so every rank uses the same setting even
    when their own measurements differ -- and it is re-evaluated every call: having merged once does not pin it to "on"
    (DBG_SHARD_MERGE_COST_MS sets what the merge is taken to cost, so that the outcome does not depend on timing noise)"""
    world, k = 3, 31
    job = Job(world, k, 0, seed=5400)
    try:
        job.set_fault("")

        def call(merge, costs):
            def one(r):
                ctx = job.ctxs[r]
                ctx.set_option("DBG_SHARD_MERGE_COST_MS", costs[r])
                p = capi.ShardParams(k, 0, 0, 2, 2, merge, 1, 0)
                tab, st = capi.KmerTable(), capi.ShardStats()
                rc = job.lib.dbg_shard_filter_kmers_dev(ctx.h, job.trs[r].ptr, C.byref(job.devs[r]), C.byref(p), C.byref(tab), C.byref(st))
                assert rc == 0, job.lib.dbg_last_error(ctx.h).decode()
                job.lib.dbg_free_table(ctx.h, C.byref(tab))
                return int(st.merge_dups)
            res, _ = run_ranks(world, one, timeout=60.0)
            return res
        assert call(-1, ["1e9"] * 3) == [0, 0, 0]                 # first call of these contexts, three ranks: off
        assert call(1, ["1e9"] * 3) == [1, 1, 1]                  # asked for
        assert call(-1, ["1e9"] * 3) == [0, 0, 0]                 # merged last time, and it did not pay: off again (not sticky)
        assert call(-1, ["-1"] * 3) == [1, 1, 1]                  # any exposed time beats a negative cost: on
        assert call(-1, ["1e9", "-1", "1e9"]) == [1, 1, 1]        # one rank votes for it: every rank merges (collective)
        assert call(-1, ["1e9"] * 3) == [0, 0, 0]
    finally:
        job.close()
