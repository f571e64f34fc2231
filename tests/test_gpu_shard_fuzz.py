"""GPU: seeded differential fuzzing of the rank-spanning entry points (dbg_shard_filter_kmers_dev, dbg_shard_compress_dev) against the
CPU oracle.  The ranks are THREADS of this process -- one dbg_ctx each on the one GPU, connected by the library's in-process transport
(dbg_transport_inprocess_create; ctypes releases the GIL inside the calls, so the ranks really run side by side and meet in the
transport's barriers).  Every case draws k (4..64), strandedness, summarizer, min_kmer_obs, the label alphabet (narrow / wide /
sparse / up to 2000 labels up to 2^24 - 1), the number of ranks (1..5), how the reads are split (uneven, a rank may hold nothing), exchange rounds, the sender-side merge
per rank and the ownership rule, and demands:

  * the ranks' tables are disjoint and their union, merged by key, is the oracle's filter_kmers over ALL reads, row for row
    (src/filter.rs:139-231; the reference's scale-out: msp shards + per-shard filter_kmers, src/test.rs:433-456);
  * the gathered graph of dbg_shard_compress_dev equals the oracle's per-table compress_kmers -> combine -> compress_graph
    (src/test.rs:459-470) node for node, label-list classes in the job-wide numbering; the tree merge gives the same canonical graph.
"""
import ctypes as C
import os
import threading

import numpy as np
import pytest

import oracle_lib as O
import refgen as R
from graph_canon import canonical_nodes, graphs_equal
from pkg import dbg, capi, D

pytestmark = pytest.mark.gpu
N_SEEDS = int(os.environ.get("DBG_SHARD_FUZZ_SEEDS", 40))


class _Tr:
    """one rank's table of an in-process transport group"""
    def __init__(self, ptr, rank):
        self.ptr, self.rank = ptr, rank


def make_group(world):
    lib = capi.load()
    arr = (C.POINTER(capi.Transport) * world)()
    assert lib.dbg_transport_inprocess_create(world, arr) == 0
    return [_Tr(arr[r], r) for r in range(world)], arr


def run_ranks(world, fn):
    out, err = [None] * world, [None] * world

    def body(r):
        try:
            out[r] = fn(r)
        except BaseException as e:                                # noqa: BLE001 -- reported below
            err[r] = e
    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for r, e in enumerate(err):
        if e is not None:
            raise AssertionError("rank %d: %r" % (r, e))
    return out


def draw(rng):
    # (round 5: k below 16 takes the key-range route of the rank-spanning call; label alphabets beyond 64 colours / 65535 the label lists,
    #  or label groups / the key-range route with DBG_LABEL_LISTS=0)
    k = int(rng.choice([int(rng.integers(16, 33)), int(rng.integers(33, 49)), int(rng.integers(49, 65)), 47, 63, int(rng.integers(4, 16))]))
    stranded = bool(rng.random() < 0.3)
    kind = int(rng.integers(0, 2))
    world = int(rng.choice([1, 2, 2, 3, 4, 5]))
    genome_len = int(rng.choice([600, 4000, 20000]))
    genome = R.random_dna(rng, genome_len)
    if rng.random() < 0.3:
        a = int(rng.integers(0, genome_len - 150))
        unit = R.random_dna(rng, int(rng.integers(1, 4)))
        genome[a:a + 150] = np.tile(unit, 150)[:150]
    n_reads = int(rng.choice([60, 400, 2500]))
    err = float(rng.choice([0.0, 0.003, 0.02]))
    seqs = []
    for _ in range(n_reads):
        ln = int(rng.integers(0, k + 2)) if rng.random() < 0.08 else int(rng.integers(k, 240))
        ln = min(ln, genome_len)
        st = int(rng.integers(0, genome_len - ln + 1))
        s = genome[st:st + ln].copy()
        m = rng.random(ln) < err
        s[m] = (s[m] + rng.integers(1, 4, size=int(m.sum()))) % 4
        if not stranded and rng.random() < 0.5:
            s = R.revcomp_bytes(s)
        seqs.append(s.astype(np.uint8))
    data, width = None, 0
    if kind:
        width = int(rng.choice([1, 2, 4]))
        alphabet = {0: np.arange(int(rng.integers(1, 24))), 1: np.arange(int(rng.integers(25, 64))),
                    2: np.sort(rng.choice(np.arange(250 if width == 1 else 60000), size=int(rng.integers(2, 60)), replace=False)),
                    # more than 64 distinct labels; with 4-byte labels also values beyond 65535 (up to 2^24 - 1)
                    3: np.sort(rng.choice(np.arange(256 if width == 1 else (65536 if width == 2 else 1 << 24)),
                                          size=int(rng.integers(65, 250 if width == 1 else 2000)), replace=False)),
                    # (round 6) full-width u32 labels: hashes / 32-bit ids, 2^32 - 1 included -- the job-wide label dictionary
                    4: np.unique(np.concatenate([rng.integers(0, 1 << 32, size=int(rng.integers(2, 300)), dtype=np.uint64),
                                                 np.array([(1 << 32) - 1, 1 << 24], dtype=np.uint64)]))}[int(rng.integers(0, 5 if width == 4 else 4))]
        data = alphabet[rng.integers(0, len(alphabet), size=n_reads)]
    # uneven split of the reads over the ranks; now and then a rank holds nothing
    cuts = np.sort(rng.integers(0, n_reads + 1, size=world - 1)) if world > 1 else np.zeros(0, np.int64)
    if world > 1 and rng.random() < 0.2:
        cuts[0] = 0
    bounds = [0] + [int(x) for x in cuts] + [n_reads]
    return dict(k=k, stranded=stranded, kind=kind, world=world, seqs=seqs, data=data, width=width, bounds=bounds,
                min_obs=int(rng.choice([1, 2, 2, 3])), rounds=int(rng.choice([0, 0, 1, 2, 5])), balance=bool(rng.integers(0, 2)),
                merge=[int(rng.choice([-1, 0, 1])) for _ in range(world)], force=bool(world == 1 and rng.random() < 0.7),
                tree=bool(rng.random() < 0.4),
                # alphabets beyond 64 colours: label lists (fast_labellists.hpp) by default, label groups / the key-range route with the knob
                lists_off=bool(rng.random() < 0.3))


def masks_to_classes(off, val):
    """CSR label lists -> (tuple per row)"""
    return [tuple(int(x) for x in val[int(off[i]):int(off[i + 1])]) for i in range(len(off) - 1)]


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_fuzz_rank_spanning_entry_points(seed):
    rng = np.random.default_rng(77000 + seed)
    c = draw(rng)
    k, stranded, kind, world = c["k"], c["stranded"], c["kind"], c["world"]
    trs, keep = make_group(world)
    lib = capi.load()
    spec = dbg.ScmapCompress() if kind else dbg.SimpleCompress("saturating_add")
    spec2 = spec if kind else dbg.SimpleCompress("max")

    def rank_main(r):
        ctx = dbg.Context(0)
        try:
            ctx.set_option("DBG_COMPRESS", "device")
            if c["lists_off"]:
                ctx.set_option("DBG_LABEL_LISTS", "0")
            lo, hi = c["bounds"][r], c["bounds"][r + 1]
            ss_o = O.SeqSet.from_byte_seqs(c["seqs"][lo:hi], data=(c["data"][lo:hi] if kind else None), sizeof_d1=c["width"] if kind else 0)
            hs = dbg.HostSeqs(ss_o.words, ss_o.start, ss_o.length, None, ss_o.data if kind else None, c["width"] if kind else 0)
            dev = capi.SeqSet()
            hc = hs.c_struct()
            ctx.check(lib.dbg_seqset_to_device(ctx.h, C.byref(hc), C.byref(dev)))
            tab, st = D.shard_filter_kmers_c(ctx, trs[r], dev, k, stranded, kind, c["min_obs"], n_rounds=c["rounds"], merge_dups=c["merge"][r],
                                             balance=c["balance"], force_exchange=c["force"])
            lib.dbg_seqset_free_device(ctx.h, C.byref(dev))
            h = capi.KmerTable()
            ctx.check(lib.dbg_table_to_host(ctx.h, C.byref(tab), C.byref(h)))
            th = dbg._table_from_c(h, k)
            lib.dbg_free_table(ctx.h, C.byref(h))
            final, local = D.shard_compress_c(ctx, trs[r], tab, k, stranded, spec, spec2, reduce=1 if c["tree"] else 0, root=world - 1)
            lib.dbg_free_table(ctx.h, C.byref(tab))
            return dict(table=th, final=final, local=local, owned=int(st.records_owned), scanned=int(st.records_scanned))
        finally:
            ctx.close()

    res = run_ranks(world, rank_main)
    for t in trs:
        lib.dbg_transport_destroy(t.ptr)
    # ---- the union of the ranks' tables is the oracle's table over all reads ----
    ss_all = O.SeqSet.from_byte_seqs(c["seqs"], data=(c["data"] if kind else None), sizeof_d1=c["width"] if kind else 0)
    want = O.filter_kmers(ss_all, k, O.COUNT_FILTER_SET if kind else O.COUNT_FILTER, c["min_obs"], stranded=stranded)
    tabs = [r_["table"] for r_ in res]
    hi = np.concatenate([t.key_hi for t in tabs]); lo = np.concatenate([t.key_lo for t in tabs])
    order = np.lexsort((lo, hi))
    assert len(order) == want.n
    assert np.array_equal(hi[order], want.key_hi) and np.array_equal(lo[order], want.key_lo)
    assert np.array_equal(np.concatenate([t.exts for t in tabs])[order], want.exts)
    if kind:
        got_sets = [s for t in tabs for s in masks_to_classes(t.set_off, t.set_val)]
        assert [got_sets[i] for i in order] == masks_to_classes(want.set_off, want.set_val)
    else:
        assert np.array_equal(np.concatenate([t.count for t in tabs])[order], want.count)
    for t in tabs:                                                 # every rank's own table is ascending
        if len(t) > 1:
            assert all((int(t.key_hi[i]), int(t.key_lo[i])) < (int(t.key_hi[i + 1]), int(t.key_lo[i + 1])) for i in range(len(t) - 1))
    if world > 1 or c["force"]:
        assert sum(r_["owned"] for r_ in res) == sum(r_["scanned"] for r_ in res)
    # ---- the second stage against the oracle's flow on the same per-rank tables ----
    glob = sorted(set(s for t in tabs for s in masks_to_classes(t.set_off, t.set_val))) if kind else None
    gid = {s: i for i, s in enumerate(glob)} if kind else None
    shard = []
    for t in tabs:
        data = (np.array([gid[s] for s in masks_to_classes(t.set_off, t.set_val)], dtype=np.uint32) if kind else t.count)
        shard.append(O.compress_kmers(k, stranded, O.SPEC_SCMAP_EQ if kind else O.SPEC_SAT_ADD, t.key_hi, t.key_lo, t.exts, data))
    og = O.graph_combine(shard).finish().compress_graph(stranded, O.SPEC_SCMAP_EQ if kind else O.SPEC_MAX)
    final = res[world - 1]["final"]
    assert final is not None and all(res[r]["final"] is None for r in range(world - 1))
    if kind:
        assert final.classes == glob
    if c["tree"] and world > 1:                                     # (the tree combines in tree order, not rank order)
        assert canonical_nodes(final.arrays(), k, stranded) == canonical_nodes(og.arrays(), k, stranded)
    else:
        assert graphs_equal(final.arrays(), og.arrays())
    for r in range(world):                                         # every rank's own shard graph, job-wide class ids
        assert graphs_equal(res[r]["local"].arrays(), shard[r].arrays())
