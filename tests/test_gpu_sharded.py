"""GPU: the staged sharded-counting entry points (dbg_shard_scan/scatter/count) with bins arriving as
several bin-ordered segments, as after the multi-GPU all-to-all.  Two virtual ranks share one GPU: their
scans use one global plan, the slabs are exchanged by tensor slicing, each virtual owner counts its bins;
the union must equal the oracle's filter_kmers over all reads.  Also runs the real orchestration
(distributed.sharded_filter_kmers) at world size 1."""
import ctypes as C
import importlib

import numpy as np
import pytest
import torch

import oracle_lib as O
from pkg import dbg, capi

pytestmark = pytest.mark.gpu
D = importlib.import_module("rust-debruijn_amd.distributed")


@pytest.fixture(scope="module")
def ctx():
    c = dbg.Context(0)
    yield c
    c.close()


def dev_seqset(engine, hs, with_data):
    dev = engine.device
    words = torch.from_numpy(hs.words.view(np.int64)).to(dev)
    start = torch.from_numpy(hs.start.view(np.int64)).to(dev)
    length = torch.from_numpy(hs.length.view(np.int32)).to(dev)
    data = torch.from_numpy(hs.data).to(dev) if with_data else None
    return engine.seqset(words, start, length, data, 1), (words, start, length, data)


def table_to_host(ctx, tab, k):
    h = capi.KmerTable()
    ctx.check(ctx.lib.dbg_table_to_host(ctx.h, C.byref(tab), C.byref(h)))
    out = dbg._table_from_c(h, k)
    ctx.lib.dbg_free_table(ctx.h, C.byref(h))
    return out


@pytest.mark.parametrize("merge", [True, False])
@pytest.mark.parametrize("kind,k,world,colours", [(0, 47, 2, 4), (1, 47, 2, 4), (0, 31, 3, 4), (0, 63, 2, 4), (1, 51, 2, 40), (1, 31, 3, 64)])
def test_virtual_ranks_multi_segment(ctx, kind, k, world, colours, merge):
    """colours > 24: the two-word colour layout, selected on every rank by the plan's global max_label.  merge: the sending
    ranks merge their identical records first (dbg_shard_plan.merge_dups); rank 0 never does, so weighted and plain records
    meet in every bin."""
    eng = D.HipEngine(ctx, torch.device("cuda", 0))
    n_reads, per = 6000, 6000 // world
    shards, keep = [], []
    for r in range(world):
        hs = dbg.synth_reads_host(n_reads=per, read_len=150, genome_len=n_reads * 150 // 30, error_rate=0.003,
                                  stranded=False, n_colours=colours, first_read=r * per)
        ss, kp = dev_seqset(eng, hs, kind == 1)
        shards.append(ss)
        keep.append(kp)
    total = sum(eng.count_instances(s, k) for s in shards)
    max_label = max(eng.max_label(s) for s in shards) if kind == 1 else 0      # (an all-reduce MAX in the real flow)
    assert max_label == (colours - 1 if kind == 1 else 0)
    plan = eng.plan(k, False, kind, 2, total, max_label, merge_dups=False)
    plan_m = eng.plan(k, False, kind, 2, total, max_label, merge_dups=merge)
    assert (plan_m.n_bins, plan_m.rec_words) == (plan.n_bins, plan.rec_words)
    rw, nb = plan.rec_words, plan.n_bins
    bounds = D.owner_bounds(nb, world, plan.bin_group)
    scanned = []
    for r, s in enumerate(shards):
        bin_off, n = eng.scan(s, plan_m if r else plan)
        recs = eng.scatter(plan, bin_off, n)
        scanned.append((bin_off, recs))
    if kind == 1 and colours > 24:                              # a plan that understates the labels is refused, not miscounted
        with pytest.raises(dbg.DbgError):
            eng.scan(shards[0], eng.plan(k, False, kind, 2, total, 0))
    merged = {}
    for owner in range(world):
        lo, hi = bounds[owner], bounds[owner + 1]
        nb_local = hi - lo
        slabs, hists = [], []
        for bin_off, recs in scanned:
            a, b = int(bin_off[lo]), int(bin_off[hi])
            slabs.append(recs[a * rw:b * rw])
            hists.append((bin_off[lo + 1:hi + 1] - bin_off[lo:hi]))
        recv = torch.cat(slabs) if sum(len(s) for s in slabs) else torch.zeros(1, dtype=torch.int64, device=eng.device)
        counts = [len(s) // rw for s in slabs]
        seg_off = torch.zeros(world, nb_local + 1, dtype=torch.int64, device=eng.device)
        base = 0
        for s in range(world):
            seg_off[s, 1:] = torch.cumsum(hists[s], 0)
            seg_off[s] += base
            base += counts[s]
        if owner % 2 == 0:
            tab = eng.count(plan, recv, seg_off, world, nb_local, total)
        else:
            # chunked form (what the pipelined exchange drives): the owned bins in three ranges, each with its own record
            # buffer and segment table, one sort at the end
            cb = D.chunk_bounds(nb_local, 3, plan.bin_group)
            eng.count_begin(plan, 64)                       # tiny hint: forces the grow-and-keep path of the output buffer
            for c in range(3):
                l2, h2 = cb[c], cb[c + 1]
                parts, seg_c, base_c = [], torch.zeros(world, h2 - l2 + 1, dtype=torch.int64, device=eng.device), 0
                for s_, (bin_off, recs) in enumerate(scanned):
                    a, b = int(bin_off[lo + l2]), int(bin_off[lo + h2])
                    parts.append(recs[a * rw:b * rw])
                    seg_c[s_] = bin_off[lo + l2:lo + h2 + 1] - bin_off[lo + l2] + base_c
                    base_c += b - a
                rc = torch.cat(parts) if base_c else torch.zeros(1, dtype=torch.int64, device=eng.device)
                eng.sync()
                eng.count_bins(plan, rc, seg_c, world, h2 - l2, 0)
            tab = eng.count_finish(plan)
        t = table_to_host(ctx, tab, k)
        eng.free_table(tab)
        assert t.keys() == sorted(t.keys())
        for i, key in enumerate(t.keys()):
            assert key not in merged
            merged[key] = (int(t.exts[i]), t.data(i))
    hs_all = dbg.synth_reads_host(n_reads=n_reads, read_len=150, genome_len=n_reads * 150 // 30, error_rate=0.003,
                                  stranded=False, n_colours=colours)
    want = O.filter_kmers(O.SeqSet(hs_all.words, hs_all.start, hs_all.length, None, hs_all.data, 1), k, kind, 2, stranded=False)
    assert sorted(merged) == want.keys()
    for i, key in enumerate(want.keys()):
        e, v = merged[key]
        assert e == int(want.exts[i])
        if kind == 1:
            assert v == [int(x) for x in want.set_val[int(want.set_off[i]):int(want.set_off[i + 1])]]
        else:
            assert v == int(want.count[i])


def test_orchestration_world1(ctx):
    eng = D.HipEngine(ctx, torch.device("cuda", 0))
    hs = dbg.synth_reads_host(n_reads=5000, read_len=150, error_rate=0.002, stranded=False, n_colours=4)
    ss, keep = dev_seqset(eng, hs, True)
    tab, total, n_local, n_recs = D.sharded_filter_kmers(eng, ss, 47, False, 1, 2)
    t = table_to_host(ctx, tab, 47)
    eng.free_table(tab)
    want = O.filter_kmers(O.SeqSet(hs.words, hs.start, hs.length, None, hs.data, 1), 47, O.COUNT_FILTER_SET, 2, stranded=False)
    assert total == 5000 * 104 and t.keys() == want.keys()
    assert np.array_equal(t.exts, want.exts) and np.array_equal(t.set_off, want.set_off) and np.array_equal(t.set_val, want.set_val)


def _records(eng, hs_list, k, kind, merge, stranded=False):
    """-> (tables of all owners merged into {key: (exts, value)}, records per sending rank)"""
    import virtual_ranks as V
    tabs, total, n_recs = V.owner_tables(eng, hs_list, k, stranded, kind, 2, merge_dups=merge, with_records=True)
    got = {}
    for tab in tabs:
        t = table_to_host(eng.ctx, tab, k)
        eng.free_table(tab)
        for i, key in enumerate(t.keys()):
            assert key not in got
            got[key] = (int(t.exts[i]), t.data(i))
    return got, n_recs


@pytest.mark.parametrize("kind,k,world", [(0, 47, 2), (0, 47, 8), (1, 31, 4), (0, 63, 8), (0, 51, 2), (0, 20, 2)])
def test_sender_merge_shrinks_the_exchange_and_changes_nothing(ctx, kind, k, world):
    """30x reads over `world` ranks: a rank holds 30/world copies of most records, the merge leaves about one of each.
    The tables with and without the merge are the same and equal the oracle's."""
    eng = D.HipEngine(ctx, torch.device("cuda", 0))
    n_reads = 16000
    per = n_reads // world
    hs_list = [dbg.synth_reads_host(n_reads=per, read_len=150, genome_len=n_reads * 150 // 30, error_rate=0.002, stranded=False,
                                    n_colours=3, first_read=r * per) for r in range(world)]
    plain, n_plain = _records(eng, hs_list, k, kind, False)
    merged, n_merged = _records(eng, hs_list, k, kind, True)
    assert merged == plain
    hs_all = dbg.synth_reads_host(n_reads=n_reads, read_len=150, genome_len=n_reads * 150 // 30, error_rate=0.002, stranded=False, n_colours=3)
    want = O.filter_kmers(O.SeqSet(hs_all.words, hs_all.start, hs_all.length, None, hs_all.data, 1), k, kind, 2, stranded=False)
    assert sorted(merged) == want.keys()
    for i, key in enumerate(want.keys()):
        e, v = merged[key]
        assert e == int(want.exts[i])
        if kind == 1:
            assert v == [int(x) for x in want.set_val[int(want.set_off[i]):int(want.set_off[i + 1])]]
        else:
            assert v == int(want.count[i])
    assert all(m <= p for m, p in zip(n_merged, n_plain))
    # how well it merges: rank 0's records against the number of distinct ones among them (identical records share a bin, and
    # these bins fit the merge kernel's staging area, so it should find nearly all of them)
    ss, keep = dev_seqset(eng, hs_list[0], kind == 1)
    total = sum(int(np.maximum(h.length.astype(np.int64) - k + 1, 0).sum()) for h in hs_list)
    plan = eng.plan(k, False, kind, 2, total, 2 if kind == 1 else 0, merge_dups=False)
    bin_off, n = eng.scan(ss, plan)
    assert n == n_plain[0]
    recs = eng.scatter(plan, bin_off, n).view(-1, plan.rec_words)
    distinct = int(torch.unique(recs, dim=0).shape[0])
    assert distinct <= n_merged[0], (distinct, n_merged[0])
    if k >= 31:                                                   # (small k: many short records per bin, more than the staging area holds)
        assert n_merged[0] <= distinct * 1.10 + 8, (distinct, n_merged[0], n_plain[0])
    if world <= 2 and kind == 0:
        assert n_merged[0] < 0.6 * n_plain[0]


@pytest.mark.parametrize("kind,k,stranded", [(0, 47, False), (1, 33, True), (0, 64, False)])
def test_sender_merge_heavy_duplicates(ctx, kind, k, stranded):
    """the same few reads hundreds of times over: weights beyond one record's 16, staging areas that refill, counts that
    saturate exactly as the oracle's do"""
    eng = D.HipEngine(ctx, torch.device("cuda", 0))
    rng = np.random.default_rng(5)
    base = [rng.integers(0, 4, size=int(rng.integers(k, 200))).astype(np.uint8) for _ in range(12)]    # base codes 0..3
    base.append(np.zeros(150, np.uint8))
    base.append(np.tile(np.array([0, 1], np.uint8), 75))
    hs_list, all_seqs, all_data = [], [], []
    for r in range(2):
        seqs = [base[int(i)] for i in rng.integers(0, len(base), size=1500)] + [base[0]] * (700 if r == 0 else 3)
        data = [int(x) for x in rng.integers(0, 5, size=len(seqs))]
        hs_list.append(dbg.HostSeqs.from_tuples([(q, 0, d) for q, d in zip(seqs, data)], data_width=1))
        all_seqs += seqs
        all_data += data
    plain, n_plain = _records(eng, hs_list, k, kind, False, stranded)
    merged, n_merged = _records(eng, hs_list, k, kind, True, stranded)
    assert merged == plain
    assert sum(n_merged) < sum(n_plain)            # (most of these bins outgrow their slab and are left alone)
    with ctx.options(DBG_SLAB_CAP="8192"):        # slabs that hold every bin: a record per 16 copies
        merged, n_merged = _records(eng, hs_list, k, kind, True, stranded)
    assert merged == plain
    assert sum(n_merged) < 0.2 * sum(n_plain)
    hs_all = dbg.HostSeqs.from_tuples([(q, 0, d) for q, d in zip(all_seqs, all_data)], data_width=1)
    want = O.filter_kmers(O.SeqSet(hs_all.words, hs_all.start, hs_all.length, None, hs_all.data, 1), k, kind, 2, stranded=stranded)
    assert sorted(merged) == want.keys()
    for i, key in enumerate(want.keys()):
        e, v = merged[key]
        assert e == int(want.exts[i])
        if kind == 1:
            assert v == [int(x) for x in want.set_val[int(want.set_off[i]):int(want.set_off[i + 1])]]
        else:
            assert v == int(want.count[i])


@pytest.mark.parametrize("k", [34, 50])
def test_sender_merge_without_spare_bits(ctx, k):
    """k = 34 and 50: the record has no 4 spare bits above its meta bits, so a plan that asks for the merge merges nothing
    (and the counting side must not read base bits as weights)"""
    eng = D.HipEngine(ctx, torch.device("cuda", 0))
    hs_list = [dbg.synth_reads_host(n_reads=4000, read_len=150, genome_len=8000 * 150 // 30, error_rate=0.002, stranded=False,
                                    n_colours=3, first_read=r * 4000) for r in range(2)]
    plain, n_plain = _records(eng, hs_list, k, 0, False)
    merged, n_merged = _records(eng, hs_list, k, 0, True)
    assert merged == plain and n_merged == n_plain
    hs_all = dbg.synth_reads_host(n_reads=8000, read_len=150, genome_len=8000 * 150 // 30, error_rate=0.002, stranded=False, n_colours=3)
    want = O.filter_kmers(O.SeqSet(hs_all.words, hs_all.start, hs_all.length, None, hs_all.data, 1), k, 0, 2, stranded=False)
    assert sorted(merged) == want.keys()
    assert [merged[key] for key in want.keys()] == [(int(e), int(c)) for e, c in zip(want.exts, want.count)]


@pytest.mark.parametrize("n_labels,k", [(5, 47), (30, 31), (64, 51)])
def test_sharded_sparse_label_alphabet(ctx, n_labels, k):
    """labels far beyond the kernel's 64 colours, at most 64 distinct ones over ALL ranks (no rank holds them all): the plan
    carries the job's alphabet, every rank maps label -> colour the same way, the label sets come back as labels"""
    import virtual_ranks as V
    eng = D.HipEngine(ctx, torch.device("cuda", 0))
    rng = np.random.default_rng(n_labels)
    alphabet = np.sort(rng.choice(np.arange(64, 65536), size=n_labels, replace=False)).astype(np.uint16)
    world, per = 3, 3000
    hs_list, datas = [], []
    for r in range(world):
        hs = dbg.synth_reads_host(n_reads=per, read_len=150, genome_len=world * per * 150 // 30, error_rate=0.002, stranded=False,
                                  n_colours=n_labels, first_read=r * per)
        # rank r never sees the labels whose index is r mod world... unless there are too few to go round
        d = hs.data.astype(np.int64)
        if n_labels >= 2 * world:
            d = np.where(d % world == r, (d + 1) % n_labels, d)
        datas.append(alphabet[d])
        hs_list.append(dbg.HostSeqs(hs.words, hs.start, hs.length, None, alphabet[d], 2))
    tabs, total = V.owner_tables(eng, hs_list, k, False, 1, 2)
    got = {}
    for tab in tabs:
        t = table_to_host(ctx, tab, k)
        eng.free_table(tab)
        for i, key in enumerate(t.keys()):
            assert key not in got
            got[key] = (int(t.exts[i]), t.data(i))
    hs_all = dbg.synth_reads_host(n_reads=world * per, read_len=150, genome_len=world * per * 150 // 30, error_rate=0.002, stranded=False, n_colours=n_labels)
    want = O.filter_kmers(O.SeqSet(hs_all.words, hs_all.start, hs_all.length, None, np.concatenate(datas), 2), k, 1, 2, stranded=False)
    assert sorted(got) == want.keys()
    for i, key in enumerate(want.keys()):
        e, v = got[key]
        assert e == int(want.exts[i])
        assert v == [int(x) for x in want.set_val[int(want.set_off[i]):int(want.set_off[i + 1])]]
    # a plan that leaves one of a rank's labels out, an unsorted list, and more than 64 labels are refused
    ss, keep = V.dev_seqset(eng, hs_list[0], True)
    mine = np.unique(datas[0]).tolist()
    with pytest.raises(dbg.DbgError):
        eng.scan(ss, eng.plan(k, False, 1, 2, total, int(alphabet[-1]), labels=mine[1:]))
    with pytest.raises(dbg.DbgError):
        eng.plan(k, False, 1, 2, total, int(alphabet[-1]), labels=mine[::-1] if len(mine) > 1 else [5, 5])
    with pytest.raises(dbg.DbgError):
        eng.plan(k, False, 1, 2, total, 70000, labels=list(range(100, 165)))
    with pytest.raises(dbg.DbgError):                                  # labels >= 64 and no list
        eng.plan(k, False, 1, 2, total, int(alphabet[-1]))


def test_orchestration_world1_sparse_labels(ctx):
    """distributed.sharded_filter_kmers collects the alphabet itself"""
    eng = D.HipEngine(ctx, torch.device("cuda", 0))
    hs = dbg.synth_reads_host(n_reads=5000, read_len=150, error_rate=0.002, stranded=False, n_colours=6)
    lab = np.array([7, 64, 900, 4096, 40000, 65535], dtype=np.uint16)[hs.data]
    import virtual_ranks as V
    ss, keep = V.dev_seqset(eng, dbg.HostSeqs(hs.words, hs.start, hs.length, None, lab, 2), True)
    tab, total, n_local, n_recs = D.sharded_filter_kmers(eng, ss, 47, False, 1, 2)
    t = table_to_host(ctx, tab, 47)
    eng.free_table(tab)
    want = O.filter_kmers(O.SeqSet(hs.words, hs.start, hs.length, None, lab, 2), 47, O.COUNT_FILTER_SET, 2, stranded=False)
    assert t.keys() == want.keys()
    assert np.array_equal(t.exts, want.exts) and np.array_equal(t.set_off, want.set_off) and np.array_equal(t.set_val, want.set_val)
