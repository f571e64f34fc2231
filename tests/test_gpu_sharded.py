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


@pytest.mark.parametrize("kind,k,world,colours", [(0, 47, 2, 4), (1, 47, 2, 4), (0, 31, 3, 4), (0, 63, 2, 4), (1, 51, 2, 40), (1, 31, 3, 64)])
def test_virtual_ranks_multi_segment(ctx, kind, k, world, colours):
    """colours > 24: the two-word colour layout, selected on every rank by the plan's global max_label"""
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
    plan = eng.plan(k, False, kind, 2, total, max_label)
    rw, nb = plan.rec_words, plan.n_bins
    bounds = D.owner_bounds(nb, world, plan.bin_group)
    scanned = []
    for s in shards:
        bin_off, n = eng.scan(s, plan)
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
