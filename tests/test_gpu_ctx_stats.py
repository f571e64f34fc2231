"""GPU: the allocation account of a ctx and the record of how the scan's slab is backed (dbg_ctx_get_stats, dbg_ctx_probe_slab,
dbg_ctx_warm, dbg_abi_version) -- the library never degrades silently (round-5 review, item 1)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from pkg import dbg, capi

pytestmark = pytest.mark.gpu


def test_stats_probe_and_warm():
    ctx = dbg.Context(0)
    assert ctx.lib.dbg_abi_version() >= 6
    st = ctx.stats()
    assert st["slab_backing"] == 0 and st["n_hipmalloc"] == 0 and ctx.probe_slab() is None
    hs = dbg.synth_reads_host(n_reads=3000, read_len=150, error_rate=0.001, stranded=False, n_colours=4)
    t, _ = dbg.filter_kmers(hs, dbg.CountFilter(2), False, False, 4, k=31, ctx=ctx)
    want = O.filter_kmers(O.SeqSet(hs.words, hs.start, hs.length, None, hs.data, 1), 31, O.COUNT_FILTER, 2, stranded=False)
    assert len(t) == want.n and np.array_equal(t.key_lo, want.key_lo) and np.array_equal(t.count, want.count)
    st = ctx.stats()
    assert st["slab_backing_name"] == "plain (slab < 4 GB)" and st["slab_bytes"] > 0 and st["slab_pooled"] == 1
    assert st["n_hipmalloc"] > 0 and st["pooled_high_water"] >= st["pooled_bytes"] > 0 and st["s_hipmalloc"] > 0
    pr = ctx.probe_slab(1 << 16)
    assert pr is not None and pr[0] > 0 and pr[1] >= 1 << 12
    # the probe scribbles over a pooled block only: the next call is unharmed
    t2, _ = dbg.filter_kmers(hs, dbg.CountFilter(2), False, False, 4, k=31, ctx=ctx)
    assert np.array_equal(t2.key_lo, want.key_lo) and np.array_equal(t2.count, want.count)
    # a short struct is filled up to its size only (hosts compiled against an older header)
    small = capi.CtxStats()
    small.struct_size = 16                  # struct_size, slab_backing, slab_bytes
    small.slab_pooled = 12345
    ctx.check(ctx.lib.dbg_ctx_get_stats(ctx.h, C.byref(small)))
    assert small.struct_size == 16 and small.slab_pooled == 12345 and small.slab_bytes == st["slab_bytes"]
    # trimming drops the slab: nothing left to probe, and the account says so
    ctx.trim()
    assert ctx.probe_slab() is None and ctx.stats()["slab_pooled"] == 0 and ctx.stats()["n_trims"] >= 1
    # warming puts a device block of the asked size into the pool and pins host blocks ahead of the first call
    n0 = ctx.stats()["n_hipmalloc"]
    ctx.warm(5 << 30, 64 << 20)
    st = ctx.stats()
    assert st["n_hipmalloc"] == n0 + 1 and st["pooled_bytes"] >= 5 << 30 and st["n_pinned_alloc"] >= 1
    ctx.close()


def test_slab_tournament_keeps_the_fastest_block():
    """The first four calls of a shape with a slab of 4 GB or more each scan into a fresh block; from the fifth on the fastest one serves.
    Same table every time."""
    import torch
    from pkg import D
    ctx = dbg.Context(0)
    lib, dev = ctx.lib, torch.device("cuda", 0)
    n_reads, k = 20_000_000, 47
    p = dbg.synth_params(n_reads=n_reads, read_len=150, genome_len=n_reads * 150 // 30, error_rate=0.001, stranded=False, n_colours=4, first_read=0)
    nw = lib.dbg_synth_words(C.byref(p))
    words = torch.empty(nw, dtype=torch.int64, device=dev)
    start = torch.empty(n_reads, dtype=torch.int64, device=dev)
    length = torch.empty(n_reads, dtype=torch.int32, device=dev)
    colour = torch.empty(n_reads, dtype=torch.uint8, device=dev)
    ctx.check(lib.dbg_synth_reads_dev(ctx.h, C.byref(p), words.data_ptr(), start.data_ptr(), length.data_ptr(), colour.data_ptr()))
    ss = capi.SeqSet(words.data_ptr(), nw, start.data_ptr(), length.data_ptr(), None, None, 0, n_reads)
    fp = capi.FilterParams(k, 0, 0, 2, 0, 4)
    seen = set()
    for call in range(6):
        t = capi.KmerTable()
        ctx.check(lib.dbg_filter_kmers_dev(ctx.h, C.byref(ss), C.byref(fp), C.byref(t)))
        seen.add((int(t.n), D.table_digest(t, dev)))
        lib.dbg_free_table(ctx.h, C.byref(t))
        st = ctx.stats()
        assert st["slab_backing"] == 1 and st["slab_bytes"] >= 4 << 30
        assert st["slab_trials_done"] == min(call + 1, 4) and len(st["slab_trial_ms"]) == min(call + 1, 4) and min(st["slab_trial_ms"]) > 0
    assert len(seen) == 1
    st = ctx.stats()
    assert st["n_fresh_blocks"] == 3 and st["n_raw_free"] == 3 and st["slab_candidates_pooled"] == 1 and st["slab_pooled"] == 1
    # a single candidate asked for: no trials, no fresh blocks
    ctx2 = dbg.Context(0)
    ctx2.set_option("DBG_SLAB_TRIALS", "1")
    for call in range(2):
        t = capi.KmerTable()
        ctx2.check(lib.dbg_filter_kmers_dev(ctx2.h, C.byref(ss), C.byref(fp), C.byref(t)))
        seen.add((int(t.n), D.table_digest(t, dev)))
        lib.dbg_free_table(ctx2.h, C.byref(t))
    st2 = ctx2.stats()
    assert len(seen) == 1 and st2["n_fresh_blocks"] == 0 and st2["n_raw_free"] == 0 and st2["slab_trials_done"] == 0
    ctx2.close()
    ctx.close()
