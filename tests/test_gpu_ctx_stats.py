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
    small.struct_size = 16
    small.slab_piece_bytes = 12345
    ctx.check(ctx.lib.dbg_ctx_get_stats(ctx.h, C.byref(small)))
    assert small.struct_size == 16 and small.slab_piece_bytes == 12345 and small.slab_bytes == st["slab_bytes"]
    # trimming drops the slab: nothing left to probe, and the account says so
    ctx.trim()
    assert ctx.probe_slab() is None and ctx.stats()["slab_pooled"] == 0 and ctx.stats()["n_trims"] >= 1
    # warming maps a slab of the asked size into the pool (piecewise from 4 GB on) and pins host blocks ahead of the first call
    ctx.warm(5 << 30, 64 << 20)
    st = ctx.stats()
    assert st["n_vmm_blocks"] == 1 and st["pooled_bytes"] >= 5 << 30 and st["n_pinned_alloc"] >= 1
    ctx.close()
