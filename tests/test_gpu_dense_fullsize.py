"""GPU, the dense path (k <= 15) at the size of BASELINE configs[1]'s read set scaled to short k-mers: 5 * 10^7 reads x 150 bp -- 6.8 * 10^9
(k = 15) and 7 * 10^9 (k = 11) k-mer instances, i.e. two batches of the partitioned form.  Size-independent properties of filter_kmers
(src/filter.rs:139-231) and an implementation-independent one:

  * keys strictly ascending (the order handed to BoomHashMap2::new, filter.rs:227-230);
  * with CountFilter(1) every instance is counted once: sum(count) == N while no key saturates (k = 15);
  * the partitioned form (instances brought into key-prefix order, counted in LDS: densepath.hip) and the device-atomic form
    (DBG_DENSE_PART=0: one atomic per instance on the directly addressed table) are two independent routes to the same table: keys, Exts,
    counts and -- CountFilterSet(2) -- label lists must be equal element for element; so must the first level's two forms (DBG_DENSE_RAW).

Everything is compared on the device (torch views over the library's device arrays); nothing reads the reference tree."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
from pkg import dbg
from test_gpu_fullsize import dev_view

pytestmark = pytest.mark.gpu
L = 150
N_READS = int(50_000_000 * float(os.environ.get("DBG_FULLSIZE_SCALE", 1.0)))


@pytest.fixture(scope="module")
def env():
    import importlib
    import torch
    capi = importlib.import_module("rust-debruijn_amd._capi")
    ctx = dbg.Context(0)
    lib = ctx.lib
    dev = torch.device("cuda", 0)
    p = dbg.synth_params(n_reads=N_READS, read_len=L, genome_len=N_READS * L // 30, error_rate=0.001, stranded=False, n_colours=4, first_read=0)
    nw = lib.dbg_synth_words(C.byref(p))
    words = torch.empty(nw, dtype=torch.int64, device=dev)
    start = torch.empty(N_READS, dtype=torch.int64, device=dev)
    length = torch.empty(N_READS, dtype=torch.int32, device=dev)
    colour = torch.empty(N_READS, dtype=torch.uint8, device=dev)
    ctx.check(lib.dbg_synth_reads_dev(ctx.h, C.byref(p), words.data_ptr(), start.data_ptr(), length.data_ptr(), colour.data_ptr()))
    e = dict(capi=capi, ctx=ctx, lib=lib, torch=torch, nw=nw, words=words, start=start, length=length, colour=colour)
    yield e
    ctx.close()
    e.clear()
    del words, start, length, colour
    torch.cuda.empty_cache()


def run(e, k, summarizer, min_obs, n_reads=None, **opts):
    capi, ctx, lib = e["capi"], e["ctx"], e["lib"]
    is_set = summarizer == 1
    ss = capi.SeqSet(e["words"].data_ptr(), e["nw"], e["start"].data_ptr(), e["length"].data_ptr(), None,
                     e["colour"].data_ptr() if is_set else None, 1 if is_set else 0, N_READS if n_reads is None else n_reads)
    fp = capi.FilterParams(k, 0, summarizer, min_obs, 0, 4)
    t = capi.KmerTable()
    with ctx.options(DBG_PATH="dense", **opts):
        ctx.check(lib.dbg_filter_kmers_dev(ctx.h, C.byref(ss), C.byref(fp), C.byref(t)))
    return t


def same_tables(e, a, b, is_set):
    torch = e["torch"]
    assert a.n == b.n and a.n > 0
    assert torch.equal(dev_view(a.key_lo, a.n), dev_view(b.key_lo, b.n))
    assert torch.equal(dev_view(a.exts, a.n, "|u1"), dev_view(b.exts, b.n, "|u1"))
    if is_set:
        assert a.n_set_val == b.n_set_val
        assert torch.equal(dev_view(a.set_off, a.n + 1), dev_view(b.set_off, b.n + 1))
        assert torch.equal(dev_view(a.set_val, a.n_set_val, "<u4"), dev_view(b.set_val, b.n_set_val, "<u4"))
    else:
        assert torch.equal(dev_view(a.count, a.n, "<u2"), dev_view(b.count, b.n, "<u2"))


@pytest.mark.parametrize("k", [11, 15])
def test_dense_fullsize_partitioned_equals_atomic(env, k):
    e = env
    lib, ctx, torch = e["lib"], e["ctx"], e["torch"]
    n_expected = N_READS * (L - k + 1)
    part = run(e, k, 0, 1)
    assert int(part.n_kmer_instances) == n_expected
    lo = dev_view(part.key_lo, part.n)
    assert bool((lo[1:] > lo[:-1]).all().item())                         # (keys < 2^30: a signed compare is an unsigned one)
    cnt = dev_view(part.count, part.n, "<u2").to(torch.int64)
    if int(cnt.max().item()) < 65535:                                    # k = 15: no key saturates, so the counts add up to the instances
        assert int(cnt.sum().item()) == n_expected
    else:
        assert k == 11
    del lo, cnt
    atomic = run(e, k, 0, 1, DBG_DENSE_PART="0", DBG_DENSE_RANGES="0")
    same_tables(e, part, atomic, False)
    lib.dbg_free_table(ctx.h, C.byref(atomic))
    walking = run(e, k, 0, 1, DBG_DENSE_RAW="0")                         # the first level's other form
    same_tables(e, part, walking, False)
    lib.dbg_free_table(ctx.h, C.byref(walking))
    lib.dbg_free_table(ctx.h, C.byref(part))


@pytest.mark.parametrize("k", [12])
def test_dense_fullsize_label_sets(env, k):
    e = env
    lib, ctx = e["lib"], e["ctx"]
    part = run(e, k, 1, 2)
    atomic = run(e, k, 1, 2, DBG_DENSE_PART="0")
    same_tables(e, part, atomic, True)
    counted = run(e, k, 0, 2)                                            # CountFilter(2) selects the same keys (filter.rs:85-100 counts observations over all labels)
    assert counted.n == part.n and e["torch"].equal(dev_view(counted.key_lo, counted.n), dev_view(part.key_lo, part.n))
    for t in (part, atomic, counted):
        lib.dbg_free_table(ctx.h, C.byref(t))


@pytest.mark.parametrize("k,summarizer", [(11, 0), (15, 0), (12, 1), (15, 1)])
def test_dense_fullsize_prefix_bit_exact(env, k, summarizer):
    """A prefix of the SAME full-size stream against the oracle, row for row (round 5; the tests above compare the dense path's
    forms with one another): the partitioned form in one batch and in many (DBG_DENSE_BATCH), and the device-atomic form --
    keys, Exts, counts / label lists of filter_kmers (src/filter.rs:139-231) over reads [0, m)."""
    e = env
    capi, lib, ctx = e["capi"], e["lib"], e["ctx"]
    m = min(100_000, N_READS)
    hs = dbg.synth_reads_host(n_reads=m, read_len=L, genome_len=N_READS * L // 30, error_rate=0.001, stranded=False, n_colours=4)
    want = O.filter_kmers(O.SeqSet(hs.words, hs.start, hs.length, None, hs.data if summarizer else None, 1 if summarizer else 0), k,
                          O.COUNT_FILTER_SET if summarizer else O.COUNT_FILTER, 1, stranded=False)
    as_np = lambda p, ct, cnt: np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(max(cnt, 1),))[:cnt].copy()
    for opts in ({}, {"DBG_DENSE_BATCH": "3000000"}, {"DBG_DENSE_PART": "0", "DBG_DENSE_RANGES": "0"}):
        t = run(e, k, summarizer, 1, n_reads=m, **opts)
        h = capi.KmerTable()
        ctx.check(lib.dbg_table_to_host(ctx.h, C.byref(t), C.byref(h)))
        lib.dbg_free_table(ctx.h, C.byref(t))
        n = int(h.n)
        assert n == want.n, opts
        assert np.array_equal(as_np(h.key_lo, C.c_uint64, n), want.key_lo) and np.array_equal(as_np(h.exts, C.c_uint8, n), want.exts), opts
        if summarizer:
            assert np.array_equal(as_np(h.set_off, C.c_uint64, n + 1), want.set_off), opts
            assert np.array_equal(as_np(h.set_val, C.c_uint32, int(h.n_set_val)), want.set_val), opts
        else:
            assert np.array_equal(as_np(h.count, C.c_uint16, n), want.count), opts
        lib.dbg_free_table(ctx.h, C.byref(h))
