"""GPU parity: CountFilterSet over full-width D1 labels (CountFilterSet<D: Ord> takes any u32, filter.rs:68-101).  Labels of 2^24 and more
-- hashes, barcodes, 32-bit ids -- run through the order-preserving label dictionary (labeldict.hip) on every route of the single call
(label lists, generic, dense), bit-exact against the CPU oracle: the sorted, de-duplicated label lists come back as the labels
themselves, 2^32 - 1 included."""
import numpy as np
import pytest

import oracle_lib as O
from pkg import dbg
from test_gpu_filter import assert_tables_equal, to_host_seqs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = dbg.Context(0)
    yield c
    c.close()


def labelled(hs, n_labels, seed, lo=0, hi=1 << 32):
    rng = np.random.default_rng(seed)
    alphabet = np.unique(rng.integers(lo, hi, size=n_labels * 2, dtype=np.uint64))[:n_labels]
    alphabet[-1] = (1 << 32) - 1                                  # the largest u32
    alphabet[0] = lo
    lab = alphabet[rng.integers(0, len(alphabet), size=len(hs.start))].astype(np.uint32)
    lab[::7] = np.uint32(alphabet[-1])                            # both ends of the range occur, often
    lab[3::7] = np.uint32(alphabet[0])
    return O.SeqSet(hs.words, hs.start, hs.length, None, lab, 4), alphabet


@pytest.mark.parametrize("path", ["auto", "fast", "generic"])
@pytest.mark.parametrize("n_labels,k,stranded,min_obs,report_all", [
    (3000, 47, False, 2, False), (40, 31, True, 1, False), (700, 63, False, 2, True), (2, 20, False, 1, False), (5000, 11, False, 2, False),
    (9, 8, True, 1, False)])
def test_full_width_labels(ctx, path, n_labels, k, stranded, min_obs, report_all):
    if path == "fast" and k < 16:
        pytest.skip("the fast path starts at k = 16")
    hs = dbg.synth_reads_host(n_reads=5000, read_len=150, error_rate=0.004, stranded=stranded, n_colours=4)
    ss, alphabet = labelled(hs, n_labels, n_labels * 7 + k)
    want = O.filter_kmers(ss, k, O.COUNT_FILTER_SET, min_obs, stranded=stranded, report_all=report_all)
    old = ctx.set_option("DBG_PATH", path)
    try:
        got, _ = dbg.filter_kmers(to_host_seqs(ss, 4), dbg.CountFilterSet(min_obs), stranded, report_all, 4, k=k, ctx=ctx)
    finally:
        ctx.set_option("DBG_PATH", old)
    assert_tables_equal(got, want, True)
    assert int(got.set_val.max()) == (1 << 32) - 1 and int(got.set_val.min()) == int(alphabet[0])
    assert np.isin(got.set_val, alphabet.astype(np.uint32)).all()


def test_full_width_labels_all_large_and_dense_ranks(ctx):
    """every label >= 2^31 (the sign bit), neighbouring values (ranks differ by one where labels differ by one), labels sharing a 1024-value
    block of the presence bitmap and straddling its 64-bit words"""
    hs = dbg.synth_reads_host(n_reads=4000, read_len=150, error_rate=0.003, stranded=False, n_colours=4)
    base = (1 << 31) + 12345
    alphabet = np.array([base, base + 1, base + 2, base + 63, base + 64, base + 65, base + 1023, base + 1024, base + 1025, (1 << 32) - 2, (1 << 32) - 1], dtype=np.uint64)
    rng = np.random.default_rng(5)
    lab = alphabet[rng.integers(0, len(alphabet), size=len(hs.start))].astype(np.uint32)
    ss = O.SeqSet(hs.words, hs.start, hs.length, None, lab, 4)
    want = O.filter_kmers(ss, 47, O.COUNT_FILTER_SET, 2, stranded=False)
    got, _ = dbg.filter_kmers(to_host_seqs(ss, 4), dbg.CountFilterSet(2), False, False, 4, k=47, ctx=ctx)
    assert_tables_equal(got, want, True)
    assert set(np.unique(got.set_val).tolist()) <= set(alphabet.tolist()) and int(np.diff(got.set_off).max()) >= 5


def test_full_width_labels_compact_sets_and_device_entry(ctx):
    """the device-resident entry point and the compact CSR of the host boundary carry the translated labels too"""
    import ctypes as C
    from pkg import capi
    hs = dbg.synth_reads_host(n_reads=3000, read_len=150, error_rate=0.003, stranded=False, n_colours=4)
    ss, alphabet = labelled(hs, 500, 99, lo=1 << 24)
    want = O.filter_kmers(ss, 51, O.COUNT_FILTER_SET, 2, stranded=False)
    hss = to_host_seqs(ss, 4).c_struct()
    fp = capi.FilterParams(51, 0, 1, 2, 0, 4, 1)                   # compact_sets bit 0: u32 offsets + D1-wide labels
    t = capi.KmerTable()
    ctx.check(ctx.lib.dbg_filter_kmers(ctx.h, C.byref(hss), C.byref(fp), C.byref(t)))
    assert t.n == want.n and t.set_val_width == 4
    sv = np.ctypeslib.as_array(C.cast(t.set_val, C.POINTER(C.c_uint32)), shape=(int(t.n_set_val),)).copy()
    ctx.lib.dbg_free_table(ctx.h, C.byref(t))
    assert np.array_equal(sv, want.set_val)


@pytest.mark.parametrize("world,k,lists_off", [(3, 47, False), (2, 31, True), (4, 63, False), (2, 11, False)])
def test_full_width_labels_rank_spanning(world, k, lists_off):
    """dbg_shard_filter_kmers_dev with labels of 2^24 and more: the ranks' distinct labels are gathered into ONE job-wide dictionary, the
    exchange and the counting run on the ranks, every owner translates its rows back -- the union of the ranks' tables is the oracle's
    table over all reads, label lists included.  One rank holds labels below 2^24 only, one rank holds no reads at all (world 4)."""
    import ctypes as C
    from pkg import capi, D
    from test_gpu_shard_fuzz import make_group, run_ranks, masks_to_classes
    hs = dbg.synth_reads_host(n_reads=3000, read_len=150, error_rate=0.003, stranded=False, n_colours=4)
    ss, alphabet = labelled(hs, 400, 1234 + k)
    lab = ss.data.copy()
    n = len(hs.start)
    bounds = [n * r // world for r in range(world + 1)]
    if world == 4:
        bounds[2] = bounds[1]                                      # rank 1 holds nothing
    lab[bounds[0]:bounds[1]] = lab[bounds[0]:bounds[1]] % (1 << 20)     # rank 0: small labels only
    ss_all = O.SeqSet(hs.words, hs.start, hs.length, None, lab, 4)
    want = O.filter_kmers(ss_all, k, O.COUNT_FILTER_SET, 2, stranded=False)
    assert int(want.set_val.max()) >= 1 << 24
    trs, keep = make_group(world)
    lib = capi.load()

    def rank_main(r):
        ctx = dbg.Context(0)
        try:
            if lists_off:
                ctx.set_option("DBG_LABEL_LISTS", "0")
            lo, hi = bounds[r], bounds[r + 1]
            st0 = int(hs.start[lo]) if lo < hi else 0
            part = dbg.HostSeqs(hs.words, (hs.start[lo:hi]).copy(), hs.length[lo:hi].copy(), None, lab[lo:hi].copy(), 4)
            dev, hc = capi.SeqSet(), part.c_struct()
            ctx.check(lib.dbg_seqset_to_device(ctx.h, C.byref(hc), C.byref(dev)))
            tab, st = D.shard_filter_kmers_c(ctx, trs[r], dev, k, False, 1, 2)
            lib.dbg_seqset_free_device(ctx.h, C.byref(dev))
            h = capi.KmerTable()
            ctx.check(lib.dbg_table_to_host(ctx.h, C.byref(tab), C.byref(h)))
            th = dbg._table_from_c(h, k)
            lib.dbg_free_table(ctx.h, C.byref(h))
            lib.dbg_free_table(ctx.h, C.byref(tab))
            return th
        finally:
            ctx.close()

    tabs = run_ranks(world, rank_main)
    for t in trs:
        lib.dbg_transport_destroy(t.ptr)
    hi_ = np.concatenate([t.key_hi for t in tabs]); lo_ = np.concatenate([t.key_lo for t in tabs])
    order = np.lexsort((lo_, hi_))
    assert len(order) == want.n and np.array_equal(hi_[order], want.key_hi) and np.array_equal(lo_[order], want.key_lo)
    assert np.array_equal(np.concatenate([t.exts for t in tabs])[order], want.exts)
    got_sets = [s for t in tabs for s in masks_to_classes(t.set_off, t.set_val)]
    assert [got_sets[i] for i in order] == masks_to_classes(want.set_off, want.set_val)


def test_full_width_labels_full_size_property():
    """BASELINE configs[1] shape at a fifth of its size (2e7 reads), 5000 labels: the table over full-width u32 labels equals the table over
    their ranks with every label translated back -- the order-preserving dictionary commutes with CountFilterSet (sorting and
    de-duplicating ranks = sorting and de-duplicating labels, filter.rs:96-98).  Everything compared on the device."""
    import ctypes as C
    import torch
    from pkg import capi
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
    del colour
    g = torch.Generator(device=dev); g.manual_seed(11)
    small = torch.randint(0, 5000, (n_reads,), device=dev, generator=g, dtype=torch.int64)
    alphabet = torch.unique(torch.randint(0, 1 << 32, (6000,), device=dev, generator=g, dtype=torch.int64))[:5000]
    alphabet[-1] = (1 << 32) - 1
    assert int(alphabet.numel()) == 5000 and bool((alphabet[1:] > alphabet[:-1]).all())
    big = alphabet[small]
    lab_small = small.to(torch.int32).contiguous()
    lab_big = (big - ((big >> 31) << 32)).to(torch.int32).contiguous()      # the u32 bit patterns in an int32 tensor
    fp = capi.FilterParams(k, 0, 1, 2, 0, 4)

    def run(lab):
        ss = capi.SeqSet(words.data_ptr(), nw, start.data_ptr(), length.data_ptr(), None, lab.data_ptr(), 4, n_reads)
        t = capi.KmerTable()
        ctx.check(lib.dbg_filter_kmers_dev(ctx.h, C.byref(ss), C.byref(fp), C.byref(t)))
        return t
    from test_gpu_fullsize import dev_view
    ts = run(lab_small)
    n, nsv = int(ts.n), int(ts.n_set_val)
    keys_s = (dev_view(ts.key_hi, n).clone(), dev_view(ts.key_lo, n).clone())
    off_s = dev_view(ts.set_off, n + 1).clone()
    val_s = dev_view(ts.set_val, nsv, "<i4").to(torch.int64)
    want = alphabet[val_s]                                                    # the translated lists
    del val_s
    lib.dbg_free_table(ctx.h, C.byref(ts))
    tb = run(lab_big)
    assert int(tb.n) == n and int(tb.n_set_val) == nsv and n > 10_000_000
    assert torch.equal(dev_view(tb.key_hi, n), keys_s[0]) and torch.equal(dev_view(tb.key_lo, n), keys_s[1])
    assert torch.equal(dev_view(tb.set_off, n + 1), off_s)
    got = dev_view(tb.set_val, nsv, "<i4").to(torch.int64) & 0xFFFFFFFF
    assert torch.equal(got, want)
    lib.dbg_free_table(ctx.h, C.byref(tb))
    ctx.close()
