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
