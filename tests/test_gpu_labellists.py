"""GPU parity of CountFilterSet over large label alphabets on the fast path (fast_labellists.hpp): more than 1024 distinct labels,
labels >= 65536, u32 labels -- one pass of bin_labels_kernel (count, append every instance's label to its k-mer's segment, sort +
de-duplicate the segments in place) -- against the CPU oracle, bit-exact.  filter.rs:85-100: validity counts observations over
all labels (unsaturated), the list is the sorted, de-duplicated labels.  DBG_PATH=fast: no silent fall-back to the generic path."""
import numpy as np
import pytest

import oracle_lib as O
import refgen as R
from pkg import dbg
from test_gpu_filter import assert_tables_equal, random_reads, to_host_seqs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = dbg.Context(0)
    yield c
    c.close()


@pytest.fixture(autouse=True)
def force_fast(ctx):
    old = ctx.set_option("DBG_PATH", "fast")
    old2 = ctx.set_option("DBG_LABEL_LISTS", "1")      # also where label groups would apply (65..1024 labels)
    yield
    ctx.set_option("DBG_PATH", old)
    ctx.set_option("DBG_LABEL_LISTS", old2)


def check(ctx, ss, k, min_obs, stranded, width, report_all=False):
    want = O.filter_kmers(ss, k, O.COUNT_FILTER_SET, min_obs, stranded=stranded, report_all=report_all)
    got, allk = dbg.filter_kmers(to_host_seqs(ss, width), dbg.CountFilterSet(min_obs), stranded, report_all, 4, k=k, ctx=ctx)
    assert_tables_equal(got, want, True)
    return got, want


@pytest.mark.parametrize("n_labels,k,stranded,min_obs,width,report_all", [
    (5000, 47, False, 2, 4, False), (2000, 31, True, 1, 2, False), (1500, 63, False, 2, 4, False), (3000, 20, False, 3, 4, True),
    (70000, 51, False, 2, 4, False), (1100, 16, True, 2, 2, False), (4000, 33, False, 1, 4, True), (9000, 56, False, 2, 4, False),
    (100, 47, False, 2, 1, False), (300, 24, True, 2, 2, False), (65, 64, False, 1, 1, False)])
def test_label_lists(ctx, n_labels, k, stranded, min_obs, width, report_all):
    hs = dbg.synth_reads_host(n_reads=6000, read_len=150, error_rate=0.004, stranded=stranded, n_colours=4)
    rng = np.random.default_rng(n_labels * 31 + k)
    top = {1: 256, 2: 65536, 4: 1 << 24}[width]                  # (the C ABI takes D1 values < 2^24, dbg_mi355x.h)
    if n_labels > top:
        pytest.skip("alphabet does not fit the label width")
    alphabet = np.unique(rng.integers(0, top, size=n_labels * 2, dtype=np.uint64))[:n_labels] if top > 65536 else \
        np.sort(rng.choice(np.arange(0, top), size=n_labels, replace=False)).astype(np.uint64)
    if width == 4:
        alphabet[-1] = (1 << 24) - 1                               # the largest label the ABI takes
        alphabet[0] = 0
    lab = alphabet[rng.integers(0, len(alphabet), size=len(hs.start))]
    ss = O.SeqSet(hs.words, hs.start, hs.length, None, lab.astype({1: np.uint8, 2: np.uint16, 4: np.uint32}[width]), width)
    got, want = check(ctx, ss, k, min_obs, stranded, width, report_all)
    assert len(got) > 1000
    sizes = np.diff(got.set_off)
    assert int(sizes.max()) >= 3


def test_label_lists_long_segments_and_repeats(ctx):
    """k-mers seen far more than 64 times (segments sorted by the wave-level network in global memory), next to ordinary ones;
    a threshold above the u16-saturated count; few labels on a heavy k-mer (heavy de-duplication) and many"""
    rng = np.random.default_rng(12)
    genome = R.random_dna(rng, 3000)
    rep = R.random_dna(rng, 80)
    seqs, lab = [], []
    for i in range(900):                                            # the repeat, under 3 labels: 34 k-mers x 900 observations
        seqs.append(rep); lab.append(int(rng.integers(0, 3)) * 100000)
    for i in range(700):                                            # poly-A: A^47 seen 104 times per read, ~1900 labels
        seqs.append(np.zeros(150, np.uint8)); lab.append(int(rng.integers(0, 1900)) * 7 + 5)
    for i in range(1500):
        a = int(rng.integers(0, 2850))
        seqs.append(genome[a:a + 150]); lab.append(int(rng.integers(0, 1 << 20)))
    for i in range(130):                                            # 65..128 copies: the first size the global network takes
        seqs.append(genome[40:190]); lab.append(i * 3)
    ss = O.SeqSet.from_byte_seqs(seqs, data=np.array(lab), sizeof_d1=4)
    for min_obs, k in ((2, 47), (70000, 47), (1, 31), (72801, 47), (100, 21)):
        got, want = check(ctx, ss, k, min_obs, False, 4)
    got, want = check(ctx, ss, 47, 2, False, 4)
    assert int(np.diff(got.set_off).max()) > 300


def test_label_lists_multi_pass_bins(ctx):
    """low coverage: more distinct k-mers per bin than the LDS table holds -- bins re-streamed in hash-selected passes, each pass
    with its own stretch of the label buffer"""
    rng = np.random.default_rng(4)
    seqs = random_reads(rng, 1500, 400000, 150, False, err=0.0)
    lab = rng.integers(0, 1 << 24, size=len(seqs))
    ss = O.SeqSet.from_byte_seqs(seqs, data=lab, sizeof_d1=4)
    with ctx.options(DBG_FAST_TARGET="60000"):
        got, _ = check(ctx, ss, 47, 1, False, 4)
    assert len(got) > 100000


def test_label_lists_ragged_reads_and_exts(ctx):
    rng = np.random.default_rng(5)
    seqs = random_reads(rng, 800, 2500, 140, False, ragged=True)
    seqs += [np.zeros(0, np.uint8), R.random_dna(rng, 47), R.random_dna(rng, 46), R.random_dna(rng, 48)]
    exts = rng.integers(0, 256, size=len(seqs))
    lab = rng.integers(0, 3000, size=len(seqs)) * 1000
    ss = O.SeqSet.from_byte_seqs(seqs, exts=exts, data=lab, sizeof_d1=4)
    for stranded in (False, True):
        check(ctx, ss, 47, 1, stranded, 4)
        check(ctx, ss, 32, 2, stranded, 4)


def test_label_lists_equal_label_groups_and_generic(ctx):
    """65..1024 labels: the three routes (lists, label groups, generic sort) give the same table"""
    hs = dbg.synth_reads_host(n_reads=20000, read_len=150, error_rate=0.003, stranded=False, n_colours=4)
    rng = np.random.default_rng(77)
    lab = (rng.integers(0, 400, size=len(hs.start)) * 91).astype(np.uint16)
    seqs = dbg.HostSeqs(hs.words, hs.start, hs.length, None, lab, 2)
    tabs = []
    for opts in (dict(DBG_LABEL_LISTS="1", DBG_PATH="fast"), dict(DBG_LABEL_LISTS="0", DBG_PATH="fast"), dict(DBG_PATH="generic")):
        with ctx.options(**opts):
            t, _ = dbg.filter_kmers(seqs, dbg.CountFilterSet(2), False, False, 4, k=47, ctx=ctx)
        tabs.append(t)
    for t in tabs[1:]:
        assert np.array_equal(t.key_hi, tabs[0].key_hi) and np.array_equal(t.key_lo, tabs[0].key_lo) and np.array_equal(t.exts, tabs[0].exts)
        assert np.array_equal(t.set_off, tabs[0].set_off) and np.array_equal(t.set_val, tabs[0].set_val)
    assert len(tabs[0]) > 10000


@pytest.mark.parametrize("knobs", [dict(DBG_SCAN="wave"), dict(DBG_FAST_NO_SLAB="1"), dict(DBG_SLAB_CAP="4"), dict(DBG_NO_REC16="1"),
                                   dict(DBG_FAST_TARGET="400"), dict(DBG_NO_STRAND_NORM="1")])
def test_label_lists_other_routes_of_the_scan_and_sort(ctx, knobs):
    """the record's label word through every way a record can take: the wave-per-read scanner, reads longer than the lane-per-read
    scanner takes (1024 bases), no slabs / tiny slabs (read-order buffer + scatter of 4- and 5-word records), the three-array sort,
    tiny bins"""
    rng = np.random.default_rng(31)
    seqs = random_reads(rng, 900, 3000, 150, False)
    genome = R.random_dna(rng, 5000)
    seqs += [genome[a:a + int(rng.integers(1100, 1700))] for a in rng.integers(0, 3000, size=40)]          # long reads
    lab = rng.integers(0, 1 << 24, size=len(seqs))
    lab[:300] = rng.integers(0, 7, size=300) * 65537                   # some labels shared by many reads
    ss = O.SeqSet.from_byte_seqs(seqs, data=lab, sizeof_d1=4)
    with ctx.options(**knobs):
        for k in (31, 47, 63):
            check(ctx, ss, k, 1, False, 4)


def test_label_lists_into_scmap_compress(ctx):
    """the pipeline of BASELINE config 5 with a large alphabet: filter_kmers(CountFilterSet) as label lists, table left in HBM ->
    dbg_compress_table_dev with ScmapCompress (label lists -> class ids by verified hash, compression.rs:68-98) -> the oracle's
    compress_kmers over the same table with the classes numbered alike"""
    import ctypes as C
    from graph_canon import graphs_equal
    from pkg import capi
    k = 47
    hs = dbg.synth_reads_host(n_reads=30000, read_len=150, genome_len=120000, error_rate=0.002, stranded=False, n_colours=4)
    lab = ((np.arange(len(hs.start)) // 40) * 4099 + 70000).astype(np.uint32)          # 750 labels >= 65536, 40 reads each
    seqs = dbg.HostSeqs(hs.words, hs.start, hs.length, None, lab, 4)
    ss, keep = dbg.upload_seqs(seqs, 0)
    with ctx.options(DBG_COMPRESS="device"):
        t = dbg.filter_kmers_dev(ss, dbg.CountFilterSet(2), False, k, ctx=ctx)
        try:
            got = dbg.compress_table_dev(False, dbg.ScmapCompress(), t, k, ctx=ctx)
            h = capi.KmerTable()
            ctx.check(ctx.lib.dbg_table_to_host(ctx.h, C.byref(t), C.byref(h)))
            th = dbg._table_from_c(h, k)
            ctx.lib.dbg_free_table(ctx.h, C.byref(h))
        finally:
            ctx.lib.dbg_free_table(ctx.h, C.byref(t))
    want_t = O.filter_kmers(O.SeqSet(hs.words, hs.start, hs.length, None, lab, 4), k, O.COUNT_FILTER_SET, 2, stranded=False)
    assert_tables_equal(th, want_t, True)
    sets = [tuple(int(x) for x in th.set_val[int(th.set_off[i]):int(th.set_off[i + 1])]) for i in range(len(th))]
    glob = sorted(set(sets))
    pos = {s: i for i, s in enumerate(glob)}
    og = O.compress_kmers(k, False, O.SPEC_SCMAP_EQ, th.key_hi, th.key_lo, th.exts, np.array([pos[s] for s in sets], dtype=np.uint32))
    assert sorted(got.classes) == glob and len(glob) > 500
    ga = dict(got.arrays())
    ga["data"] = np.array([pos[got.classes[int(c)]] for c in got.data], dtype=np.uint32)
    assert graphs_equal(ga, og.arrays())
    assert 100 < len(got) < len(th)
