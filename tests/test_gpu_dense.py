"""GPU parity of the dense path (densepath.hip: directly addressed count / Exts / label-mask tables for 4 <= k <= 15, emitted
in key order by one compaction pass) against the CPU oracle, bit-exact, with DBG_PATH=dense so that a silent fall-back to the
generic sort-based path cannot make these tests pass.  Mirrors the reference's use of filter_kmers (src/filter.rs:139-231)."""
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
def force_dense(ctx):
    old = ctx.set_option("DBG_PATH", "dense")
    yield
    ctx.set_option("DBG_PATH", old)


def run_dense(ctx, ss, k, summarizer, min_obs, stranded, report_all=True, data_width=0):
    is_set = summarizer == O.COUNT_FILTER_SET
    want = O.filter_kmers(ss, k, summarizer, min_obs, stranded=stranded, report_all=report_all)
    summ = (dbg.CountFilterSet if is_set else dbg.CountFilter)(min_obs)
    got, _ = dbg.filter_kmers(to_host_seqs(ss, data_width), summ, stranded, report_all, 4, k=k, ctx=ctx)
    assert_tables_equal(got, want, is_set)
    return got


@pytest.mark.parametrize("k", list(range(4, 16)))
@pytest.mark.parametrize("stranded", [False, True])
def test_dense_count_filter(ctx, k, stranded):
    rng = np.random.default_rng(40 * k + stranded)
    seqs = random_reads(rng, 300, 4 ** min(k, 6), 150, stranded, ragged=(k % 3 == 0))
    seqs += [np.zeros(150, np.uint8), np.full(150, 3, np.uint8), np.tile(np.array([0, 3], np.uint8), 75)]   # homopolymers, a palindromic repeat
    ss = O.SeqSet.from_byte_seqs(seqs)
    for min_obs in (1, 2):
        run_dense(ctx, ss, k, O.COUNT_FILTER, min_obs, stranded)


@pytest.mark.parametrize("k,width,labels", [(15, 1, list(range(4))), (11, 1, list(range(64))), (8, 2, [0, 7, 63]), (5, 4, list(range(30))), (13, 1, [5])])
def test_dense_count_filter_set(ctx, k, width, labels):
    rng = np.random.default_rng(k + len(labels))
    seqs = random_reads(rng, 500, 3000, 150, False)
    data = np.asarray(labels)[rng.integers(0, len(labels), size=len(seqs))]
    ss = O.SeqSet.from_byte_seqs(seqs, data=data, sizeof_d1=width)
    for min_obs in (1, 3):
        run_dense(ctx, ss, k, O.COUNT_FILTER_SET, min_obs, False, data_width=width)


def test_dense_saturation_and_boundary_exts(ctx):
    """u16 saturation of the count (filter.rs:57) on 70 000 copies of one read; non-empty seq_exts land on the first / last
    k-mer only; sequences shorter than k contribute nothing."""
    rng = np.random.default_rng(6)
    rep = R.random_dna(rng, 40)
    seqs = [rep] * 70000 + random_reads(rng, 100, 500, 60, True, ragged=True)
    exts = rng.integers(0, 256, size=len(seqs)).astype(np.uint8)
    ss = O.SeqSet.from_byte_seqs(seqs, exts=exts)
    got = run_dense(ctx, ss, 12, O.COUNT_FILTER, 2, True)
    assert int(got.count.max()) == 65535
    run_dense(ctx, ss, 12, O.COUNT_FILTER, 65536, True)          # min above the saturated count: nothing valid
    got = run_dense(ctx, ss, 7, O.COUNT_FILTER, 1, False)       # (k <= 7: counted in per-workgroup LDS tables first)
    assert int(got.count.max()) == 65535


def test_dense_equals_generic_and_refuses_other_shapes(ctx):
    hs = dbg.synth_reads_host(n_reads=20000, read_len=150, error_rate=0.002, stranded=False, n_colours=4)
    for summ in (dbg.CountFilter(2), dbg.CountFilterSet(2)):
        a, _ = dbg.filter_kmers(hs, summ, False, True, 4, k=13, ctx=ctx)
        with ctx.options(DBG_PATH="generic"):
            b, _ = dbg.filter_kmers(hs, summ, False, True, 4, k=13, ctx=ctx)
        assert np.array_equal(a.key_lo, b.key_lo) and np.array_equal(a.exts, b.exts) and np.array_equal(a.all_lo, b.all_lo)
        if summ.kind == 0:
            assert np.array_equal(a.count, b.count)
        else:
            assert np.array_equal(a.set_off, b.set_off) and np.array_equal(a.set_val, b.set_val)
        assert len(a) > 0
    with pytest.raises(dbg.DbgError):                            # k = 16 is the super-k-mer path's
        dbg.filter_kmers(hs, dbg.CountFilter(1), False, False, 4, k=16, ctx=ctx)
    rng = np.random.default_rng(2)
    seqs = random_reads(rng, 100, 1000, 150, False)
    ss = O.SeqSet.from_byte_seqs(seqs, data=rng.integers(0, 200, size=len(seqs)), sizeof_d1=1)
    with pytest.raises(dbg.DbgError):                            # labels >= 64
        dbg.filter_kmers(to_host_seqs(ss, 1), dbg.CountFilterSet(1), False, False, 4, k=12, ctx=ctx)


@pytest.mark.parametrize("k", [8, 9, 12, 15])
def test_dense_partitioned_batches_and_forms(ctx, k):
    """8 <= k <= 15 count in LDS after partitioning the k-mer instances by their key's top bits (one level for k <= 11, two above).  Reads
    are taken in batches that bound the instance buffers: with batches of ~5 000 instances (many batches accumulating in one table), with
    one batch, and in the device-atomic form (DBG_DENSE_PART=0) the table is the oracle's; ragged reads, homopolymers (one hot partition),
    labels."""
    rng = np.random.default_rng(900 + k)
    seqs = random_reads(rng, 400, 2000, 150, False, ragged=True)
    seqs += [np.zeros(700, np.uint8), np.full(333, 3, np.uint8), np.tile(np.array([0, 1, 2, 3], np.uint8), 100)]
    data = rng.integers(0, 40, size=len(seqs))
    for is_set in (False, True):
        ss = O.SeqSet.from_byte_seqs(seqs, data=data if is_set else None, sizeof_d1=1 if is_set else 0)
        summ = O.COUNT_FILTER_SET if is_set else O.COUNT_FILTER
        # (DBG_DENSE_RAW=0: level 1 walks the reads inside the scatter instead of extracting into a raw buffer first; k = 8 without partitions:
        #  key-range passes in LDS, or atomics)
        for opts in (dict(DBG_DENSE_BATCH="5000"), dict(), dict(DBG_DENSE_RAW="0"), dict(DBG_DENSE_RAW="0", DBG_DENSE_BATCH="7000"), dict(DBG_DENSE_PART="0"),
                     dict(DBG_DENSE_PART="0", DBG_DENSE_RANGES="0")):
            with ctx.options(**opts):
                run_dense(ctx, ss, k, summ, 2, False, data_width=1 if is_set else 0)
