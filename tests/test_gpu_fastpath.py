"""GPU parity of the fast counting path (super-k-mer bins + per-bin LDS hash tables, fastpath.hip)
against the CPU oracle, bit-exact, with DBG_PATH=fast so that a silent fall-back to the generic
radix path cannot make these tests pass.  Also cross-checks fast == generic on the same input."""
import os

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
    """every test of this file insists on the fast path (the wave-per-bin counting kernel of round 3, which these tests also ran
    with, was measured 1.5-1.7x slower and left the product in round 5: branch exp-wavecount)"""
    old = ctx.set_option("DBG_PATH", "fast")
    yield
    ctx.set_option("DBG_PATH", old)


def run_fast(ctx, ss, k, summarizer, min_obs, stranded, data_width=0):
    is_set = summarizer == O.COUNT_FILTER_SET
    want = O.filter_kmers(ss, k, summarizer, min_obs, stranded=stranded)
    summ = (dbg.CountFilterSet if is_set else dbg.CountFilter)(min_obs)
    got, _ = dbg.filter_kmers(to_host_seqs(ss, data_width), summ, stranded, False, 4, k=k, ctx=ctx)
    assert_tables_equal(got, want, is_set)
    return got


@pytest.mark.parametrize("k,stranded,min_obs", [
    (47, False, 2), (47, False, 1), (47, True, 1), (31, True, 1), (31, False, 2), (32, False, 1), (33, False, 2),
    (16, False, 1), (21, True, 2), (51, False, 2), (63, False, 2), (64, False, 1), (64, True, 1),
])
def test_fast_count_filter(ctx, k, stranded, min_obs):
    rng = np.random.default_rng(1000 + k * 3 + stranded)
    seqs = random_reads(rng, 600, 4000, 150, stranded)
    run_fast(ctx, O.SeqSet.from_byte_seqs(seqs), k, O.COUNT_FILTER, min_obs, stranded)


@pytest.mark.parametrize("k,ncol", [(47, 4), (31, 24), (63, 7)])
def test_fast_count_filter_set(ctx, k, ncol):
    rng = np.random.default_rng(k + ncol)
    seqs = random_reads(rng, 700, 3000, 150, False)
    data = rng.integers(0, ncol, size=len(seqs))
    ss = O.SeqSet.from_byte_seqs(seqs, data=data, sizeof_d1=1)
    run_fast(ctx, ss, k, O.COUNT_FILTER_SET, 2, False, data_width=1)


def test_fast_multi_pass_bins(ctx):
    """Low coverage: far more distinct k-mers per bin than the 2048-entry LDS table holds, so bins are
    re-streamed in hash-selected passes (and split recursively on overflow)."""
    rng = np.random.default_rng(4)
    seqs = random_reads(rng, 1500, 400000, 150, False, err=0.0)
    got = run_fast(ctx, O.SeqSet.from_byte_seqs(seqs), 47, O.COUNT_FILTER, 1, False)
    assert len(got) > 100000


def test_fast_heavy_repeat_and_saturation(ctx):
    rng = np.random.default_rng(6)
    rep = R.random_dna(rng, 60)
    seqs = [rep] * 70000 + random_reads(rng, 200, 2000, 150, True)
    ss = O.SeqSet.from_byte_seqs(seqs)
    got = run_fast(ctx, ss, 47, O.COUNT_FILTER, 2, True)
    assert int(got.count.max()) == 65535                      # u16 saturation (filter.rs:57)
    run_fast(ctx, ss, 47, O.COUNT_FILTER, 65536, True)        # min above the saturated count: nothing valid


def test_fast_low_complexity(ctx):
    seqs = [np.zeros(150, np.uint8)] * 30 + [np.tile(np.array([0, 3], np.uint8), 75)] * 20
    seqs += [np.tile(np.array([0, 1, 2, 3], np.uint8), 40)] * 10 + [R.from_ascii(R.DEGEN)] * 4
    for k, stranded in ((31, False), (32, False), (47, True), (21, False)):
        run_fast(ctx, O.SeqSet.from_byte_seqs(seqs), k, O.COUNT_FILTER, 1, stranded)


def test_fast_ragged_and_boundary_exts(ctx):
    rng = np.random.default_rng(5)
    seqs = random_reads(rng, 500, 2500, 140, False, ragged=True)
    seqs += [np.zeros(0, np.uint8), R.random_dna(rng, 47), R.random_dna(rng, 46), R.random_dna(rng, 48)]
    exts = rng.integers(0, 256, size=len(seqs))
    ss = O.SeqSet.from_byte_seqs(seqs, exts=exts)
    for stranded in (False, True):
        run_fast(ctx, ss, 47, O.COUNT_FILTER, 1, stranded)


def test_fast_equals_generic_on_synthetic_stream(ctx):
    hs = dbg.synth_reads_host(n_reads=30000, read_len=150, error_rate=0.002, stranded=False, n_colours=4)
    for summ in (dbg.CountFilter(2), dbg.CountFilterSet(2)):
        ctx.set_option("DBG_PATH", "fast")
        a, _ = dbg.filter_kmers(hs, summ, False, False, 4, k=47, ctx=ctx)
        ctx.set_option("DBG_PATH", "generic")
        b, _ = dbg.filter_kmers(hs, summ, False, False, 4, k=47, ctx=ctx)
        assert np.array_equal(a.key_hi, b.key_hi) and np.array_equal(a.key_lo, b.key_lo)
        assert np.array_equal(a.exts, b.exts)
        if summ.kind == 0:
            assert np.array_equal(a.count, b.count)
        else:
            assert np.array_equal(a.set_off, b.set_off) and np.array_equal(a.set_val, b.set_val)
        assert len(a) > 0


def test_fast_path_refuses_unsupported_shapes(ctx):
    """more than 1024 distinct labels: label lists (fast_labellists.hpp) since round 5; with DBG_LABEL_LISTS=0 the fast path
    refuses the shape and the default dispatch takes the generic (sort-based) CountFilterSet"""
    rng = np.random.default_rng(1)
    seqs = random_reads(rng, 1100, 1000, 150, False)
    ss = O.SeqSet.from_byte_seqs(seqs, data=np.arange(1100) * 7 + 40, sizeof_d1=2)
    want = O.filter_kmers(ss, 47, O.COUNT_FILTER_SET, 1, stranded=False)
    got, _ = dbg.filter_kmers(to_host_seqs(ss, 2), dbg.CountFilterSet(1), False, False, 4, k=47, ctx=ctx)
    assert_tables_equal(got, want, True)
    with ctx.options(DBG_LABEL_LISTS="0"):
        with pytest.raises(dbg.DbgError):
            dbg.filter_kmers(to_host_seqs(ss, 2), dbg.CountFilterSet(1), False, False, 4, k=47, ctx=ctx)
        ctx.set_option("DBG_PATH", "auto")
        got, _ = dbg.filter_kmers(to_host_seqs(ss, 2), dbg.CountFilterSet(1), False, False, 4, k=47, ctx=ctx)
    assert_tables_equal(got, want, True)


@pytest.mark.parametrize("n_labels,k,stranded,min_obs,report_all", [(65, 47, False, 2, False), (200, 31, True, 1, False), (1024, 63, False, 2, False),
                                                                    (130, 20, False, 3, True), (500, 51, False, 2, False)])
def test_fast_label_groups(ctx, n_labels, k, stranded, min_obs, report_all):
    """more than 64 distinct labels: one CountFilter run for the valid k-mers, one 64-colour run per label group joined into it
    (fast_manylabels.hpp).  Validity counts observations over ALL labels, the lists come out ascending."""
    hs = dbg.synth_reads_host(n_reads=5000, read_len=150, error_rate=0.004, stranded=stranded, n_colours=4)
    rng = np.random.default_rng(n_labels * 31 + k)
    alphabet = np.sort(rng.choice(np.arange(0, 65536), size=n_labels, replace=False)).astype(np.uint16)
    lab = alphabet[rng.integers(0, n_labels, size=len(hs.start))]
    lab[:n_labels] = alphabet                                      # every label occurs
    ss = O.SeqSet(hs.words, hs.start, hs.length, None, lab, 2)
    want = O.filter_kmers(ss, k, O.COUNT_FILTER_SET, min_obs, stranded=stranded, report_all=report_all)
    with ctx.options(DBG_DEBUG="1", DBG_LABEL_LISTS="0"):          # (label lists are the default since round 5: test_gpu_labellists.py)
        got, allk = dbg.filter_kmers(to_host_seqs(ss, 2), dbg.CountFilterSet(min_obs), stranded, report_all, 4, k=k, ctx=ctx)
    assert len(got) > 1000
    assert_tables_equal(got, want, True)
    if report_all:
        assert np.array_equal(got.all_hi, want.all_hi) and np.array_equal(got.all_lo, want.all_lo)
    with ctx.options(DBG_NO_LABEL_GROUPS="1", DBG_PATH="auto"):    # the same through the generic path
        gen, _ = dbg.filter_kmers(to_host_seqs(ss, 2), dbg.CountFilterSet(min_obs), stranded, report_all, 4, k=k, ctx=ctx)
    assert_tables_equal(gen, want, True)


def test_label_groups_threshold_above_the_saturated_count(ctx):
    """CountFilterSet::summarize compares the UNSATURATED number of observations with min_kmer_obs (filter.rs:85-100).  The
    label-group passes take their valid k-mers from a CountFilter run, whose validity test reads the u16-saturated count, so a
    threshold above 65535 must not go that way (round-3 advisor finding: it returned an empty table): one k-mer seen 72 800
    times under 100 labels, min_kmer_obs = 70 000."""
    rng = np.random.default_rng(3)
    genome = R.random_dna(rng, 2000)
    seqs = [[0] * 150 for _ in range(700)]                          # poly-A: the k-mer A^47, 104 times per read
    lab = [int(x) for x in rng.integers(0, 100, 700) * 37]
    for i in range(300):                                            # ordinary reads, far below the threshold
        a = int(rng.integers(0, 1850))
        seqs.append(genome[a:a + 150]); lab.append(int(rng.integers(0, 100)) * 37)
    ss = O.SeqSet.from_byte_seqs(seqs, data=np.array(lab), sizeof_d1=2)
    for min_obs in (70000, 65536, 72801):
        want = O.filter_kmers(ss, 47, O.COUNT_FILTER_SET, min_obs, stranded=False)
        assert want.n == (1 if min_obs <= 72800 else 0)
        for lists in ("0", "1"):
            with ctx.options(DBG_PATH="auto", DBG_LABEL_LISTS=lists):
                got, _ = dbg.filter_kmers(to_host_seqs(ss, 2), dbg.CountFilterSet(min_obs), False, False, 4, k=47, ctx=ctx)
            assert_tables_equal(got, want, True)


def test_fast_label_groups_uneven(ctx):
    """one group holds nearly all reads, one a single read, one only reads shorter than k; k-mers shared by every group"""
    rng = np.random.default_rng(9)
    genome = R.random_dna(rng, 3000)
    seqs, lab = [], []
    for i in range(1500):
        a = int(rng.integers(0, 2850))
        seqs.append(genome[a:a + 150])
        lab.append(int(rng.integers(0, 64)) * 3)                    # group 0: labels 0..189 step 3
    seqs.append(genome[100:250]); lab.append(40000)                 # a lone read in the last group
    for i in range(70):                                             # 70 more labels, on reads too short to hold a k-mer
        seqs.append(genome[5 * i:5 * i + 30]); lab.append(1000 + i)
    for i in range(6):                                              # and a third populated group
        seqs.append(genome[200 * i:200 * i + 190]); lab.append(30000 + i)
    ss = O.SeqSet.from_byte_seqs(seqs, data=np.array(lab), sizeof_d1=2)
    for min_obs in (1, 2, 40):
        want = O.filter_kmers(ss, 47, O.COUNT_FILTER_SET, min_obs, stranded=False)
        for lists in ("0", "1"):
            with ctx.options(DBG_LABEL_LISTS=lists):
                got, _ = dbg.filter_kmers(to_host_seqs(ss, 2), dbg.CountFilterSet(min_obs), False, False, 4, k=47, ctx=ctx)
            assert_tables_equal(got, want, True)


@pytest.mark.parametrize("k,stranded,kind", [(47, False, 0), (31, True, 0), (64, False, 0), (47, False, 1), (16, False, 0)])
def test_fast_report_all_kmers(ctx, k, stranded, kind):
    """report_all_kmers (filter.rs:208-212) through the fast path: every distinct k-mer, ascending, next to the valid table."""
    hs = dbg.synth_reads_host(n_reads=3000, read_len=150, error_rate=0.004, stranded=stranded, n_colours=4)
    summ = dbg.CountFilterSet(2) if kind else dbg.CountFilter(2)
    t, allk = dbg.filter_kmers(hs, summ, stranded, True, 4, k=k, ctx=ctx)
    ss = O.SeqSet(hs.words, hs.start, hs.length, None, hs.data if kind else None, 1 if kind else 0)
    want = O.filter_kmers(ss, k, kind, 2, stranded=stranded, report_all=True)
    assert np.array_equal(t.key_hi, want.key_hi) and np.array_equal(t.key_lo, want.key_lo) and np.array_equal(t.exts, want.exts)
    assert np.array_equal(t.all_hi, want.all_hi) and np.array_equal(t.all_lo, want.all_lo)
    assert len(allk) == len(want.all_lo) > len(t)


@pytest.mark.parametrize("k", [31, 47, 64])
def test_fast_long_sequences(ctx, k):
    """Contig-length inputs: many 128-window tiles per sequence (minimizer runs and pieces carried across tile borders),
    plus a long low-complexity stretch whose minimizer value never changes (pieces cut every W windows)."""
    rng = np.random.default_rng(77 + k)
    seqs = [R.random_dna(rng, int(n)) for n in (3000, 5000, 1234, 129 + k, 128 + k, 127 + k, 257 + k)]
    rep = np.tile(R.random_dna(rng, 5), 400)                       # period-5 repeat, 2000 bp
    seqs += [rep, np.concatenate([R.random_dna(rng, 300), np.zeros(900, np.uint8), R.random_dna(rng, 300)])]
    seqs = seqs + [R.revcomp_bytes(s) for s in seqs[:4]]
    for stranded in (False, True):
        ss = O.SeqSet.from_byte_seqs(seqs)
        want = O.filter_kmers(ss, k, O.COUNT_FILTER, 1, stranded=stranded)
        got, _ = dbg.filter_kmers(to_host_seqs(ss, 0), dbg.CountFilter(1), stranded, False, 4, k=k, ctx=ctx)
        assert_tables_equal(got, want, False)


@pytest.mark.parametrize("k", [16, 17, 20, 21, 22, 23, 32, 33, 34, 35, 36, 48, 49, 50, 51, 52, 63, 64])
def test_fast_k_sweep(ctx, k):
    """Every k at which the internal layout changes: minimizer length (k = 21, 23), key width (33), words per
    super-k-mer record incl. its 20 meta bits (35, 51)."""
    hs = dbg.synth_reads_host(n_reads=4000, read_len=150, error_rate=0.003, stranded=False, n_colours=4)
    for kind, stranded in ((0, False), (1, False), (0, True)):
        ss = O.SeqSet(hs.words, hs.start, hs.length, None, hs.data if kind else None, 1 if kind else 0)
        want = O.filter_kmers(ss, k, kind, 2, stranded=stranded)
        summ = dbg.CountFilterSet(2) if kind else dbg.CountFilter(2)
        got, _ = dbg.filter_kmers(hs if kind else dbg.HostSeqs(hs.words, hs.start, hs.length), summ, stranded, False, 4, k=k, ctx=ctx)
        assert_tables_equal(got, want, kind == 1)


@pytest.mark.parametrize("env", [{"DBG_NO_HYBRID_SORT": "1"}, {"DBG_NO_REC16": "1"}, {"DBG_NO_REC16": "1", "DBG_NO_HYBRID_SORT": "1"},
                                 {"DBG_FAST_NO_SLAB": "1"}, {"DBG_SORT": "bytealigned"}, {"DBG_ONESWEEP": "0"}])
@pytest.mark.parametrize("k,kind", [(47, 1), (31, 0), (63, 0)])
def test_fast_sort_variants(ctx, env, k, kind):
    """DBG_FAST_NO_SLAB: what happens when the slabs do not fit in memory (every record through the read-order buffer and
    the scatter pass).  The order-restoring sort has four code paths (16-byte records or three arrays; prefix passes + LDS finisher or the
    plain LSD sort that also serves as its fall-back): all must give the same table."""
    hs = dbg.synth_reads_host(n_reads=3000, read_len=150, error_rate=0.003, stranded=False, n_colours=4)
    ss = O.SeqSet(hs.words, hs.start, hs.length, None, hs.data if kind else None, 1 if kind else 0)
    want = O.filter_kmers(ss, k, kind, 2, stranded=False)
    with ctx.options(**env):
        summ = dbg.CountFilterSet(2) if kind else dbg.CountFilter(2)
        got, _ = dbg.filter_kmers(hs if kind else dbg.HostSeqs(hs.words, hs.start, hs.length), summ, False, False, 4, k=k, ctx=ctx)
    assert_tables_equal(got, want, kind == 1)


@pytest.mark.parametrize("grid", ["0", "1", "3", "5000"])
@pytest.mark.parametrize("k,kind,labels", [(47, 1, 4), (31, 0, 0), (63, 1, 40), (47, 1, 3000)], ids=["k47-set4", "k31-count", "k63-set40", "k47-lists"])
def test_fast_resident_workgroups(ctx, grid, k, kind, labels):
    """The counting kernels (bin_count, bin_labels) run as resident workgroups that take bins from a ticket counter (round 6).  Whatever the
    grid -- one workgroup per bin (DBG_FAST_PERSIST=0, the old launch), a single resident workgroup, three, or more than there are bins --
    the table is the oracle's; many small bins (DBG_FAST_TARGET) so that every workgroup goes round its loop many times."""
    hs = dbg.synth_reads_host(n_reads=4000, read_len=150, error_rate=0.003, stranded=False, n_colours=max(min(labels, 200), 1))
    width, data = 0, None
    if kind:                                                         # 200 distinct 16-bit labels: the label-list kernel
        width = 1 if labels <= 255 else 2
        data = np.asarray(hs.data) if width == 1 else (np.asarray(hs.data).astype(np.uint32) * 300 + 7).astype(np.uint16)
    ss = O.SeqSet(hs.words, hs.start, hs.length, None, data, width)
    want = O.filter_kmers(ss, k, kind, 2, stranded=False)
    with ctx.options(DBG_FAST_PERSIST=grid, DBG_FAST_TARGET="600"):
        summ = dbg.CountFilterSet(2) if kind else dbg.CountFilter(2)
        got, _ = dbg.filter_kmers(dbg.HostSeqs(hs.words, hs.start, hs.length, None, data, width), summ, False, False, 4, k=k, ctx=ctx)
    assert_tables_equal(got, want, kind == 1)


def _with_env(ctx, env, fn):
    with ctx.options(**env):
        return fn()


@pytest.mark.parametrize("k", [31, 47, 63])
def test_fast_duplicate_records_weights_and_colours(ctx, k):
    """bin_count counts identical super-k-mer records of a batch once (weight + colour union).  Hundreds of copies of a
    few reads, on both strands and in every colour, make weights of several hundred per staged record; the counts
    (saturating u16 for CountFilter), the Exts and the label sets must still be the reference's."""
    rng = np.random.default_rng(77 + k)
    base = [R.random_dna(rng, 150) for _ in range(6)]
    seqs, data = [], []
    for i in range(2400):
        b = base[int(rng.integers(0, 6))] if i % 9 else R.random_dna(rng, 150)
        if rng.random() < 0.5:
            b = (3 - b)[::-1].copy()
        seqs.append(b)
        data.append(int(rng.integers(0, 24)))
    for target in ("8000", "200000"):
        _with_env(ctx, {"DBG_FAST_TARGET": target}, lambda: (
            run_fast(ctx, O.SeqSet.from_byte_seqs(seqs), k, O.COUNT_FILTER, 2, False),
            run_fast(ctx, O.SeqSet.from_byte_seqs(seqs, data=data, sizeof_d1=1), k, O.COUNT_FILTER_SET, 3, False, data_width=1),
            run_fast(ctx, O.SeqSet.from_byte_seqs(seqs), k, O.COUNT_FILTER, 1, True)))


@pytest.mark.parametrize("k,kind", [(63, 0), (63, 1), (51, 1), (47, 0)])
def test_fast_large_bins_of_distinct_records(ctx, k, kind):
    """Bins far larger than a batch whose records are all different: the distinct records of a batch need more chunks than
    the chunk map holds (the duplicate filter is undone for that batch and the rest is staged again), and the table
    overflows into hash-selected passes."""
    rng = np.random.default_rng(5 + k + kind)
    seqs = random_reads(rng, 2500, 2000000, 150, False, err=0.0)
    data = rng.integers(0, 5, size=len(seqs)) if kind else None
    ss = O.SeqSet.from_byte_seqs(seqs, data=data, sizeof_d1=1) if kind else O.SeqSet.from_byte_seqs(seqs)
    _with_env(ctx, {"DBG_FAST_TARGET": "60000"},
              lambda: run_fast(ctx, ss, k, O.COUNT_FILTER_SET if kind else O.COUNT_FILTER, 1, False, data_width=1 if kind else 0))


def test_fast_large_bins_high_coverage(ctx):
    hs = dbg.synth_reads_host(n_reads=6000, read_len=150, error_rate=0.002, stranded=False, n_colours=4)
    ss = O.SeqSet(hs.words, hs.start, hs.length, None, hs.data, 1)
    want = O.filter_kmers(ss, 47, O.COUNT_FILTER_SET, 2, stranded=False)
    got = _with_env(ctx, {"DBG_FAST_TARGET": "40000"}, lambda: dbg.filter_kmers(hs, dbg.CountFilterSet(2), False, False, 4, k=47, ctx=ctx)[0])
    assert_tables_equal(got, want, True)


@pytest.mark.parametrize("k,kind", [(31, 0), (47, 1), (47, 0), (63, 1), (63, 0)])
@pytest.mark.parametrize("shape", ["uniform", "groups_200", "groups_1000"])
def test_fast_sort_group_finisher_regimes(ctx, k, kind, shape, capfd):
    """The order-restoring sort picks its finisher from the table size: between 16 and 32 k-mers per group of equal prefix
    (4096 .. 8192 valid k-mers for an 8-bit prefix -- and 2.7e8 .. 5.4e8 for the 24 bits of the benchmark) a wave sorts each
    group; a group of more than 128 sends the call back to the walking finisher, one of more than 512 to the plain LSD sort.
    Reads made of 'AAAA' + four random bases put every eighth k-mer into the prefix group of AAAA."""
    rng = np.random.default_rng(900 + k + kind + len(shape))
    n_skew = {"uniform": 0, "groups_200": 12, "groups_1000": 60}[shape]
    seqs = [R.random_dna(rng, 150) for _ in range(52 - min(n_skew, 40))]
    for _ in range(n_skew):
        blocks = [np.concatenate([np.zeros(4, np.uint8), R.random_dna(rng, 4)]) for _ in range(19)]
        seqs.append(np.concatenate(blocks)[:150])
    data = rng.integers(0, 6, size=len(seqs)) if kind else None
    ss = O.SeqSet.from_byte_seqs(seqs, data=data, sizeof_d1=1) if kind else O.SeqSet.from_byte_seqs(seqs)
    with ctx.options(DBG_DEBUG="1"):
        got = run_fast(ctx, ss, k, O.COUNT_FILTER_SET if kind else O.COUNT_FILTER, 1, False, data_width=1 if kind else 0)
    err = capfd.readouterr().err
    gave_up = [int(l.split("form ")[1][0]) for l in err.splitlines() if l.startswith("[sort] finisher form")]
    if 4096 <= len(got) <= 8192:                              # the size range in which a wave sorts each group
        assert gave_up in {"uniform": ([],), "groups_200": ([0],), "groups_1000": ([0], [0, 1])}[shape], err


@pytest.mark.parametrize("k,width,labels", [(47, 1, [31, 40, 77, 200, 255]), (31, 2, list(range(1000, 60000, 2600))), (63, 4, [24, 25, 26, 65535]),
                                            (47, 1, list(range(100, 124)))])
def test_fast_sparse_label_alphabet(ctx, k, width, labels):
    """CountFilterSet labels >= 24: the LDS colour mask has 24 bits, so an alphabet of at most 24 distinct labels (< 65536) is
    mapped to colour indices in label order and mapped back when the label sets are written (filter.rs:85-100 sorts and
    de-duplicates the labels, so order is all that has to survive)."""
    assert len(labels) <= 24
    rng = np.random.default_rng(k + width)
    seqs = random_reads(rng, 600, 3000, 150, False)
    data = np.asarray(labels)[rng.integers(0, len(labels), size=len(seqs))]
    ss = O.SeqSet.from_byte_seqs(seqs, data=data, sizeof_d1=width)
    run_fast(ctx, ss, k, O.COUNT_FILTER_SET, 2, False, data_width=width)


@pytest.mark.parametrize("k,width,labels", [(47, 1, list(range(40))), (31, 1, list(range(64))), (63, 2, list(range(500, 30000, 470))),
                                            (51, 1, [0, 23, 24, 31, 32, 33, 62, 63]), (33, 4, list(range(7, 70, 1)))])
def test_fast_wide_colour_sets(ctx, k, width, labels):
    """CountFilterSet with 25..64 colours (labels < 64, or a sparse alphabet of at most 64 distinct labels < 65536): the WIDE
    layout of the counting kernel -- two colour words + an Exts byte per table entry, the sort carries the record's position and
    Exts / colours are gathered afterwards -- must give the reference's sorted, de-duplicated label lists (filter.rs:85-100).
    Several copies of every read in different colours make multi-label sets the rule."""
    rng = np.random.default_rng(k + width + len(labels))
    base = random_reads(rng, 250, 2500, 150, False)
    seqs = [b for b in base for _ in range(4)]
    data = np.asarray(labels)[rng.integers(0, len(labels), size=len(seqs))]
    ss = O.SeqSet.from_byte_seqs(seqs, data=data, sizeof_d1=width)
    got = run_fast(ctx, ss, k, O.COUNT_FILTER_SET, 2, False, data_width=width)
    sizes = np.diff(got.set_off)
    assert int(sizes.max()) >= 3 and int(got.set_val.max()) == max(labels)
    if len(labels) > 24:
        assert int(got.set_val.max()) >= 24                                   # the wide layout was needed


def test_more_than_1024_labels(ctx):
    """more than 1024 distinct labels: label lists on the fast path (round 5, test_gpu_labellists.py); with DBG_LABEL_LISTS=0 the generic
    path under the default dispatch, and DBG_PATH=fast must not fall back silently"""
    rng = np.random.default_rng(3)
    seqs = random_reads(rng, 1500, 2000, 150, False)
    data = rng.permutation(1500) * 40 + 7
    ss = O.SeqSet.from_byte_seqs(seqs, data=data, sizeof_d1=2)
    want = O.filter_kmers(ss, 47, O.COUNT_FILTER_SET, 1, stranded=False)
    got, _ = dbg.filter_kmers(to_host_seqs(ss, 2), dbg.CountFilterSet(1), False, False, 4, k=47, ctx=ctx)
    assert_tables_equal(got, want, True)
    with ctx.options(DBG_LABEL_LISTS="0"):
        with ctx.options(DBG_PATH="auto"):
            got, _ = dbg.filter_kmers(to_host_seqs(ss, 2), dbg.CountFilterSet(1), False, False, 4, k=47, ctx=ctx)
        assert_tables_equal(got, want, True)
        with pytest.raises(dbg.DbgError):
            dbg.filter_kmers(to_host_seqs(ss, 2), dbg.CountFilterSet(1), False, False, 4, k=47, ctx=ctx)


@pytest.mark.parametrize("k", [64, 63, 48, 32])
@pytest.mark.parametrize("n_valid_target", [1, 5, 25, 40])
def test_fast_tiny_tables_are_sorted(ctx, k, n_valid_target):
    """A table of at most 32 valid k-mers takes no prefix pass at all: the finisher sees one group whose prefix has zero bits.  At
    k = 64 that prefix was computed as `hi >> 64` -- not a shift -- and the table came back unsorted (found by tests/test_gpu_fuzz.py,
    seed 2066).  Few valid k-mers out of many: a handful of reads repeated three times among singletons, CountFilter(3)."""
    rng = np.random.default_rng(k * 100 + n_valid_target)
    rep = R.random_dna(rng, k + n_valid_target - 1)
    seqs = [rep] * 3 + random_reads(rng, 25, 400000, 150, False, err=0.0)      # (a genome that sparse: no other k-mer is seen three times)
    ss = O.SeqSet.from_byte_seqs(seqs)
    got = run_fast(ctx, ss, k, O.COUNT_FILTER, 3, False)
    assert len(got) == n_valid_target


@pytest.mark.parametrize("k,kind", [(47, 0), (31, 1), (63, 0), (21, 1)])
def test_fast_strand_normalised_records(ctx, k, kind):
    """Non-stranded counting with odd k stores every super-k-mer piece as the smaller of (piece, reverse complement), so that the
    copies reads of the two strands produce are identical records (they merge in the counting workgroup and in the sender-side
    merge).  Same table with and without (DBG_NO_STRAND_NORM), on reads from both strands, with read Exts at the ends."""
    hs = dbg.synth_reads_host(n_reads=4000, read_len=150, error_rate=0.003, stranded=False, n_colours=5)
    rng = np.random.default_rng(k)
    exts = rng.integers(0, 256, size=len(hs.start)).astype(np.uint8)
    ss = O.SeqSet(hs.words, hs.start, hs.length, exts, hs.data if kind else None, 1 if kind else 0)
    want = O.filter_kmers(ss, k, kind, 2, stranded=False)
    summ = (dbg.CountFilterSet if kind else dbg.CountFilter)(2)
    h = dbg.HostSeqs(hs.words, hs.start, hs.length, exts, hs.data if kind else None, 1 if kind else 0)
    got, _ = dbg.filter_kmers(h, summ, False, False, 4, k=k, ctx=ctx)
    assert_tables_equal(got, want, bool(kind))
    with ctx.options(DBG_NO_STRAND_NORM="1"):
        plain, _ = dbg.filter_kmers(h, summ, False, False, 4, k=k, ctx=ctx)
    assert_tables_equal(plain, want, bool(kind))
