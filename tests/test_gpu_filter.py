"""GPU parity: dbg_filter_kmers (HIP, through the C ABI) vs the CPU oracle, bit-exact.

Mirrors the reference's own use of filter_kmers (src/test.rs:238-244, :344-350, :450-456):
same arguments, and the oracle's ascending-key vectors are the expected output.
"""
import numpy as np
import pytest

import oracle_lib as O
import refgen as R
from pkg import dbg, capi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = dbg.Context(0)
    yield c
    c.close()


def to_host_seqs(ss, data_width=0):
    return dbg.HostSeqs(ss.words, ss.start, ss.length, ss.exts, ss.data, data_width if ss.data is not None else 0)


def assert_tables_equal(got, want, is_set):
    assert len(got) == want.n
    assert np.array_equal(got.key_hi, want.key_hi)
    assert np.array_equal(got.key_lo, want.key_lo)
    assert np.array_equal(got.exts, want.exts)
    if is_set:
        assert np.array_equal(got.set_off, want.set_off)
        assert np.array_equal(got.set_val, want.set_val)
    else:
        assert np.array_equal(got.count, want.count)
    assert np.array_equal(got.all_hi, want.all_hi)
    assert np.array_equal(got.all_lo, want.all_lo)


@pytest.fixture(autouse=True, params=["auto", "generic"])
def path_mode(request, ctx):
    """Every test runs through the default dispatch (fast path where it applies) and with the generic
    extract -> radix sort -> reduce path forced."""
    import os
    old = ctx.set_option("DBG_PATH", request.param)
    yield request.param
    ctx.set_option("DBG_PATH", old)


def run_both(ctx, ss, k, summarizer, min_obs, stranded, report_all=False, data_width=0):
    is_set = summarizer == O.COUNT_FILTER_SET
    want = O.filter_kmers(ss, k, summarizer, min_obs, stranded=stranded, report_all=report_all)
    summ = (dbg.CountFilterSet if is_set else dbg.CountFilter)(min_obs)
    got, _ = dbg.filter_kmers(to_host_seqs(ss, data_width), summ, stranded, report_all, 4, k=k, ctx=ctx)
    assert_tables_equal(got, want, is_set)
    return got, want


def random_reads(rng, n_reads, genome_len, read_len, stranded, err=0.01, ragged=False):
    genome = R.random_dna(rng, genome_len)
    seqs = []
    for _ in range(n_reads):
        ln = int(rng.integers(0, read_len + 1)) if ragged else read_len
        st = int(rng.integers(0, genome_len - ln + 1))
        s = genome[st:st + ln].copy()
        m = rng.random(ln) < err
        s[m] = (s[m] + rng.integers(1, 4, size=int(m.sum()))) % 4
        if not stranded and rng.random() < 0.5:
            s = R.revcomp_bytes(s)
        seqs.append(s.astype(np.uint8))
    return seqs


@pytest.mark.parametrize("k,stranded,min_obs", [
    (31, True, 1),       # BASELINE config 1 shape: k=31 stranded CountFilter(1)
    (31, False, 2), (32, False, 1), (33, False, 2), (47, False, 2), (47, True, 1), (51, False, 1),
    (63, False, 2), (64, False, 1), (16, False, 1), (4, True, 1), (5, False, 1),
])
def test_count_filter_parity(ctx, k, stranded, min_obs):
    rng = np.random.default_rng(k * 10 + stranded)
    seqs = random_reads(rng, 400, 3000, 150, stranded)
    ss = O.SeqSet.from_byte_seqs(seqs)
    run_both(ctx, ss, k, O.COUNT_FILTER, min_obs, stranded, report_all=True)


@pytest.mark.parametrize("k,width", [(47, 1), (31, 1), (63, 2), (47, 4)])
def test_count_filter_set_parity(ctx, k, width):
    rng = np.random.default_rng(77 + k + width)
    seqs = random_reads(rng, 500, 2500, 150, False)
    data = rng.integers(0, 4 if width == 1 else 3000, size=len(seqs))
    ss = O.SeqSet.from_byte_seqs(seqs, data=data, sizeof_d1=width)
    run_both(ctx, ss, k, O.COUNT_FILTER_SET, 2, False, report_all=True, data_width=width)


@pytest.mark.parametrize("k,width,n_labels", [(47, 1, 4), (31, 2, 3000), (51, 4, 70000), (12, 1, 40)])
def test_compact_label_sets_at_the_host_boundary(ctx, k, width, n_labels):
    """dbg_filter_params.compact_sets (round 5): the host table's CSR arrives with set_off as u32 and set_val in the width of D1 --
    the same lists as the plain form and the oracle's (src/filter.rs:85-100), fewer bytes over PCIe."""
    import ctypes as C
    rng = np.random.default_rng(900 + k + width)
    seqs = random_reads(rng, 500, 2500, 150, False)
    data = rng.integers(0, n_labels, size=len(seqs))
    ss = O.SeqSet.from_byte_seqs(seqs, data=data, sizeof_d1=width)
    want = O.filter_kmers(ss, k, O.COUNT_FILTER_SET, 2, stranded=False)
    hs = to_host_seqs(ss, width)
    got, _ = dbg.filter_kmers(hs, dbg.CountFilterSet(2), False, False, 4, k=k, ctx=ctx, compact_sets=3)
    assert_tables_equal(got, want, True)
    got, _ = dbg.filter_kmers(hs, dbg.CountFilter(2), False, False, 4, k=k, ctx=ctx, compact_sets=2)        # flag 2: key_hi omitted for k <= 32
    assert_tables_equal(got, O.filter_kmers(ss, k, O.COUNT_FILTER, 2, stranded=False), False)
    # the raw C table really is narrow
    cs, t = hs.c_struct(), capi.KmerTable()
    p = capi.FilterParams(k, 0, 1, 2, 0, 4, 1)
    ctx.check(ctx.lib.dbg_filter_kmers(ctx.h, C.byref(cs), C.byref(p), C.byref(t)))
    assert (t.set_off_width, t.set_val_width) == (4, width) and t.n == want.n and t.n_set_val == len(want.set_val) and t.key_hi
    off = np.ctypeslib.as_array(C.cast(t.set_off, C.POINTER(C.c_uint32)), shape=(t.n + 1,))
    val = np.ctypeslib.as_array(C.cast(t.set_val, C.POINTER({1: C.c_uint8, 2: C.c_uint16, 4: C.c_uint32}[width])), shape=(max(int(t.n_set_val), 1),))[:int(t.n_set_val)]
    assert np.array_equal(off, want.set_off) and np.array_equal(val.astype(np.uint32), want.set_val)
    ctx.lib.dbg_free_table(ctx.h, C.byref(t))
    # CountFilter ignores the flag; the plain form stays what it was
    p0 = capi.FilterParams(k, 0, 1, 2, 0, 4, 0)
    ctx.check(ctx.lib.dbg_filter_kmers(ctx.h, C.byref(cs), C.byref(p0), C.byref(t)))
    assert (t.set_off_width, t.set_val_width) == (0, 0)
    ctx.lib.dbg_free_table(ctx.h, C.byref(t))


def test_ragged_empty_and_boundary_exts(ctx):
    """Sequences shorter than k contribute nothing (filter.rs:154, lib.rs:813); len == k yields one k-mer
    carrying seq_exts on both sides; non-empty seq_exts land on the first/last k-mer only."""
    rng = np.random.default_rng(5)
    seqs = random_reads(rng, 300, 1500, 120, False, ragged=True)
    seqs += [np.zeros(0, np.uint8), R.random_dna(rng, 47), R.random_dna(rng, 46), R.random_dna(rng, 48)]
    exts = rng.integers(0, 256, size=len(seqs))
    ss = O.SeqSet.from_byte_seqs(seqs, exts=exts)
    for stranded in (False, True):
        run_both(ctx, ss, 47, O.COUNT_FILTER, 1, stranded, report_all=True)


def test_empty_input(ctx):
    ss = O.SeqSet.from_byte_seqs([])
    got, want = run_both(ctx, ss, 31, O.COUNT_FILTER, 1, False)
    assert len(got) == 0
    ss = O.SeqSet.from_byte_seqs([R.random_dna(np.random.default_rng(1), 20)])
    got, want = run_both(ctx, ss, 31, O.COUNT_FILTER, 1, False)
    assert len(got) == 0


def test_palindromes_and_ties(ctx):
    """Even-k palindromic k-mers take the rc branch of min_rc_flip on every observation (lib.rs:224-231)."""
    rng = np.random.default_rng(9)
    half = R.random_dna(rng, 16)
    pal = np.concatenate([half, R.revcomp_bytes(half)])
    seqs = [np.concatenate([R.random_dna(rng, 10), pal, R.random_dna(rng, 10)]) for _ in range(6)]
    ss = O.SeqSet.from_byte_seqs(seqs)
    got, want = run_both(ctx, ss, 32, O.COUNT_FILTER, 1, False)
    assert R.kmer_int(pal) in got.keys()


def test_count_saturates_at_u16(ctx):
    s = R.random_dna(np.random.default_rng(3), 40)
    ss = O.SeqSet.from_byte_seqs([s] * 70000)
    got, want = run_both(ctx, ss, 40, O.COUNT_FILTER, 65535, True)
    assert len(got) == 1 and int(got.count[0]) == 65535
    got, _ = run_both(ctx, ss, 40, O.COUNT_FILTER, 65536, True)
    assert len(got) == 0


def test_degenerate_repeat(ctx):                     # test.rs:170-193 input
    seq = R.from_ascii(R.DEGEN)
    ss = O.SeqSet.from_byte_seqs([seq, seq])
    run_both(ctx, ss, 31, O.COUNT_FILTER, 2, False)


def test_synthetic_stream_parity_and_device_host_generator(ctx):
    """C1-shaped run on the repo's own synthetic stream; also pins device generator == host generator."""
    hs = dbg.synth_reads_host(n_reads=10000, read_len=150, error_rate=0.001, stranded=True, n_colours=0)
    ss = O.SeqSet(hs.words, hs.start, hs.length)
    got, want = run_both(ctx, ss, 31, O.COUNT_FILTER, 1, True)
    assert got.n_kmer_instances == 10000 * 120
    import torch
    import ctypes as C
    p = dbg.synth_params(n_reads=10000, read_len=150, error_rate=0.001, stranded=True, n_colours=0)
    nw = ctx.lib.dbg_synth_words(C.byref(p))
    words = torch.empty(nw, dtype=torch.int64, device="cuda:0")
    start = torch.empty(10000, dtype=torch.int64, device="cuda:0")
    length = torch.empty(10000, dtype=torch.int32, device="cuda:0")
    ctx.check(ctx.lib.dbg_synth_reads_dev(ctx.h, C.byref(p), words.data_ptr(), start.data_ptr(), length.data_ptr(), None))
    assert np.array_equal(words.cpu().numpy().view(np.uint64), hs.words)
    assert np.array_equal(start.cpu().numpy().view(np.uint64), hs.start)


def test_reference_would_panic(ctx):
    ss = O.SeqSet.from_byte_seqs([R.random_dna(np.random.default_rng(1), 100)])
    with pytest.raises(dbg.DbgError):
        dbg.filter_kmers(to_host_seqs(ss), dbg.CountFilter(1), False, False, 0, k=31, ctx=ctx)   # memory_size = 0
    with pytest.raises(dbg.DbgError):
        dbg.filter_kmers(to_host_seqs(ss), dbg.CountFilter(1), False, False, 4, k=65, ctx=ctx)


@pytest.mark.parametrize("k,kind,stranded", [(31, 0, False), (47, 1, False), (15, 0, True), (64, 1, False), (5, 0, False)])
def test_generic_path_in_key_range_passes(ctx, path_mode, k, kind, stranded):
    """The generic path takes as many passes over ranges of the canonical k-mers' top byte as one pass's record limit asks for
    (2^32 - 1 records, or what device memory holds; here forced down to a few thousand) -- the device analogue of the
    reference's bucket ranges (filter.rs:156-168).  Same table, same all_kmers list, whatever the number of passes."""
    if path_mode != "generic":
        pytest.skip("generic path")
    rng = np.random.default_rng(k * 7 + kind)
    seqs = random_reads(rng, 400, 3000, 120, stranded, err=0.01, ragged=True)
    data = rng.integers(0, 300, size=len(seqs)) if kind else None
    ss = O.SeqSet.from_byte_seqs(seqs, data=data, sizeof_d1=2) if kind else O.SeqSet.from_byte_seqs(seqs)
    want = O.filter_kmers(ss, k, kind, 2, stranded=stranded, report_all=True)
    summ = (dbg.CountFilterSet if kind else dbg.CountFilter)(2)
    seen = set()
    for limit in ("1000000000", "9000", "2500"):
        with ctx.options(DBG_GENERIC_PASS_MAX=limit):
            got, allk = dbg.filter_kmers(to_host_seqs(ss, 2 if kind else 0), summ, stranded, True, 4, k=k, ctx=ctx)
        assert_tables_equal(got, want, kind == 1)
        assert allk == [(int(h) << 64) | int(l) for h, l in zip(want.all_hi, want.all_lo)]
        seen.add(got.n_passes)
    assert 1 in seen and max(seen) >= 2
