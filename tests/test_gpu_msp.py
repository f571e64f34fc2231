"""GPU parity: dbg_msp_sequence (HIP scanner, through the C ABI) vs the CPU oracle's restatement of
Scanner::scan / msp_sequence (src/msp.rs:207-324), bit-exact including the tie rule."""
import numpy as np
import pytest

import oracle_lib as O
import refgen as R
from pkg import dbg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = dbg.Context(0)
    yield c
    c.close()


def check_batch(ctx, seqs, k, p, perm=None, rc=True, lmer_words=0):
    res = dbg.msp_sequence_batch(k, seqs, perm, rc, p, lmer_words, ctx)
    off = res["piece_off"]
    assert len(off) == len(seqs) + 1 and off[0] == 0
    perm64 = None if perm is None else np.asarray(perm, dtype=np.uint64)
    for i, s in enumerate(seqs):
        a, b = int(off[i]), int(off[i + 1])
        if len(s) < k:
            assert a == b                                   # msp.rs:294-296
            continue
        mi, st, ln, mp = O.msp_scan(s, k, p, perm64, rc=rc)
        want = O.msp_sequence(s, k, p, perm64, rc=rc, lmer_words=lmer_words)
        assert b - a == len(st), (i, b - a, len(st))
        assert np.array_equal(res["start"][a:b], st)
        assert np.array_equal(res["len"][a:b], ln)
        assert np.array_equal(res["minimizer_pos"][a:b], mp)
        assert np.array_equal(res["bucket"][a:b], want[0])
        assert np.array_equal(res["exts"][a:b], want[1])
        if lmer_words:
            assert np.array_equal(res["lmer"][a:b], want[4])
    return res


@pytest.mark.parametrize("k,p", [(47, 8), (31, 6), (35, 5), (63, 8), (50, 8), (16, 5), (21, 10), (47, 12), (64, 16),
                                 (33, 15), (47, 14)])
def test_msp_random_reads(ctx, k, p):
    rng = np.random.default_rng(k * 100 + p)
    seqs = [R.random_dna(rng, int(n)) for n in rng.integers(0, 400, size=150)]
    seqs += [R.random_dna(rng, k), R.random_dna(rng, k + 1), R.random_dna(rng, k - 1), np.zeros(0, np.uint8)]
    check_batch(ctx, seqs, k, p, None, True)
    check_batch(ctx, seqs, k, p, None, False)


def test_msp_low_complexity_ties(ctx):
    """All-A, dinucleotide and short-period repeats exercise the rightmost-min / strict-entry tie rule."""
    rng = np.random.default_rng(4)
    seqs = [np.zeros(n, np.uint8) for n in (47, 48, 100, 150, 300)]
    seqs += [np.tile(np.array([0, 3], np.uint8), 100), np.tile(np.array([0, 1, 2], np.uint8), 80),
             np.tile(R.random_dna(rng, 7), 40), R.from_ascii(R.DEGEN)]
    for k, p in ((31, 6), (47, 8), (21, 5)):
        check_batch(ctx, seqs, k, p, None, True)
        check_batch(ctx, seqs, k, p, None, False)


def test_msp_permutation(ctx):
    rng = np.random.default_rng(8)
    p = 6
    perm = rng.permutation(1 << (2 * p)).astype(np.uint32)
    seqs = [R.random_dna(rng, 150) for _ in range(200)]
    check_batch(ctx, seqs, 31, p, perm, True)
    check_batch(ctx, seqs, 31, p, perm, False)


def test_msp_lmer_words(ctx):
    rng = np.random.default_rng(12)
    seqs = [R.random_dna(rng, 150) for _ in range(100)]
    check_batch(ctx, seqs, 47, 8, None, True, lmer_words=3)          # Lmer3: max_len 92 >= 2*47-8 = 86
    check_batch(ctx, seqs, 63, 8, None, True, lmer_words=4)          # Lmer4: max_len 124 >= 118
    with pytest.raises(dbg.DbgError):                                # msp.rs:292 assertion
        dbg.msp_sequence_batch(63, seqs, None, True, 8, 3, ctx)


def test_msp_reference_sample_vectors(ctx):                         # msp.rs:551-581 inputs
    from golden_inputs import MSP_V1, MSP_V2
    seqs = [np.array(MSP_V1, np.uint8), np.array(MSP_V2, np.uint8)]
    check_batch(ctx, seqs, 35, 5, None, True)


def test_msp_single_sequence_api(ctx):
    """msp_sequence::<Kmer8, _>(k, seq, None, true) for one read, like src/test.rs:436."""
    rng = np.random.default_rng(2)
    dna = R.random_dna(rng, 150)
    pieces = dbg.msp_sequence(50, dna, None, True, p=8, ctx=ctx)
    want = set(R.kmers_of(dna, 50))
    got = set()
    for bucket, exts, v in pieces:
        got.update(R.kmers_of(v, 50))
        assert bucket == min(R.canon(8, x) for x in R.kmers_of(v, 8))
    assert want == got                                               # msp.rs:340-355


def test_msp_then_filter_equals_direct_filter(ctx):
    """Per-k-mer results are independent of the MSP sharding: feeding the pieces (with their boundary
    Exts) to filter_kmers gives exactly the table of the whole reads (src/test.rs:326-355)."""
    rng = np.random.default_rng(31)
    genome = R.random_dna(rng, 2000)
    reads = []
    for _ in range(300):
        st = int(rng.integers(0, 1850))
        s = genome[st:st + 150].copy()
        reads.append(R.revcomp_bytes(s) if rng.random() < 0.5 else s)
    k, p = 47, 8
    res = dbg.msp_sequence_batch(k, reads, None, True, p, 0, ctx)
    pieces, pexts = [], []
    for i, r in enumerate(reads):
        for j in range(int(res["piece_off"][i]), int(res["piece_off"][i + 1])):
            s, l = int(res["start"][j]), int(res["len"][j])
            pieces.append(r[s:s + l])
            pexts.append(int(res["exts"][j]))
    a, _ = dbg.filter_kmers([(s, e, None) for s, e in zip(pieces, pexts)], dbg.CountFilter(2), False, False, 4, k=k, ctx=ctx)
    b, _ = dbg.filter_kmers([(s, 0, None) for s in reads], dbg.CountFilter(2), False, False, 4, k=k, ctx=ctx)
    assert a.keys() == b.keys()
    assert np.array_equal(a.exts, b.exts) and np.array_equal(a.count, b.count)


@pytest.mark.parametrize("k,p,alphabet", [(21, 3, 2), (31, 4, 2), (25, 5, 3), (47, 6, 2), (40, 3, 4), (64, 8, 2)])
def test_msp_tie_heavy_short_reads(ctx, k, p, alphabet):
    """Small alphabets and tiny p make equal p-mers inside one window the normal case: every step of the wave-per-read
    orbit (first strictly smaller entry vs rightmost minimum after expiry) is decided by the tie rule.  Lengths straddle the
    256-position limit between the wave-per-read kernel and the literal lane-per-read scanner."""
    rng = np.random.default_rng(1000 * k + p)
    seqs = [rng.integers(0, alphabet, size=int(n)).astype(np.uint8) for n in rng.integers(k, 300, size=300)]
    seqs += [rng.integers(0, alphabet, size=n).astype(np.uint8) for n in (255 + p, 256 + p - 1, 256 + p, 257 + p)]
    perm = rng.permutation(4 ** p).astype(np.uint32)
    check_batch(ctx, seqs, k, p, None, True)
    check_batch(ctx, seqs, k, p, None, False)
    check_batch(ctx, seqs, k, p, perm, True, lmer_words=0 if k > 47 else 3)


@pytest.mark.parametrize("k,p", [(47, 8), (31, 6), (21, 11), (106, 11)])
def test_msp_kernel_routes(ctx, k, p):
    """Three kernels share the reads by length and shape (lane-per-read up to 1024 bases when p <= 11; wave-per-read up to 256
    p-mer positions; the literal loop beyond): one batch holds reads on every side of those borders, tie-heavy ones included,
    and the wave route forced for everything (DBG_MSP=wave) must give the same pieces."""
    rng = np.random.default_rng(k + p)
    lens = [k, k + 1, 150, 255 + p, 256 + p, 257 + p, 1023, 1024, 1025, 1500, 3000, 40, 0]
    seqs = [R.random_dna(rng, n) for n in lens if n == 0 or n >= 1]
    seqs += [np.zeros(1024, np.uint8), np.tile(np.array([0, 3], np.uint8), 512), np.tile(R.random_dna(rng, 5), 205)[:1024],
             rng.integers(0, 2, size=1024).astype(np.uint8), rng.integers(0, 2, size=1030).astype(np.uint8)]
    for rc in (True, False):
        a = check_batch(ctx, seqs, k, p, None, rc, lmer_words=4 if 2 * k - p <= 124 else 0)
        with ctx.options(DBG_MSP="wave"):
            b = check_batch(ctx, seqs, k, p, None, rc, lmer_words=4 if 2 * k - p <= 124 else 0)
        for name in ("piece_off", "bucket", "exts", "start", "len", "minimizer_pos"):
            assert np.array_equal(a[name], b[name]), name
        with ctx.options(DBG_MSP="twopass"):                 # the lane kernel without its piece stash: count pass + emit pass
            b = check_batch(ctx, seqs, k, p, None, rc, lmer_words=4 if 2 * k - p <= 124 else 0)
        for name in ("piece_off", "bucket", "exts", "start", "len", "minimizer_pos"):
            assert np.array_equal(a[name], b[name]), name


@pytest.mark.parametrize("k,p", [(47, 8), (31, 6), (21, 4)])
def test_msp_stash_boundary(ctx, k, p):
    """The lane kernel keeps a read's first 16 pieces in a stash and re-scans only the reads with more: reads with 1 .. ~60
    pieces in one batch (the 17th piece is the first to come from the second scan), and a batch in which no read needs it."""
    rng = np.random.default_rng(k * 100 + p)
    lens = list(rng.integers(k, 700, size=400)) + [k, k + 1]
    seqs = [R.random_dna(rng, int(n)) for n in lens]
    res = check_batch(ctx, seqs, k, p, None, True, lmer_words=3 if 2 * k - p <= 92 else 0)
    per_read = np.diff(res["piece_off"])
    assert per_read.max() > 17 and (per_read == 16).any() or per_read.max() > 17       # both sides of the stash size occur
    with ctx.options(DBG_MSP="twopass"):
        two = check_batch(ctx, seqs, k, p, None, True, lmer_words=3 if 2 * k - p <= 92 else 0)
    for name in ("piece_off", "bucket", "exts", "start", "len", "minimizer_pos"):
        assert np.array_equal(res[name], two[name]), name
    short = [R.random_dna(rng, int(n)) for n in rng.integers(k, k + 40, size=200)]     # at most a few pieces each: no second scan
    assert np.diff(check_batch(ctx, short, k, p, None, False)["piece_off"]).max() <= 16


def test_msp_permutation_values_beyond_packing(ctx):
    """The lane kernel packs (score, position) into 32 bits; a `permutation` table with scores of 2^22 and more (not a
    permutation of 0..4^p, which the reference does not demand) must take the other kernels and still be exact."""
    rng = np.random.default_rng(11)
    p, k = 6, 31
    perm = (rng.permutation(4 ** p).astype(np.uint64) * np.uint64(1 << 20)).astype(np.uint32)     # scores up to 2^32
    seqs = [R.random_dna(rng, int(n)) for n in rng.integers(20, 600, size=60)]
    seqs.append(rng.integers(0, 2, size=500).astype(np.uint8))
    check_batch(ctx, seqs, k, p, perm, True)
    small = rng.permutation(4 ** p).astype(np.uint32)
    check_batch(ctx, seqs, k, p, small, True)
    check_batch(ctx, seqs, k, p, small, False)
