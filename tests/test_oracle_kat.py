"""Pins the CPU oracle against every known-answer test the reference holds for this path's
building blocks, and against the reference's own randomised test invariants (ported, seeded).

Reference tests restated here (paths under /root/reference/src):
  kmer.rs:10-34 (doctest), dna_string.rs:19-26 (doctest), dna_string.rs:1061-1068 + :1100-1111
  (142-bp k-mer enumeration), :1092-1098 (too short), msp.rs:404-542 (scanner invariants),
  msp.rs:382-402 (k-mer set preservation), test.rs:233-295 (simplify_from_kmers),
  test.rs:299-414 (reassemble_contigs), test.rs:418-504 (reassemble_sharded),
  test.rs:170-193 (degenerate repeat), SURVEY.md Appendix A (hand-derived micro example).
"""
import numpy as np
import pytest

import oracle_lib as O
import refgen as R
from graph_canon import canonical_nodes, graph_kmer_set, node_bases


# ------------------------------------------------------------------ kmer.rs doctest (:10-34)
def test_kmer_doctest():
    k1 = O.kmer_from_ascii("ACGTACGTACGTACGT")
    rc = O.kmer_rc(16, k1)
    assert O.kmer_rc(16, rc) == k1
    assert O.kmer_extend(16, k1, 3, O.LEFT) == O.kmer_from_ascii("TACGTACGTACGTACG")
    s = "TACGTACGTACGTACGTT"
    ks = sorted(O.kmer_from_ascii(s[i:i + 16]) for i in range(3))
    assert ks == [O.kmer_from_ascii(x) for x in
                  ("ACGTACGTACGTACGT", "CGTACGTACGTACGTT", "TACGTACGTACGTACG")]


# ------------------------------------------------------------------ dna_string.rs doctest (:19-26)
def test_dnastring_doctest():
    s = "ACAGCAGCAGCACGTATGACAGATAGTGACAGCAGTTTGTGACCGCAAGAGCAGTAATATGATG"
    words = O.dnastring_pack(R.from_ascii(s))
    km = O.seq_get_kmer(words, 10, 30, 0, 16)          # slice(10, 40).get_kmer::<Kmer16>(0)
    assert km == O.kmer_from_ascii("CACGTATGACAGATAG")
    assert O.kmer_to_ascii(16, km) == "CACGTATGACAGATAG"


# ------------------------------------------------------------------ dna_string.rs:1061-1068, :1100-1111
DNA142 = ("TGCATTAGAAAACTCCTTGCCTGTCAGCCCGACAGGTAGAAACTCATTAATCCACACATTGA"
          "CTCTATTTCAGGTAAATATGACGTCAACTCCTGCATGTTGAAGGCAGTGAGTGGCTGAAACAGCATCAAGGCGTGAAGGC")


@pytest.mark.parametrize("k", [32, 31, 16, 47, 63, 64, 5])
def test_dnastring_kmers(k):
    b = R.from_ascii(DNA142)
    words = O.dnastring_pack(b)
    assert np.array_equal(words, O.dnastring_pack(b, push=True))
    assert np.array_equal(words, O.pack_bases(b))
    for i in range(len(DNA142) - k + 1):
        km = O.seq_get_kmer(words, 0, len(DNA142), i, k)
        assert O.kmer_to_ascii(k, km) == DNA142[i:i + k]
        assert km == R.kmer_int(b[i:i + k])
    # iter_kmers through filter_kmers (stranded, CountFilter(1)) yields exactly these k-mers
    ss = O.SeqSet.from_byte_seqs([b])
    t = O.filter_kmers(ss, k, O.COUNT_FILTER, 1, stranded=True)
    assert sorted(t.keys()) == sorted(set(R.kmers_of(b, k)))


def test_kmers_too_short():                            # dna_string.rs:1092-1098
    ss = O.SeqSet.from_byte_seqs([R.from_ascii("TGCATTAGAA")])
    t = O.filter_kmers(ss, 32, O.COUNT_FILTER, 1, stranded=True)
    assert t.n == 0


# ------------------------------------------------------------------ kmer property tests (kmer.rs:848-930)
@pytest.mark.parametrize("k", [4, 8, 15, 16, 20, 31, 32, 33, 47, 48, 51, 63, 64])
def test_kmer_ops_vs_python(k):
    rng = np.random.default_rng(1000 + k)
    for _ in range(300):
        b = R.random_dna(rng, k)
        v = R.kmer_int(b)
        assert O.kmer_from_ascii(R.to_ascii(b)) == v
        rc = O.kmer_rc(k, v)
        assert rc == R.kmer_rc_int(k, v)
        assert O.kmer_rc(k, rc) == v
        mn, flip = O.kmer_min_rc_flip(k, v)
        assert mn == min(v, rc) and flip == (not v < rc)
        nb = int(rng.integers(0, 4))
        assert O.kmer_extend(k, v, nb, O.RIGHT) == R.kmer_int(list(b[1:]) + [nb])
        assert O.kmer_extend(k, v, nb, O.LEFT) == R.kmer_int([nb] + list(b[:-1]))
    # palindromes: only even k (lib.rs:244-246); ties flip (lib.rs:224-231)
    half = R.random_dna(rng, k // 2)
    if k % 2 == 0:
        pal = R.kmer_int(list(half) + list(R.revcomp_bytes(half)))
        assert O.kmer_is_palindrome(k, pal)
        assert O.kmer_min_rc_flip(k, pal) == (pal, True)


def test_exts_rc():
    for e in range(256):
        assert O.exts_rc(e) == R.exts_rc_py(e)
        assert O.exts_rc(O.exts_rc(e)) == e
    assert O.exts_rc(0x01) == 0x80


# ------------------------------------------------------------------ SURVEY Appendix A micro example
def test_appendix_a_micro():
    ss = O.SeqSet.from_byte_seqs([[0, 1, 2, 3]])
    k = 4                                              # bucket() needs k >= 4: use ACGT itself
    t = O.filter_kmers(ss, k, O.COUNT_FILTER, 1, stranded=True)
    assert t.keys() == [0b00011011] and list(t.exts) == [0] and list(t.count) == [1]
    # k=5 over ACGTA: ACGTA (exts: right none) ; stranded keys
    ss = O.SeqSet.from_byte_seqs([[0, 1, 2, 3, 0, 1]])
    t = O.filter_kmers(ss, 5, O.COUNT_FILTER, 1, stranded=True)
    a, c = R.kmer_int([0, 1, 2, 3, 0]), R.kmer_int([1, 2, 3, 0, 1])
    assert t.keys() == [a, c]
    assert list(t.exts) == [1 << (4 + 1), 1 << 0]      # ACGTA -> right C ; CGTAC <- left A


# ------------------------------------------------------------------ filter_kmers vs dictionary model
@pytest.mark.parametrize("k,stranded", [(31, True), (31, False), (47, False), (63, False), (64, False), (32, False),
                                        (8, False), (16, True)])
def test_filter_vs_naive(k, stranded):
    rng = np.random.default_rng(7 * k + stranded)
    genome = R.random_dna(rng, 600)
    seqs, exts = [], []
    for _ in range(120):
        ln = int(rng.integers(0, 160))
        st = int(rng.integers(0, len(genome) - ln + 1))
        s = genome[st:st + ln].copy()
        if not stranded and rng.random() < 0.5:
            s = R.revcomp_bytes(s)
        seqs.append(s)
        exts.append(int(rng.integers(0, 256)))
    ss = O.SeqSet.from_byte_seqs(seqs, exts=exts)
    model = R.naive_filter(seqs, exts, k, stranded)
    for min_obs in (1, 2, 3):
        t = O.filter_kmers(ss, k, O.COUNT_FILTER, min_obs, stranded=stranded, report_all=True)
        want = sorted(v for v, (c, _) in model.items() if c >= min_obs)
        assert t.keys() == want
        assert [int(x) for x in t.count] == [model[v][0] for v in want]
        assert [int(x) for x in t.exts] == [model[v][1] for v in want]
        allk = [(int(h) << 64) | int(l) for h, l in zip(t.all_hi, t.all_lo)]
        assert allk == sorted(model.keys())


def test_filter_multipass_equals_single():
    """memory_size only changes the pass count (filter.rs:156-168), never the result."""
    rng = np.random.default_rng(5)
    seqs = [R.random_dna(rng, 150) for _ in range(40)]
    ss = O.SeqSet.from_byte_seqs(seqs, data=np.arange(40) % 4, sizeof_d1=1)
    a = O.filter_kmers(ss, 31, O.COUNT_FILTER, 1, stranded=False, memory_size=4)
    assert a.passes == 1
    with pytest.raises(RuntimeError):
        O.filter_kmers(ss, 31, O.COUNT_FILTER, 1, stranded=False, memory_size=0)


@pytest.mark.parametrize("stranded", [False, True])
@pytest.mark.parametrize("summ", ["count", "set"])
def test_threaded_sharded_baseline_counts_like_unsharded(stranded, summ):
    """bench.py's many-core CPU leg (msp_sequence -> shards -> filter_kmers per shard on threads, the pipeline of
    test.rs:418-504) keeps exactly the k-mers the unsharded filter_kmers keeps."""
    rng = np.random.default_rng(11)
    genome = R.random_dna(rng, 3000)
    seqs = []
    for _ in range(400):
        a = int(rng.integers(0, 3000 - 150))
        seqs.append(genome[a:a + 150].copy())
    ss = O.SeqSet.from_byte_seqs(seqs, data=np.arange(400) % 4, sizeof_d1=1)
    which = O.COUNT_FILTER if summ == "count" else O.COUNT_FILTER_SET
    want = len(O.filter_kmers(ss, 47, which, 2, stranded=stranded).key_lo)
    for nt, ns in ((1, 1), (3, 7), (4, 64)):
        _, got = O.time_filter_kmers_sharded_mt(ss, 47, which, 2, stranded, nt, ns)
        assert got == want


def test_count_filter_saturates():
    s = R.random_dna(np.random.default_rng(3), 40)
    n = 70000
    ss = O.SeqSet.from_byte_seqs([s] * n)
    t = O.filter_kmers(ss, 40, O.COUNT_FILTER, 65536, stranded=True)      # filter.rs:55-61
    assert t.n == 0
    t = O.filter_kmers(ss, 40, O.COUNT_FILTER, 65535, stranded=True)
    assert t.n == 1 and int(t.count[0]) == 65535


def test_count_filter_set():
    rng = np.random.default_rng(11)
    genome = R.random_dna(rng, 300)
    seqs, data = [], []
    for i in range(60):
        st = int(rng.integers(0, 200))
        seqs.append(genome[st:st + 100])
        data.append(i % 5)
    ss = O.SeqSet.from_byte_seqs(seqs, data=data, sizeof_d1=1)
    k = 47
    t = O.filter_kmers(ss, k, O.COUNT_FILTER_SET, 2, stranded=False)
    model = {}
    for s, d in zip(seqs, data):
        for v in R.kmers_of(s, k):
            model.setdefault(R.canon(k, v), []).append(d)
    want = sorted(v for v, ds in model.items() if len(ds) >= 2)
    assert t.keys() == want
    for i, v in enumerate(want):
        got = [int(x) for x in t.set_val[int(t.set_off[i]):int(t.set_off[i + 1])]]
        assert got == sorted(set(model[v]))


# ------------------------------------------------------------------ MSP scanner invariants (msp.rs:404-486)
def check_msp_slices(k, p, seq, minimizer, start, length, mpos, score):
    m = len(seq)
    covered = np.zeros(m - k + 1, dtype=bool)
    for s, l in zip(start, length):
        s, l = int(s), int(l)
        end = s + l - k + 1
        assert not covered[s:end].any(), "base already covered"
        covered[s:end] = True
    assert covered.all(), "a pmer wasn't covered"
    for l in length:
        assert p <= int(l) <= 2 * k - p
    pm = R.kmers_of(seq, p)
    sc = [score(v) for v in pm]
    for mi, s, l, mp in zip(minimizer, start, length, mpos):
        s, l, mp = int(s), int(l), int(mp)
        assert int(mi) == pm[mp]
        assert all(sc[i] >= score(int(mi)) for i in range(s, s + l - p + 1)), "found better pmer within slice"
    for mi, s, l, mp in list(zip(minimizer, start, length, mpos))[:-1]:
        s, l, mp = int(s), int(l), int(mp)
        next_kmer_pos = s + l - k + 1
        covers = next_kmer_pos <= mp
        next_score = sc[s + l - p + 1]
        assert next_score < score(int(mi)) or not covers, "slice ended before it should have"


@pytest.mark.parametrize("p", [5, 8, 10, 12, 14, 15, 16])
def test_msp_scanner_invariants(p):
    rng = np.random.default_rng(p)
    for k in (16, 21, 31, 35, 47, 63):
        if p >= k:
            continue
        lens = [k, k + 1, 2 * k, 150, 300, 20 * k]
        for m in lens:
            seq = R.random_dna(rng, m)
            mi, st, ln, mp = O.msp_scan(seq, k, p, None, rc=False)
            check_msp_slices(k, p, seq, mi, st, ln, mp, lambda v: v)
            mi, st, ln, mp = O.msp_scan(seq, k, p, None, rc=False, score_mode=1)
            atc = lambda v: sum(1 for j in range(p) if ((v >> (2 * j)) & 3) in (0, 3))
            check_msp_slices(k, p, seq, mi, st, ln, mp, atc)
        for m in (k, k + 3, 3 * k):                      # DnaString::blank (all A), msp.rs:520-531
            seq = np.zeros(m, dtype=np.uint8)
            mi, st, ln, mp = O.msp_scan(seq, k, p, None, rc=False)
            check_msp_slices(k, p, seq, mi, st, ln, mp, lambda v: v)


def test_msp_poly_a_tie_rule():
    """msp.rs:127-141,218-228,241-247: rescans pick the rightmost minimal p-mer; a tie on entry
    does not replace.  All-A: first interval's minimizer is at k-p, next at 2(k-p)+1, ..."""
    k, p, m = 21, 5, 100
    mi, st, ln, mp = O.msp_scan(np.zeros(m, np.uint8), k, p, None, rc=True)
    assert list(mp[:3]) == [k - p, 2 * (k - p) + 1, 3 * (k - p) + 2]
    assert list(st[:3]) == [0, k - p + 1, 2 * (k - p) + 2]


def test_msp_kmer_set_preserved():                      # msp.rs:382-402 (k=50, p=8, rc=true)
    rng = np.random.default_rng(77)
    for _ in range(100):
        dna = R.random_dna(rng, 150)
        bu, ex, st, ln = O.msp_sequence(dna, 50, 8, None, rc=True)
        want = set(R.kmers_of(dna, 50))
        got = set()
        for s, l in zip(st, ln):
            got.update(R.kmers_of(dna[int(s):int(s) + int(l)], 50))
        assert want == got
        for b, s, l, e in zip(bu, st, ln, ex):
            s, l = int(s), int(l)
            le = (1 << int(dna[s - 1])) if s > 0 else 0
            re = (1 << int(dna[s + l])) if s + l < len(dna) else 0
            assert int(e) == (re << 4) | le              # lib.rs:645-660
            pm = [R.canon(8, v) for v in R.kmers_of(dna[s:s + l], 8)]
            assert int(b) == min(pm)                     # bucket = canonical minimizer (msp.rs:115-117)


def test_msp_sample_vectors_run():                      # msp.rs:551-581 inputs (k=35, p=5): invariants only
    from golden_inputs import MSP_V1, MSP_V2
    for v in (MSP_V1, MSP_V2):
        seq = np.array(v, dtype=np.uint8)
        mi, st, ln, mp = O.msp_scan(seq, 35, 5, None, rc=True)
        check_msp_slices(35, 5, seq, mi, st, ln, mp, lambda x: R.canon(5, x))


def test_msp_lmer_layout():                             # vmer.rs:32-47,125-134
    rng = np.random.default_rng(2)
    dna = R.random_dna(rng, 150)
    bu, ex, st, ln, lw = O.msp_sequence(dna, 47, 8, None, rc=True, vmer_max_len=92, lmer_words=3)
    for i in range(len(bu)):
        s, l = int(st[i]), int(ln[i])
        assert int(lw[i][2]) & 0xFF == l
        w = lw[i].copy()
        w[2] &= np.uint64(~0xFF & O.M64)
        assert list(O.unpack_bases(w, 0, l)) == list(dna[s:s + l])
    with pytest.raises(RuntimeError):                   # assert V::max_len() >= 2k-p (msp.rs:292)
        O.msp_sequence(dna, 63, 8, None, rc=True, vmer_max_len=92)


# ------------------------------------------------------------------ reassemble_contigs (test.rs:299-414)
def reassemble_contigs(contigs, k, stranded=False, seed_perm=None):
    p = 6
    kmer_set = set()
    for c in contigs:
        kmer_set.update(R.canon(k, v) for v in R.kmers_of(c, k))
    pieces, pexts, pdata = [], [], []
    perm = np.arange(1 << (2 * p), dtype=np.uint64)
    for c in contigs:
        if len(c) < k:
            continue
        bu, ex, st, ln = O.msp_sequence(c, k, p, perm, rc=True)
        for d in (0, 1):                                 # each piece is fed twice (test.rs:323-324)
            for e, s, l in zip(ex, st, ln):
                pieces.append(c[int(s):int(s) + int(l)])
                pexts.append(int(e))
                pdata.append(d)
    msp_kmers = set()
    for s in pieces:
        msp_kmers.update(R.canon(k, v) for v in R.kmers_of(s, k))
    assert kmer_set == msp_kmers                         # test.rs:341
    ss = O.SeqSet.from_byte_seqs(pieces, exts=pexts, data=pdata, sizeof_d1=1)
    t = O.filter_kmers(ss, k, O.COUNT_FILTER, 2, stranded=stranded)
    assert set(t.keys()) == kmer_set                     # test.rs:355
    ext_set = set()
    for v, e in zip(t.keys(), t.exts):
        e = int(e)
        for b in range(4):
            if e & (1 << b):
                ext_set.add(R.canon(k, O.kmer_extend(k, v, b, O.LEFT)))
            if e & (1 << (4 + b)):
                ext_set.add(R.canon(k, O.kmer_extend(k, v, b, O.RIGHT)))
    if len(kmer_set) > 1:
        assert kmer_set >= ext_set                       # no invented k-mers
    g = O.compress_kmers(k, stranded, O.SPEC_SAT_ADD, t.key_hi, t.key_lo, t.exts, t.count, seed_perm)
    ga = g.arrays()
    all_contig = set()
    for i in range(len(ga["start"])):
        b = node_bases(ga, i)
        cs = set(R.canon(k, v) for v in R.kmers_of(b, k))
        assert kmer_set >= cs
        all_contig |= cs
        ks = R.kmers_of(b, k)
        e = int(ga["exts"][i])
        for x in range(4):
            if e & (1 << x):
                assert R.canon(k, O.kmer_extend(k, ks[0], x, O.LEFT)) in kmer_set
            if e & (1 << (4 + x)):
                assert R.canon(k, O.kmer_extend(k, ks[-1], x, O.RIGHT)) in kmer_set
    assert all_contig == kmer_set                        # test.rs:413
    assert sum(int(x) - k + 1 for x in ga["length"]) == len(kmer_set)   # each k-mer in exactly one node
    return t, g


def test_simple_kmer_compress():                        # test.rs:156-160 (IntKmer<u64>, k=32)
    reassemble_contigs(R.simple_random_contigs(np.random.default_rng(1)), 32)


def test_degen_seq_asm():                               # test.rs:168-180 (K31)
    seq = R.from_ascii(R.DEGEN)
    reassemble_contigs([seq, seq], 31)


@pytest.mark.parametrize("seed", range(6))
def test_complex_kmer_compress(seed):                   # test.rs:196-201
    reassemble_contigs(R.random_contigs(np.random.default_rng(100 + seed)), 32)


@pytest.mark.parametrize("k", [47, 63])
def test_complex_kmer_compress_wide(k):
    reassemble_contigs(R.random_contigs(np.random.default_rng(k)), k)


# ------------------------------------------------------------------ simplify_from_kmers (test.rs:233-295)
def simplify_from_kmers(contigs, k, stranded=False):
    ss = O.SeqSet.from_byte_seqs(contigs)
    t = O.filter_kmers(ss, k, O.COUNT_FILTER, 1, stranded=stranded)
    g = O.compress_kmers(k, stranded, O.SPEC_ADD_MOD, t.key_hi, t.key_lo, t.exts, t.count)
    assert g.is_compressed(O.SPEC_ADD_MOD) is None       # test.rs:248-254
    # one node per k-mer (test.rs:257-262), then compress_graph (test.rs:265-274)
    n = t.n
    words = np.zeros(0, np.uint64)
    seqs = []
    for v in t.keys():
        seqs.append([(v >> (2 * (k - 1 - j))) & 3 for j in range(k)])
    s2 = O.SeqSet.from_byte_seqs(seqs)
    ug = O.graph_from_arrays(k, stranded, s2.words, s2.start, s2.length, t.exts, np.ones(n, np.uint32))
    simp = ug.finish().compress_graph(stranded, O.SPEC_WRAP_ADD)
    assert simp.is_compressed(O.SPEC_WRAP_ADD) is None
    sa = simp.arrays()
    # every valid k-mer appears in exactly one node (MPHF bijection check, test.rs:281-294)
    seen = []
    for i in range(len(sa["start"])):
        seen.extend(v if stranded else R.canon(k, v) for v in R.kmers_of(node_bases(sa, i), k))
    assert sorted(seen) == t.keys()
    # both routes give the same unitigs up to orientation / order / cycle cut
    assert ([x[:2] for x in canonical_nodes(g.arrays(), k, stranded)] ==
            [x[:2] for x in canonical_nodes(sa, k, stranded)])
    return g


def test_simple_path_compress():
    simplify_from_kmers(R.simple_random_contigs(np.random.default_rng(9)), 32)


@pytest.mark.parametrize("seed", range(8))
def test_complex_path_compress_k31(seed):               # test.rs:218-223
    simplify_from_kmers(R.random_contigs(np.random.default_rng(200 + seed)), 31)


@pytest.mark.parametrize("seed", range(3))
def test_complex_path_compress_stranded(seed):          # stranded=true is unpinned by the reference (F7)
    simplify_from_kmers(R.random_contigs(np.random.default_rng(300 + seed)), 31, stranded=True)


# ------------------------------------------------------------------ reassemble_sharded (test.rs:418-504)
def reassemble_sharded(contigs, k, stranded=False):
    kmer_set = set()
    for c in contigs:
        kmer_set.update(R.canon(k, v) for v in R.kmers_of(c, k))
    shards = {}
    for c in contigs:
        if len(c) < k:
            continue
        bu, ex, st, ln = O.msp_sequence(c, k, 6, None, rc=True)
        for b, e, s, l in zip(bu, ex, st, ln):
            v = shards.setdefault(int(b), ([], [], []))
            for d in (0, 1):
                v[0].append(c[int(s):int(s) + int(l)])
                v[1].append(int(e))
                v[2].append(d)
    graphs = []
    for b in sorted(shards):
        sq, ex, da = shards[b]
        ss = O.SeqSet.from_byte_seqs(sq, exts=ex, data=da, sizeof_d1=1)
        t = O.filter_kmers(ss, k, O.COUNT_FILTER, 2, stranded=stranded)
        graphs.append(O.compress_kmers(k, stranded, O.SPEC_SAT_ADD, t.key_hi, t.key_lo, t.exts, t.count))
    comb = O.graph_combine(graphs).finish()
    dbg = comb.compress_graph(False, O.SPEC_MAX)         # test.rs:470
    ga = dbg.arrays()
    assert graph_kmer_set(ga, k, False) == kmer_set      # test.rs:503
    for i in range(len(ga["start"])):
        ks = R.kmers_of(node_bases(ga, i), k)
        e = int(ga["exts"][i])
        for x in range(4):
            if e & (1 << x):
                assert R.canon(k, O.kmer_extend(k, ks[0], x, O.LEFT)) in kmer_set
            if e & (1 << (4 + x)):
                assert R.canon(k, O.kmer_extend(k, ks[-1], x, O.RIGHT)) in kmer_set
    return dbg


def test_simple_sharded():
    reassemble_sharded(R.simple_random_contigs(np.random.default_rng(21)), 32)


def test_degen_seq_asm_sharded():
    seq = R.from_ascii(R.DEGEN)
    reassemble_sharded([seq, seq], 31)


@pytest.mark.parametrize("seed", range(4))
def test_complex_sharded(seed):
    reassemble_sharded(R.random_contigs(np.random.default_rng(400 + seed)), 32)


def test_sharded_equals_unsharded_unitigs():
    """The sharded flow and the direct flow agree on the unitig set (up to rc / order / cycle cut)."""
    contigs = R.random_contigs(np.random.default_rng(55))
    k = 31
    dbg = reassemble_sharded(contigs, k)
    pieces = []
    for c in contigs:
        pieces.extend([c, c])
    t = O.filter_kmers(O.SeqSet.from_byte_seqs(pieces), k, O.COUNT_FILTER, 2, stranded=False)
    g = O.compress_kmers(k, False, O.SPEC_MAX, t.key_hi, t.key_lo, t.exts, t.count)
    a = [x[:2] for x in canonical_nodes(dbg.arrays(), k, False)]
    b = [x[:2] for x in canonical_nodes(g.arrays(), k, False)]
    assert a == b


# ------------------------------------------------------------------ seed-order independence
@pytest.mark.parametrize("seed", range(5))
def test_seed_order_invariance(seed):
    """The canonicalised unitig set must not depend on the (unpinned) MPHF slot order."""
    rng = np.random.default_rng(500 + seed)
    contigs = R.random_contigs(rng) if seed else R.simple_random_contigs(rng)
    k = 31 if seed % 2 else 32
    ss = O.SeqSet.from_byte_seqs(contigs)
    t = O.filter_kmers(ss, k, O.COUNT_FILTER, 1, stranded=False)
    base = canonical_nodes(O.compress_kmers(k, False, O.SPEC_SAT_ADD, t.key_hi, t.key_lo, t.exts, t.count).arrays(),
                           k, False)
    for _ in range(4):
        perm = rng.permutation(t.n).astype(np.uint64)
        g = O.compress_kmers(k, False, O.SPEC_SAT_ADD, t.key_hi, t.key_lo, t.exts, t.count, perm)
        assert canonical_nodes(g.arrays(), k, False) == base


def test_scmap_compress_breaks_on_colour():
    """ScmapCompress: join only equal data (compression.rs:84-98)."""
    rng = np.random.default_rng(8)
    a = R.random_dna(rng, 80)
    k = 31
    ss = O.SeqSet.from_byte_seqs([a])
    t = O.filter_kmers(ss, k, O.COUNT_FILTER, 1, stranded=True)
    col = np.zeros(t.n, np.uint32)
    order = {v: i for i, v in enumerate(t.keys())}
    path = R.kmers_of(a, k)
    for j, v in enumerate(path):
        col[order[v]] = 0 if j < 20 else 1
    g = O.compress_kmers(k, True, O.SPEC_SCMAP_EQ, t.key_hi, t.key_lo, t.exts, col)
    ga = g.arrays()
    assert sorted(int(x) for x in ga["length"]) == sorted([20 + k - 1, len(path) - 20 + k - 1])
    assert g.is_compressed(O.SPEC_SCMAP_EQ) is None


# ------------------------------------------------------------------ compress_kmers_no_exts (compression.rs:619-659)
@pytest.mark.parametrize("k,stranded", [(31, False), (32, True), (47, False), (63, False)])
def test_compress_kmers_no_exts_against_a_set_model(k, stranded):
    """The Exts the restatement derives are those of an independent Python set model (one bit per neighbour whose min_rc is in
    the set, whatever `stranded` says: compression.rs:626), the result equals compress_kmers on those Exts (:657-658), and on
    the complete k-mer set of error-free contigs the derived Exts contain the Exts the reads themselves showed."""
    rng = np.random.default_rng(k)
    contigs = R.random_contigs(rng)
    t = O.filter_kmers(O.SeqSet.from_byte_seqs(contigs), k, O.COUNT_FILTER, 1, stranded=False)
    keys = t.keys()
    kset = set(keys)
    perm = rng.permutation(len(keys))                                   # any key order (a slice of (K, D) pairs)
    hi, lo, cnt = t.key_hi[perm], t.key_lo[perm], t.count[perm]
    g, ex = O.compress_kmers_no_exts(k, stranded, O.SPEC_SAT_ADD, hi, lo, cnt)
    for i, p in enumerate(perm):
        v, e = keys[p], 0
        for b in range(4):
            if R.canon(k, O.kmer_extend(k, v, b, O.LEFT)) in kset:
                e |= 1 << b
            if R.canon(k, O.kmer_extend(k, v, b, O.RIGHT)) in kset:
                e |= 16 << b
        assert e == int(ex[i])
        assert int(t.exts[p]) & ~e == 0                                  # what the reads showed is a subset
    want = O.compress_kmers(k, stranded, O.SPEC_SAT_ADD, hi, lo, ex, cnt)
    from graph_canon import graphs_equal
    assert graphs_equal(g.arrays(), want.arrays())
    with pytest.raises(RuntimeError):                                    # assert_eq!(kmer_set.len(), keys.len())
        O.compress_kmers_no_exts(k, stranded, O.SPEC_SAT_ADD, np.concatenate([hi, hi[:1]]), np.concatenate([lo, lo[:1]]), None)
