"""CPU: oracle restatement of the data formats either side of the path, pinned to the reference's known answers:
from_acgt_bytes (dna_string.rs:222-250; tests :952-983 and bitops_avx2.rs:138-216) and write_gfa (graph.rs:537-611)."""
import numpy as np

import oracle_lib as O
import refgen as R

ACGT = b"ACGT"


def test_pack_known_words():
    # bitops_avx2.rs:141-149: the eight fixed 32-mers; the packed word is the Kmer32 with the first base in the top bits
    cases = {
        b"C" * 32: 0x5555555555555555, b"G" * 32: 0xAAAAAAAAAAAAAAAA, b"T" * 32: 0xFFFFFFFFFFFFFFFF,
        b"AT" * 16: 0x3333333333333333, b"A" * 16 + b"T" * 16: 0x00000000FFFFFFFF,
        b"ACGT" * 8: 0x1B1B1B1B1B1B1B1B, b"AAAAAAAACCCCCCCCGGGGGGGGTTTTTTTT": 0x00005555AAAAFFFF,
        b"AAAACCCCGGGGTTTT" * 2: 0x0055AAFF0055AAFF,
    }
    for s, want in cases.items():
        assert int(O.pack_acgt(s)[0]) == want
        assert O.unpack_acgt(O.pack_acgt(s), 0, 32) == s


def test_pack_matches_from_dna_string():
    # dna_string.rs:952-983: from_acgt_bytes == from_dna_string, len, to_string round trip
    for dna in [b"", b"A", b"C", b"G", b"T", b"GC", b"ATA", b"ACGTACGT", b"ACGTAAAAAAAAAATTATATAACGT", b"AACGTAAAAAAAAAATTATATAACGT"]:
        w = O.pack_acgt(dna)
        want = O.dnastring_pack([ACGT.index(c) for c in dna]) if len(dna) else np.zeros(0, np.uint64)
        assert np.array_equal(w, want)
        assert O.unpack_acgt(w, 0, len(dna)) == dna


def test_pack_invalid_and_lowercase():
    # base_to_bits (lib.rs:65-73): case-insensitive, anything else -> A; bitops_avx2.rs:193-216 checks every byte value
    rng = np.random.default_rng(1)
    for _ in range(50):
        n = int(rng.integers(1, 200))
        a = rng.integers(0, 256, n).astype(np.uint8)
        bits = [{65: 0, 97: 0, 67: 1, 99: 1, 71: 2, 103: 2, 84: 3, 116: 3}.get(int(c), 0) for c in a]
        want = O.dnastring_pack(bits)
        assert np.array_equal(O.pack_acgt(a.tobytes()), want)


def test_write_gfa_small_graph():
    """Hand-checkable GFA: two overlapping reads sharing a branch (k = 5)."""
    k = 5
    #  ACGTTGCA + branch: ...TTGCAC and ...TTGCAG
    seqs = [R.from_ascii("ACGTTGCAC"), R.from_ascii("CGTTGCAGT")]
    ss = O.SeqSet.from_byte_seqs(seqs)
    t = O.filter_kmers(ss, k, O.COUNT_FILTER, 1, stranded=True)
    g = O.compress_kmers(k, True, O.SPEC_SAT_ADD, t.key_hi, t.key_lo, t.exts, t.count)
    txt = g.write_gfa().decode()
    lines = txt.strip().split("\n")
    assert lines[0] == "H\tVN:Z:debruijn-rs"
    s_lines = [l for l in lines if l.startswith("S\t")]
    l_lines = [l for l in lines if l.startswith("L\t")]
    assert len(s_lines) == len(g)
    # every S line carries its node id and an ACGT string of the node's length
    arr = g.arrays()
    for i, l in enumerate(s_lines):
        f = l.split("\t")
        assert int(f[1]) == i and len(f[2]) == int(arr["length"][i]) and set(f[2]) <= set("ACGT")
    # stranded graph: every link overlaps by k-1 and joins the right end of one node to the left end of another
    seqs_txt = [l.split("\t")[2] for l in s_lines]
    assert l_lines
    for l in l_lines:
        f = l.split("\t")
        a, b = int(f[1]), int(f[3])
        assert f[5] == "%dM" % (k - 1)
        if f[2] == "+" and f[4] == "+":
            assert seqs_txt[a][-(k - 1):] == seqs_txt[b][:k - 1]
        elif f[2] == "-" and f[4] == "-":                       # left edge of a into the right end of b
            assert seqs_txt[b][-(k - 1):] == seqs_txt[a][:k - 1]
        else:
            raise AssertionError("strand switch in a stranded graph: " + l)
