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


def _py_siphash(c, d, k0, k1, msg):
    """independent statement of SipHash-c-d (Aumasson & Bernstein 2012, figure 2.1) in Python integers"""
    M = (1 << 64) - 1
    rotl = lambda x, b: ((x << b) | (x >> (64 - b))) & M
    v = [k0 ^ 0x736f6d6570736575, k1 ^ 0x646f72616e646f6d, k0 ^ 0x6c7967656e657261, k1 ^ 0x7465646279746573]

    def rnd():
        v[0] = (v[0] + v[1]) & M; v[1] = rotl(v[1], 13) ^ v[0]; v[0] = rotl(v[0], 32)
        v[2] = (v[2] + v[3]) & M; v[3] = rotl(v[3], 16) ^ v[2]
        v[0] = (v[0] + v[3]) & M; v[3] = rotl(v[3], 21) ^ v[0]
        v[2] = (v[2] + v[1]) & M; v[1] = rotl(v[1], 17) ^ v[2]; v[2] = rotl(v[2], 32)
    n = len(msg)
    for i in range(0, n - n % 8, 8):
        m = int.from_bytes(msg[i:i + 8], "little")
        v[3] ^= m
        for _ in range(c):
            rnd()
        v[0] ^= m
    b = ((n & 0xff) << 56) | int.from_bytes(msg[n - n % 8:], "little")
    v[3] ^= b
    for _ in range(c):
        rnd()
    v[0] ^= b
    v[2] ^= 0xff
    for _ in range(d):
        rnd()
    return v[0] ^ v[1] ^ v[2] ^ v[3]


def test_siphash_published_vectors():
    """DefaultHasher is third-party to the crate (Rust std: SipHash-1-3, zero key).  The round function and the
    padding are pinned by the SipHash paper's 2-4 test values (appendix A: key 00..0f, message 00..0e ->
    a129ca6149be45e5; empty message -> 726fdb47dd0e0e31) and by the first SipHash-1-3 value of the std test-suite
    (library/core/tests/hash/sip.rs, key 00..0f, empty message -> bytes dc c4 0f 05 58 01 ac ab)."""
    k0 = int.from_bytes(bytes(range(8)), "little")
    k1 = int.from_bytes(bytes(range(8, 16)), "little")
    assert O.siphash(2, 4, k0, k1, bytes(range(15))) == 0xa129ca6149be45e5
    assert O.siphash(2, 4, k0, k1, b"") == 0x726fdb47dd0e0e31
    assert O.siphash(1, 3, k0, k1, b"").to_bytes(8, "little").hex() == "dcc40f055801acab"
    rng = np.random.default_rng(3)
    for n in list(range(0, 40)) + [63, 64, 65, 200]:
        msg = rng.integers(0, 256, n).astype(np.uint8).tobytes()
        for c, d, a, b in ((1, 3, 0, 0), (2, 4, k0, k1), (1, 3, k0, k1)):
            assert O.siphash(c, d, a, b, msg) == _py_siphash(c, d, a, b, msg)


def test_pack_hashn():
    """from_acgt_bytes_hashn (dna_string.rs:255-278): ACGT as from_acgt_bytes; any other character at pos becomes
    SipHash-1-3(0,0)(len(name) u64 LE || name || pos u64 LE) % 4 (impl Hash for [u8] and usize in core::hash)."""
    rng = np.random.default_rng(4)
    for name in [b"", b"r", b"read_000001/1", b"@A00519:12:HXXXX:1:1101:1000:2000 1:N:0:ACGT"]:
        for n in [1, 31, 32, 33, 150, 301]:
            a = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)].copy()
            a[rng.random(n) < 0.1] |= 0x20
            junk = rng.random(n) < 0.2
            a[junk] = np.frombuffer(b"NnRY.-*x", np.uint8)[rng.integers(0, 8, int(junk.sum()))]
            got = O.pack_acgt_hashn(a.tobytes(), name)
            bits = []
            for pos, c in enumerate(a.tobytes()):
                v = {65: 0, 97: 0, 67: 1, 99: 1, 71: 2, 103: 2, 84: 3, 116: 3}.get(c)
                if v is None:
                    v = _py_siphash(1, 3, 0, 0, len(name).to_bytes(8, "little") + name + pos.to_bytes(8, "little")) % 4
                bits.append(v)
            assert np.array_equal(got, O.dnastring_pack(bits))
    # without invalid characters it is from_acgt_bytes; the replacement depends on the name and on the position
    s = b"ACGTTGCAACGTTGCAACGTTGCAACGTTGCAACG"
    assert np.array_equal(O.pack_acgt_hashn(s, b"x"), O.pack_acgt(s))
    n_run = b"N" * 64
    a, b = O.unpack_acgt(O.pack_acgt_hashn(n_run, b"read1"), 0, 64), O.unpack_acgt(O.pack_acgt_hashn(n_run, b"read2"), 0, 64)
    assert a != b and len(set(a)) == 4 and a == O.unpack_acgt(O.pack_acgt_hashn(n_run, b"read1"), 0, 64)


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
