"""Seeded restatements of the reference's random test-input generators
(/root/reference/src/test.rs:13-132) plus small pure-Python k-mer helpers used
to cross-check the oracle independently of its own bit tricks.

The reference uses an unseeded thread_rng (test.rs:20,37,47,100); here every
generator takes a numpy Generator so failures are reproducible.
"""
import numpy as np

ASCII = "ACGT"


def random_dna(rng, n):                       # test.rs:19-25
    return rng.integers(0, 4, size=n, dtype=np.uint8)


def to_ascii(b):
    return "".join(ASCII[int(x)] for x in b)


def from_ascii(s):
    return np.array([ASCII.index(c) for c in s], dtype=np.uint8)


def revcomp_bytes(b):
    return (3 - np.asarray(b, dtype=np.uint8))[::-1].copy()


def simple_random_contigs(rng):               # test.rs:58-95
    p1, p2, pc, p3, p4 = (random_dna(rng, n) for n in (40, 30, 100, 30, 40))
    c1 = np.concatenate([p1, pc, p3])
    c2 = np.concatenate([p2, pc, p4])
    pal1 = random_dna(rng, 33)
    pal2 = revcomp_bytes(pal1)
    c3 = np.concatenate([random_dna(rng, 30), pal1, pal2, random_dna(rng, 50)])
    return [c1, c2, c3]


def random_contigs(rng):                      # test.rs:98-132
    nchunks = max(5, int(rng.gamma(0.6, 25.0)))
    chunks = [random_dna(rng, max(10, int(rng.gamma(1.5, 200.0)))) for _ in range(nchunks)]
    nchrom = max(4, int(rng.gamma(0.6, 25.0)))
    chroms = []
    for _ in range(nchrom):
        chrom_chunks = max(4, int(rng.gamma(0.6, 25.0)))
        parts = [chunks[int(rng.integers(0, nchunks))] for _ in range(chrom_chunks)]
        chroms.append(np.concatenate(parts))
    return chroms


DEGEN = "AAAAATAAAATAAAATAAAATAAAATAAAATAAAATAAAATAAAA"   # test.rs:171,184


# ---- pure-python k-mer model (strings / python ints) --------------------------------------
def kmer_int(b):
    """bytes 0-3 -> integer, base 0 most significant (kmer.rs:429-437)."""
    v = 0
    for x in b:
        v = (v << 2) | int(x)
    return v


def kmer_rc_int(k, v):
    out = 0
    for _ in range(k):
        out = (out << 2) | (3 - (v & 3))
        v >>= 2
    return out


def canon(k, v):
    r = kmer_rc_int(k, v)
    return v if v < r else r


def kmers_of(b, k):
    b = [int(x) for x in b]
    if len(b) < k:
        return []
    v = kmer_int(b[:k])
    mask = (1 << (2 * k)) - 1
    out = [v]
    for x in b[k:]:
        v = ((v << 2) | x) & mask
        out.append(v)
    return out


def exts_rc_py(e):
    """lib.rs:729-748: swap nibbles, reverse the 4 bits inside each nibble."""
    l, r = e & 0xF, e >> 4
    rev4 = lambda x: ((x & 1) << 3) | ((x & 2) << 1) | ((x & 4) >> 1) | ((x & 8) >> 3)
    return (rev4(l) << 4) | rev4(r)


def naive_filter(seqs, seq_exts, k, stranded):
    """Dictionary model of filter_kmers' per-k-mer result: {canonical kmer: (count, exts)}.
    Built from lib.rs:812-841 + filter.rs:190-196 semantics with python ints only."""
    table = {}
    for s, e in zip(seqs, seq_exts):
        s = [int(x) for x in s]
        n = len(s)
        if n < k:
            continue
        ks = kmers_of(s, k)
        for j, v in enumerate(ks):
            left = (e & 0x0F) if j == 0 else (1 << s[j - 1])
            right = (e & 0xF0) if j + k == n else (1 << (4 + s[j + k]))
            ex = left | right
            if not stranded:
                r = kmer_rc_int(k, v)
                if not (v < r):
                    v, ex = r, exts_rc_py(ex)
            c, oe = table.get(v, (0, 0))
            table[v] = (c + 1, oe | ex)
    return table
