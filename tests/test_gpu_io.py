"""GPU parity for the data formats either side of the path: dbg_pack_acgt / dbg_unpack_acgt vs the oracle's
from_acgt_bytes / to_ascii_vec (dna_string.rs:222-250, :297-299; bitops_avx2.rs:138-216), and dbg_graph_edges /
dbg_graph_to_gfa vs the oracle's write_gfa (graph.rs:223-291, :537-611): byte-identical text."""
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


def test_pack_known_and_edge_lengths(ctx):
    for s in [b"", b"A", b"ACGTACGT", b"C" * 32, b"ACGT" * 8, b"AAAAAAAACCCCCCCCGGGGGGGGTTTTTTTT", b"ACGTAAAAAAAAAATTATATAACGT",
              b"acgtnNxX" * 9, b"T" * 31, b"G" * 33, b"T" * 64, b"T" * 65]:
        w, bad = dbg.pack_acgt(s, ctx)
        assert np.array_equal(w, O.pack_acgt(s))
        assert bad == sum(1 for c in s if c not in b"ACGTacgt")
        up = dbg.unpack_acgt(w, 0, len(s), ctx)
        assert up == O.unpack_acgt(w, 0, len(s))
        assert up == bytes(b"ACGT"[{65: 0, 97: 0, 67: 1, 99: 1, 71: 2, 103: 2, 84: 3, 116: 3}.get(c, 0)] for c in s)


def test_pack_random_bytes(ctx):
    rng = np.random.default_rng(11)
    for n in [1, 31, 32, 33, 1000, 4096, 100_003, 1_000_000]:
        # mostly ACGT, some lowercase, some arbitrary bytes (every byte value appears)
        a = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)].copy()
        lower = rng.random(n) < 0.1
        a[lower] |= 0x20
        junk = rng.random(n) < 0.05
        a[junk] = rng.integers(0, 256, int(junk.sum())).astype(np.uint8)
        w, bad = dbg.pack_acgt(a, ctx)
        assert np.array_equal(w, O.pack_acgt(a.tobytes()))
        valid = np.isin(a, np.frombuffer(b"ACGTacgt", np.uint8))
        assert bad == int((~valid).sum())
        first = int(rng.integers(0, n))
        m = int(rng.integers(0, n - first + 1))
        assert dbg.unpack_acgt(w, first, m, ctx) == O.unpack_acgt(w, first, m)


def test_pack_hashn_batch(ctx):
    """dbg_pack_acgt_hashn vs the oracle's from_acgt_bytes_hashn (dna_string.rs:255-278), read by read"""
    rng = np.random.default_rng(21)
    reads, names = [], []
    for i in range(300):
        n = int(rng.choice([0, 1, 5, 31, 32, 33, 100, 150, 151, 400]))
        a = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)].copy()
        a[rng.random(n) < 0.1] |= 0x20
        p_junk = float(rng.choice([0.0, 0.01, 0.3, 1.0]))
        junk = rng.random(n) < p_junk
        a[junk] = rng.integers(0, 256, int(junk.sum())).astype(np.uint8)
        reads.append(a.tobytes())
        names.append(b"" if i % 17 == 0 else b"@run7:%d:%d/1" % (i, int(rng.integers(0, 10 ** int(rng.integers(1, 12))))))
    words, start, rep = dbg.pack_acgt_hashn(reads, names, ctx)
    assert rep == sum(1 for r in reads for c in r if c not in b"ACGTacgt")
    for r, nm, st in zip(reads, names, start):
        want = O.unpack_acgt(O.pack_acgt_hashn(r, nm), 0, len(r))
        assert dbg.unpack_acgt(words, int(st), len(r), ctx) == want
    # no invalid characters: identical to dbg_pack_acgt
    clean = [bytes(b"ACGT"[x] for x in rng.integers(0, 4, 150)) for _ in range(50)]
    w1, _, rep = dbg.pack_acgt_hashn(clean, [b"n"] * 50, ctx)
    assert rep == 0 and np.array_equal(w1, dbg.pack_acgt(b"".join(clean), ctx)[0])
    assert dbg.pack_acgt_hashn([], [], ctx)[2] == 0
    hs = dbg.seqs_from_acgt(reads, ctx=ctx, names=names)
    assert np.array_equal(hs.words, words)


def test_ascii_reads_through_filter(ctx):
    """ASCII reads -> GPU packer -> filter_kmers equals the same reads given as 0-3 bytes."""
    rng = np.random.default_rng(5)
    contigs = R.random_contigs(rng)
    ascii_reads = [R.to_ascii(c).encode() for c in contigs]
    hs = dbg.seqs_from_acgt(ascii_reads, ctx=ctx)
    t1, _ = dbg.filter_kmers(hs, dbg.CountFilter(1), False, False, 4, k=31, ctx=ctx)
    t2, _ = dbg.filter_kmers([(c, 0, None) for c in contigs], dbg.CountFilter(1), False, False, 4, k=31, ctx=ctx)
    assert np.array_equal(t1.key_lo, t2.key_lo) and np.array_equal(t1.exts, t2.exts) and np.array_equal(t1.count, t2.count)


@pytest.mark.parametrize("k,stranded", [(31, False), (31, True), (47, False), (64, False), (16, False)])
def test_gfa_matches_oracle(ctx, k, stranded, tmp_path):
    rng = np.random.default_rng(100 + k + stranded)
    for trial in range(2):
        contigs = R.random_contigs(rng) if trial else R.simple_random_contigs(rng)
        t, _ = dbg.filter_kmers([(c, 0, None) for c in contigs], dbg.CountFilter(1), stranded, False, 4, k=k, ctx=ctx)
        g = dbg.compress_kmers_with_hash(stranded, dbg.SimpleCompress("saturating_add"), t, k=k, ctx=ctx)
        ga = g.arrays()
        og = O.graph_from_arrays(k, stranded, ga["words"], ga["start"], ga["length"], ga["exts"], ga["data"])
        want = og.write_gfa()
        got = g.write_gfa(ctx)
        assert got == want
        assert got.count(b"\nS\t") == len(g) and got.startswith(b"H\tVN:Z:debruijn-rs\n")
        # to_gfa writes the same bytes to a file (graph.rs:598-601)
        p = tmp_path / ("g%d.gfa" % trial)
        g.to_gfa(p, ctx)
        assert p.read_bytes() == want
        # edges(): consistent with the L lines
        ed = g.edges(ctx)
        n_l = sum(1 for i, (l, r) in enumerate(ed) for e in l if e[0] >= i) + sum(1 for i, (l, r) in enumerate(ed) for e in r if e[0] > i)
        assert n_l == got.count(b"\nL\t")


def test_gfa_uncompressed_and_empty(ctx):
    """One node per k-mer (test.rs:257-264 shape): many short nodes, every overlap is an edge; and the empty graph."""
    k = 21
    rng = np.random.default_rng(9)
    contigs = R.simple_random_contigs(rng)
    t, _ = dbg.filter_kmers([(c, 0, None) for c in contigs], dbg.CountFilter(1), False, False, 4, k=k, ctx=ctx)
    n = len(t)
    ps = dbg.PackedDnaStringSet.from_seqs([np.array([(v >> (2 * (k - 1 - i))) & 3 for i in range(k)], np.uint8) for v in t.keys()])
    g = dbg.BaseGraph(k, ps, t.exts.copy(), t.count.astype(np.uint32), False)
    og = O.graph_from_arrays(k, False, ps.words, ps.start, ps.length, g.exts, g.data)
    assert g.write_gfa(ctx) == og.write_gfa()
    empty = dbg.BaseGraph(k, dbg.PackedDnaStringSet(np.zeros(2, np.uint64), np.zeros(0, np.uint64), np.zeros(0, np.uint32), 0),
                          np.zeros(0, np.uint8), np.zeros(0, np.uint32), False)
    assert empty.write_gfa(ctx) == b"H\tVN:Z:debruijn-rs\n"


@pytest.mark.parametrize("k,stranded,dw", [(31, False, 2), (47, False, 4), (63, True, 2)])
def test_serde_of_a_gpu_built_graph(ctx, k, stranded, dw):
    """serde forms (graph.rs:43-50 derives) of a graph built by the GPU path: serialise, deserialise, literal equality; the
    JSON text is also read by Python's json module -- an independent reader -- and must describe the same BaseGraph, field
    by field (sequences.sequence.storage/len, start, length, exts[].val, data, stranded, phantom = null)."""
    import json
    rng = np.random.default_rng(700 + k)
    contigs = R.random_contigs(rng)
    t, _ = dbg.filter_kmers([(c, 0, None) for c in contigs], dbg.CountFilter(1), stranded, False, 4, k=k, ctx=ctx)
    g = dbg.compress_kmers_with_hash(stranded, dbg.SimpleCompress("saturating_add"), t, k=k, ctx=ctx)
    a = g.arrays()
    nw = (a["n_bases"] + 31) // 32
    for fmt in (dbg.SERDE_BINCODE, dbg.SERDE_JSON):
        blob = dbg.graph_serialize(g, fmt, data_width=dw)
        h = dbg.graph_deserialize(blob, k, fmt, data_width=dw)
        b = h.arrays()
        assert a["n_bases"] == b["n_bases"] and g.stranded == h.stranded
        assert np.array_equal(a["words"][:nw], b["words"][:nw])
        for name in ("start", "length", "exts", "data"):
            assert np.array_equal(a[name], b[name]), name
        assert dbg.graph_serialize(h, fmt, data_width=dw) == blob
        # the deserialised graph is the same graph to the rest of the library
        assert h.write_gfa(ctx) == g.write_gfa(ctx)
    j = json.loads(dbg.graph_serialize(g, dbg.SERDE_JSON, data_width=dw).decode())
    assert list(j) == ["sequences", "exts", "data", "stranded", "phantom"] and j["phantom"] is None and j["stranded"] is stranded
    assert list(j["sequences"]) == ["sequence", "start", "length"] and list(j["sequences"]["sequence"]) == ["storage", "len"]
    assert j["sequences"]["sequence"]["len"] == a["n_bases"]
    assert j["sequences"]["sequence"]["storage"] == [int(x) for x in a["words"][:nw]]
    assert j["sequences"]["start"] == [int(x) for x in a["start"]] and j["sequences"]["length"] == [int(x) for x in a["length"]]
    assert [e["val"] for e in j["exts"]] == [int(x) for x in a["exts"]] and j["data"] == [int(x) for x in a["data"]]
    # every node sequence read back through the independent reader spells k-mers of the index
    words = np.array(j["sequences"]["sequence"]["storage"], dtype=np.uint64)
    truth = set(t.keys())
    for s0, ln in list(zip(j["sequences"]["start"], j["sequences"]["length"]))[:50]:
        bases = O.unpack_bases(np.concatenate([words, np.zeros(2, np.uint64)]), s0, ln)
        for v in R.kmers_of(bases, k):
            assert (v if stranded else R.canon(k, v)) in truth
