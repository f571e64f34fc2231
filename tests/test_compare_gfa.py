"""tools/compare_gfa.py: the canonicalising GFA comparer a maintainer runs against a GFA written by the real crate
(DebruijnGraph::write_gfa, src/graph.rs:537-616).  Here it is exercised on GFA text of the oracle: the same k-mer index
compressed under different seed orders (= different MPHF slot orders) differs in node order, strands and cycle cuts but is the
SAME GRAPH; removing a k-mer makes it DIFFERENT."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O
import refgen as R

TOOL = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "compare_gfa.py")


def run(a, b, *extra):
    r = subprocess.run([sys.executable, TOOL, str(a), str(b), *extra], capture_output=True, text=True)
    return r.returncode, r.stdout


def gfa_of(k, stranded, t, seed_order):
    g = O.compress_kmers(k, stranded, O.SPEC_SAT_ADD, t.key_hi, t.key_lo, t.exts, t.count, seed_order)
    return g.finish().write_gfa() if hasattr(g, "finish") else g.write_gfa()


@pytest.mark.parametrize("k,stranded", [(31, False), (21, True), (47, False)])
def test_seed_orders_give_the_same_graph(tmp_path, k, stranded):
    rng = np.random.default_rng(50 + k)
    contigs = R.random_contigs(rng)
    # an isolated cycle: a circular sequence read once around plus k-1 bases
    circ = R.random_dna(rng, 60)
    contigs.append(np.concatenate([circ, circ[:k - 1 + 5]]))
    t = O.filter_kmers(O.SeqSet.from_byte_seqs(contigs), k, O.COUNT_FILTER, 1, stranded=stranded)
    files = []
    for i, so in enumerate([None, rng.permutation(t.n).astype(np.uint64), np.arange(t.n, dtype=np.uint64)[::-1].copy()]):
        p = tmp_path / ("g%d.gfa" % i)
        p.write_bytes(gfa_of(k, stranded, t, so))
        files.append(p)
    assert files[0].read_bytes() != files[1].read_bytes()               # literally different text ...
    extra = ["--stranded"] if stranded else []
    for f in files[1:]:
        rc, out = run(files[0], f, *extra)
        assert rc == 0 and "SAME GRAPH" in out, out                     # ... same graph
    # drop one k-mer from the index: different unitigs
    keep = np.ones(t.n, bool)
    keep[t.n // 2] = False
    g2 = O.compress_kmers(k, stranded, O.SPEC_SAT_ADD, t.key_hi[keep], t.key_lo[keep], t.exts[keep], t.count[keep])
    p = tmp_path / "other.gfa"
    p.write_bytes(g2.finish().write_gfa() if hasattr(g2, "finish") else g2.write_gfa())
    rc, out = run(files[0], p, *extra)
    assert rc == 1 and "DIFFERENT" in out
    rc, out = run(files[0], tmp_path / "missing.gfa")
    assert rc == 2
