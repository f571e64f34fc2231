"""GPU parity of the sharded second stage: BaseGraph::combine + compress_graph (finish, fix_exts, find_link,
CompressFromGraph -- src/graph.rs:71-142,252-377; src/compression.rs:100-349) against the oracle,
literal BaseGraph equality; plus the reference's own sharded pipeline (src/test.rs:418-504) end to end on
the GPU entry points."""
import numpy as np
import pytest

import oracle_lib as O
import refgen as R
from graph_canon import canonical_nodes, graph_kmer_set, graphs_equal
from pkg import dbg

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["device", "host"])
def compress_mode(request, ctx):
    """Every test runs with the node walk of compress_graph (and the k-mer walk of compress_kmers_with_hash) on the
    device (DBG_COMPRESS=device: chains by pointer jumping; it fails instead of falling back) and on the host."""
    import os
    old = ctx.set_option("DBG_COMPRESS", request.param)
    yield request.param
    ctx.set_option("DBG_COMPRESS", old)


@pytest.fixture(scope="module")
def ctx():
    c = dbg.Context(0)
    yield c
    c.close()


def to_oracle_graph(g):
    a = g.arrays()
    return O.graph_from_arrays(g.k, g.stranded, np.concatenate([a["words"], np.zeros(2, np.uint64)]), a["start"], a["length"],
                               a["exts"], a["data"])


def sharded_pipeline(ctx, contigs, k, p=6):
    """msp_sequence -> per-shard filter_kmers(CountFilter(2)) -> compress_kmers_with_hash -> combine ->
    compress_graph(false, max), all through the product; returns (gpu graph, shard graphs)."""
    shards = {}
    res = dbg.msp_sequence_batch(k, [c for c in contigs if len(c) >= k], None, True, p, 0, ctx)
    cl = [c for c in contigs if len(c) >= k]
    for i, c in enumerate(cl):
        for j in range(int(res["piece_off"][i]), int(res["piece_off"][i + 1])):
            s, l = int(res["start"][j]), int(res["len"][j])
            v = shards.setdefault(int(res["bucket"][j]), [])
            for d in (0, 1):
                v.append((c[s:s + l], int(res["exts"][j]), d))
    graphs = []
    for b in sorted(shards):
        t, _ = dbg.filter_kmers(shards[b], dbg.CountFilter(2), False, False, 4, k=k, ctx=ctx)
        graphs.append(dbg.compress_kmers_with_hash(False, dbg.SimpleCompress("saturating_add"), t, k=k, ctx=ctx))
    comb = dbg.combine_graphs(graphs, ctx=ctx)
    return dbg.compress_graph(False, dbg.SimpleCompress("max"), comb, ctx=ctx), graphs, comb


@pytest.mark.parametrize("k,seed", [(32, 1), (31, 2), (47, 3), (63, 4)])
def test_sharded_second_stage_parity(ctx, k, seed):
    rng = np.random.default_rng(seed)
    contigs = R.random_contigs(rng) if seed > 1 else R.simple_random_contigs(rng)
    got, shard_graphs, comb = sharded_pipeline(ctx, contigs, k)
    # oracle: same shard graphs -> combine -> finish -> compress_graph
    og = O.graph_combine([to_oracle_graph(g) for g in shard_graphs])
    assert graphs_equal(comb.arrays(), og.arrays())                              # BaseGraph::combine
    want = og.finish().compress_graph(False, O.SPEC_MAX)
    assert graphs_equal(got.arrays(), want.arrays()), "compress_graph differs from the oracle"
    # the reference's assertions (test.rs:476-503)
    truth = set()
    for c in contigs:
        truth.update(R.canon(k, v) for v in R.kmers_of(c, k))
    assert graph_kmer_set(got.arrays(), k, False) == truth
    assert to_oracle_graph(got).is_compressed(O.SPEC_MAX) is None


def test_compress_graph_from_single_kmer_nodes(ctx):
    """test.rs:257-274: one node per k-mer, then compress_graph must give a fully compressed graph equal (up
    to orientation/order) to compress_kmers_with_hash on the same table."""
    rng = np.random.default_rng(9)
    contigs = R.random_contigs(rng)
    k = 31
    t, _ = dbg.filter_kmers([(c, 0, None) for c in contigs], dbg.CountFilter(1), False, False, 4, k=k, ctx=ctx)
    seqs = [[(v >> (2 * (k - 1 - j))) & 3 for j in range(k)] for v in t.keys()]
    ug = dbg.BaseGraph(k, dbg.PackedDnaStringSet.from_seqs(seqs), t.exts.copy(), np.ones(len(t), np.uint32), False)
    got = dbg.compress_graph(False, dbg.SimpleCompress("wrapping_add"), ug, ctx=ctx)
    want = to_oracle_graph(ug).finish().compress_graph(False, O.SPEC_WRAP_ADD)
    assert graphs_equal(got.arrays(), want.arrays())
    direct = dbg.compress_kmers_with_hash(False, dbg.SimpleCompress("wrapping_add"), t, k=k, data=np.ones(len(t), np.uint32), ctx=ctx)
    assert canonical_nodes(got.arrays(), k, False) == canonical_nodes(direct.arrays(), k, False)


def test_compress_graph_with_censored_nodes(ctx):
    """tip cleaning shape (test.rs:506-572): censor short nodes, exts towards them are dropped by fix_exts."""
    rng = np.random.default_rng(5)
    base = R.random_dna(rng, 300)
    tip = np.concatenate([base[:150], R.random_dna(rng, 8)])
    k = 31
    seqs = [(base, 0, None)] * 5 + [(tip, 0, None)] * 2
    t, _ = dbg.filter_kmers(seqs, dbg.CountFilter(2), False, False, 4, k=k, ctx=ctx)
    g = dbg.compress_kmers_with_hash(False, dbg.SimpleCompress("wrapping_add"), t, k=k, ctx=ctx)
    censor = [i for i in range(len(g)) if int(g.sequences.length[i]) < 2 * k]
    assert censor
    got = dbg.compress_graph(False, dbg.SimpleCompress("wrapping_add"), g, censor_nodes=censor, ctx=ctx)
    want = to_oracle_graph(g).finish().compress_graph(False, O.SPEC_WRAP_ADD, censor=censor)
    assert graphs_equal(got.arrays(), want.arrays())
    assert len(got) < len(g)


def test_combine_rejects_mixed_strandedness(ctx):
    a = dbg.BaseGraph(31, dbg.PackedDnaStringSet.from_seqs([R.random_dna(np.random.default_rng(1), 40)]), np.zeros(1, np.uint8),
                      np.zeros(1, np.uint32), True)
    b = dbg.BaseGraph(31, dbg.PackedDnaStringSet.from_seqs([R.random_dna(np.random.default_rng(2), 40)]), np.zeros(1, np.uint8),
                      np.zeros(1, np.uint32), False)
    with pytest.raises(dbg.DbgError):                                            # graph.rs:89-91
        dbg.combine_graphs([a, b], ctx=ctx)
