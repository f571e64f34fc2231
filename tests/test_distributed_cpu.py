"""world_size-2 gloo run of the sharded-counting orchestration (rust-debruijn_amd/distributed.py)
on CPU, with the oracle as the per-rank compute engine: the union of the per-rank tables must equal
filter_kmers over all reads, and every k-mer must live on exactly one rank."""
import os
import pickle
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
WORLD = 2
N_READS = 120


SPARSE = np.array([3, 70, 130, 255], dtype=np.uint8)      # a label alphabet beyond the counting kernel's 64 colours


def _worker(rank, world, port, kind, stranded, out_dir, chunks, n_bins=None, sparse=False):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import importlib
    import oracle_lib as O
    from oracle_engine import OracleEngine
    dbg = importlib.import_module("rust-debruijn_amd")
    import model_orchestration as D                          # the engine-agnostic Python model of the flow (tests/model_orchestration.py)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = N_READS // world
    hs = dbg.synth_reads_host(n_reads=per, read_len=150, genome_len=N_READS * 150 // 30, error_rate=0.005,
                              stranded=stranded, n_colours=4, first_read=rank * per)
    ss = O.SeqSet(hs.words, hs.start, hs.length, None, SPARSE[hs.data] if sparse else hs.data, 1)
    eng = OracleEngine(n_bins)
    tab, total, n_local, n_recs = D.sharded_filter_kmers(eng, ss, 47, stranded, kind, 2, n_chunks=chunks)
    assert total == N_READS * 104 and n_local == per * 104
    # labels >= 64: every rank is handed the alphabet of the whole job (a max-reduction of presence flags)
    assert eng.labels == (SPARSE.tolist() if sparse and kind == 1 else None)
    res = dict(keys=tab.keys(), exts=tab.exts.tolist(), count=tab.count.tolist(), set_off=tab.set_off.tolist(),
               set_val=tab.set_val.tolist())
    pickle.dump(res, open(os.path.join(out_dir, "rank%d.pkl" % rank), "wb"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,stranded,chunks,n_bins,sparse", [(0, False, None, None, False), (1, False, 1, None, False), (0, True, 3, None, False),
                                                                (0, False, None, 5, False), (1, False, 3, 3, False), (1, False, None, None, True)])
def test_sharded_counting_gloo_world2(tmp_path, kind, stranded, chunks, n_bins, sparse):
    """chunks = ranges the owned bins are exchanged and counted in (None = the default pipeline depth); n_bins = 5 or 3:
    the two ranks own different numbers of bins (2 + 3, 1 + 2), fewer than the pipeline depth -- every rank must still
    arrive at the same number of exchange rounds and the same cuts"""
    import importlib
    import oracle_lib as O
    O.build()
    dbg = importlib.import_module("rust-debruijn_amd")
    port = 29600 + kind * 2 + int(stranded) + (n_bins or 0) * 4 + int(sparse) * 64 + (os.getpid() % 200)
    mp.spawn(_worker, args=(WORLD, port, kind, stranded, str(tmp_path), chunks, n_bins, sparse), nprocs=WORLD, join=True)
    parts = [pickle.load(open(tmp_path / ("rank%d.pkl" % r), "rb")) for r in range(WORLD)]
    hs = dbg.synth_reads_host(n_reads=N_READS, read_len=150, genome_len=N_READS * 150 // 30, error_rate=0.005,
                              stranded=stranded, n_colours=4)
    want = O.filter_kmers(O.SeqSet(hs.words, hs.start, hs.length, None, SPARSE[hs.data] if sparse else hs.data, 1), 47, kind, 2, stranded=stranded)
    merged = {}
    for p in parts:
        assert p["keys"] == sorted(p["keys"])                       # each rank's table is ascending
        for i, key in enumerate(p["keys"]):
            assert key not in merged, "k-mer counted on two ranks"
            vals = p["set_val"][p["set_off"][i]:p["set_off"][i + 1]] if kind == 1 else p["count"][i]
            merged[key] = (p["exts"][i], vals)
    assert sorted(merged) == want.keys()
    assert all(len(p["keys"]) > 0 for p in parts)                  # both ranks own work
    for i, key in enumerate(want.keys()):
        e, v = merged[key]
        assert e == int(want.exts[i])
        if kind == 1:
            assert v == [int(x) for x in want.set_val[int(want.set_off[i]):int(want.set_off[i + 1])]]
        else:
            assert v == int(want.count[i])


def _compress_worker(rank, world, port, kind, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import importlib
    import oracle_lib as O
    from oracle_engine import OracleEngine
    dbg = importlib.import_module("rust-debruijn_amd")
    import model_orchestration as D                          # the engine-agnostic Python model of the flow (tests/model_orchestration.py)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = N_READS // world
    k = 51 if kind == 1 else 47
    hs = dbg.synth_reads_host(n_reads=per, read_len=150, genome_len=N_READS * 150 // 30, error_rate=0.005,
                              stranded=False, n_colours=3, first_read=rank * per)
    ss = O.SeqSet(hs.words, hs.start, hs.length, None, hs.data, 1)
    eng = OracleEngine()
    tab, total, n_local, n_recs = D.sharded_filter_kmers(eng, ss, k, False, kind, 2)
    spec = dbg.ScmapCompress() if kind == 1 else dbg.SimpleCompress("saturating_add")
    final, local = D.sharded_compress(eng, tab, k, False, spec, dst=0, second_spec=spec if kind == 1 else dbg.SimpleCompress("max"))
    res = dict(local=D._graph_payload(local), tab=dict(key_hi=tab.key_hi, key_lo=tab.key_lo, exts=tab.exts, count=tab.count,
                                                       set_off=tab.set_off, set_val=tab.set_val))
    if rank == 0:
        assert final is not None
        res["final"] = D._graph_payload(final)
    else:
        assert final is None
    pickle.dump(res, open(os.path.join(out_dir, "crank%d.pkl" % rank), "wb"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", [0, 1])
def test_sharded_compress_gloo_world2(tmp_path, kind):
    """The rank-spanning end of the pipeline (test.rs:459-470) over gloo, world size 2: per-rank compress of the owned bins,
    graphs gathered on rank 0, BaseGraph::combine + compress_graph there.  kind 1 = the config-5 shape: k = 51,
    CountFilterSet labels -> rank-local classes -> ScmapCompress, classes unified before the second stage.  The result must
    be what the same flow gives when it is run in one process straight on the oracle."""
    import importlib
    import oracle_lib as O
    from graph_canon import graphs_equal
    O.build()
    import model_orchestration as D                          # the engine-agnostic Python model of the flow (tests/model_orchestration.py)
    port = 29900 + kind + (os.getpid() % 90)
    mp.spawn(_compress_worker, args=(WORLD, port, kind, str(tmp_path)), nprocs=WORLD, join=True)
    parts = [pickle.load(open(tmp_path / ("crank%d.pkl" % r), "rb")) for r in range(WORLD)]
    k = 51 if kind == 1 else 47
    # the same flow, single process: per-rank tables -> compress_kmers -> combine -> compress_graph
    glob = sorted(set(tuple(int(x) for x in p["tab"]["set_val"][int(p["tab"]["set_off"][i]):int(p["tab"]["set_off"][i + 1])])
                      for p in parts for i in range(len(p["tab"]["key_lo"])))) if kind == 1 else None
    shard_graphs = []
    for p in parts:
        t = p["tab"]
        if kind == 1:
            pos = {s: i for i, s in enumerate(glob)}
            data = np.array([pos[tuple(int(x) for x in t["set_val"][int(t["set_off"][i]):int(t["set_off"][i + 1])])] for i in range(len(t["key_lo"]))],
                            dtype=np.uint32)
        else:
            data = t["count"]
        shard_graphs.append(O.compress_kmers(k, False, O.SPEC_SCMAP_EQ if kind == 1 else O.SPEC_SAT_ADD, t["key_hi"], t["key_lo"], t["exts"], data))
    want = O.graph_combine(shard_graphs).finish().compress_graph(False, O.SPEC_SCMAP_EQ if kind == 1 else O.SPEC_MAX)
    got = parts[0]["final"]
    assert graphs_equal(got, want.arrays())
    if kind == 1:
        assert got["classes"] == glob and len(glob) > 1
    assert len(got["start"]) < sum(len(p["local"]["start"]) for p in parts)     # the second stage joined unitigs across shards


def test_second_stage_leaves_the_shard_graphs_alone():
    """world size 1 hands `local` itself to second_stage and returns it: data and classes of the shard graph must still
    belong together afterwards (rank-local ids into the rank-local table), whatever numbering the second stage uses."""
    import importlib
    import oracle_lib as O
    from oracle_engine import OracleEngine
    O.build()
    dbg = importlib.import_module("rust-debruijn_amd")
    import model_orchestration as D                          # the engine-agnostic Python model of the flow (tests/model_orchestration.py)
    hs = dbg.synth_reads_host(n_reads=N_READS, read_len=150, genome_len=N_READS * 150 // 30, error_rate=0.005, stranded=False, n_colours=3)
    ss = O.SeqSet(hs.words, hs.start, hs.length, None, hs.data, 1)
    eng = OracleEngine()
    tab, _, _, _ = D.sharded_filter_kmers(eng, ss, 51, False, 1, 2)
    ref = eng.compress_table(tab, 51, False, dbg.ScmapCompress())
    # a table whose rank-local numbering is NOT the sorted-tuple numbering, as the HIP engine's (ordered by mask) is not
    perm = np.arange(len(ref.classes))[::-1].copy()
    shuffled = [ref.classes[i] for i in np.argsort(perm)]
    before = [ref.classes[int(c)] for c in ref.data]

    class Eng(OracleEngine):
        def compress_table(self, tab, k, stranded, spec):
            g = OracleEngine.compress_table(self, tab, k, stranded, spec)
            g.data = perm[np.asarray(g.data, dtype=np.int64)].astype(np.uint32)
            g.classes = shuffled
            return g
    final, local = D.sharded_compress(Eng(), tab, 51, False, dbg.ScmapCompress())
    assert local.classes == shuffled
    assert [local.classes[int(c)] for c in local.data] == before
    assert final.classes == sorted(set(before)) and len(final) <= len(local)
