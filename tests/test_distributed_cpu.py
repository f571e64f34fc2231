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


def _worker(rank, world, port, kind, stranded, out_dir, chunks, n_bins=None):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import importlib
    import oracle_lib as O
    from oracle_engine import OracleEngine
    dbg = importlib.import_module("rust-debruijn_amd")
    D = importlib.import_module("rust-debruijn_amd.distributed")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = N_READS // world
    hs = dbg.synth_reads_host(n_reads=per, read_len=150, genome_len=N_READS * 150 // 30, error_rate=0.005,
                              stranded=stranded, n_colours=4, first_read=rank * per)
    ss = O.SeqSet(hs.words, hs.start, hs.length, None, hs.data, 1)
    tab, total, n_local, n_recs = D.sharded_filter_kmers(OracleEngine(n_bins), ss, 47, stranded, kind, 2, n_chunks=chunks)
    assert total == N_READS * 104 and n_local == per * 104
    res = dict(keys=tab.keys(), exts=tab.exts.tolist(), count=tab.count.tolist(), set_off=tab.set_off.tolist(),
               set_val=tab.set_val.tolist())
    pickle.dump(res, open(os.path.join(out_dir, "rank%d.pkl" % rank), "wb"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,stranded,chunks,n_bins", [(0, False, None, None), (1, False, 1, None), (0, True, 3, None),
                                                         (0, False, None, 5), (1, False, 3, 3)])
def test_sharded_counting_gloo_world2(tmp_path, kind, stranded, chunks, n_bins):
    """chunks = ranges the owned bins are exchanged and counted in (None = the default pipeline depth); n_bins = 5 or 3:
    the two ranks own different numbers of bins (2 + 3, 1 + 2), fewer than the pipeline depth -- every rank must still
    arrive at the same number of exchange rounds and the same cuts"""
    import importlib
    import oracle_lib as O
    O.build()
    dbg = importlib.import_module("rust-debruijn_amd")
    port = 29600 + kind * 2 + int(stranded) + (n_bins or 0) * 4 + (os.getpid() % 200)
    mp.spawn(_worker, args=(WORLD, port, kind, stranded, str(tmp_path), chunks, n_bins), nprocs=WORLD, join=True)
    parts = [pickle.load(open(tmp_path / ("rank%d.pkl" % r), "rb")) for r in range(WORLD)]
    hs = dbg.synth_reads_host(n_reads=N_READS, read_len=150, genome_len=N_READS * 150 // 30, error_rate=0.005,
                              stranded=stranded, n_colours=4)
    want = O.filter_kmers(O.SeqSet(hs.words, hs.start, hs.length, None, hs.data, 1), 47, kind, 2, stranded=stranded)
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
