import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np, importlib, torch
import oracle_lib as O
from pkg import dbg
from virtual_ranks import owner_tables
from test_gpu_sharded import table_to_host
D = importlib.import_module("rust-debruijn_amd.distributed")
seed = int(sys.argv[1])
rng = np.random.default_rng(50000 + seed)
k = int(rng.integers(16, 65)); world = int(rng.integers(2, 5)); kind = int(rng.integers(0, 2))
colours = int(rng.choice([3, 20, 40, 64])) if kind else 1
per = int(rng.choice([40, 400])); n_reads = per * world
glen = max(n_reads * 150 // int(rng.choice([3, 30])), 400); err = float(rng.choice([0.0, 0.004]))
print("k", k, "world", world, "kind", kind, "colours", colours, "per", per, "glen", glen, "err", err)
shards = [dbg.synth_reads_host(n_reads=per, read_len=150, genome_len=glen, error_rate=err, stranded=False, n_colours=max(colours, 1), first_read=r * per) for r in range(world)]
ctx = dbg.Context(0)
eng = D.HipEngine(ctx, torch.device("cuda", 0))
for merge in (False, True):
    tabs, total = owner_tables(eng, shards, k, False, kind, 1, merge_dups=merge)
    host = [table_to_host(ctx, t, k) for t in tabs]
    for t in tabs: eng.free_table(t)
    gs, ogs = [], []
    for t in host:
        if len(t) == 0: continue
        d = t.count.astype(np.uint32) if not kind else np.diff(t.set_off).astype(np.uint32)
        gs.append(dbg.compress_kmers_with_hash(False, dbg.SimpleCompress("max"), t, k=k, data=d, ctx=ctx))
        ogs.append(O.compress_kmers(k, False, O.SPEC_MAX, t.key_hi, t.key_lo, t.exts, d))
    try:
        wantg = O.graph_combine(ogs).finish().compress_graph(False, O.SPEC_MAX)
        print("merge", merge, "oracle ok")
    except Exception as e:
        print("merge", merge, "oracle raised", repr(e)[:200])
    try:
        got = dbg.compress_graph(False, dbg.SimpleCompress("max"), dbg.combine_graphs(gs, ctx=ctx), ctx=ctx)
        print("merge", merge, "gpu ok")
    except Exception as e:
        print("merge", merge, "gpu raised", repr(e)[:200])
