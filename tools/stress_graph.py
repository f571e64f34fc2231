"""stress: the second stage (combine + compress_graph) and the per-owner compress, repeated while other processes share the GPU"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, importlib, torch
import oracle_lib as O
from pkg import dbg
from virtual_ranks import owner_tables
from test_gpu_sharded import table_to_host
from graph_canon import graphs_equal
D = importlib.import_module("rust-debruijn_amd.distributed")
ctx = dbg.Context(0)
eng = D.HipEngine(ctx, torch.device("cuda", 0))
tag = sys.argv[1]
bad = 0
for seed in range(int(sys.argv[2]), int(sys.argv[3])):
    rng = np.random.default_rng(50000 + seed)
    k = int(rng.integers(16, 65)); world = int(rng.integers(2, 5)); kind = int(rng.integers(0, 2))
    colours = int(rng.choice([3, 20, 40, 64])) if kind else 1
    per = int(rng.choice([40, 400])); n_reads = per * world
    glen = max(n_reads * 150 // int(rng.choice([3, 30])), 400); err = float(rng.choice([0.0, 0.004]))
    shards = [dbg.synth_reads_host(n_reads=per, read_len=150, genome_len=glen, error_rate=err, stranded=False, n_colours=max(colours, 1), first_read=r * per) for r in range(world)]
    tabs, total = owner_tables(eng, shards, k, False, kind, 1)
    host = [table_to_host(ctx, t, k) for t in tabs]
    for t in tabs: eng.free_table(t)
    ogs, ds = [], []
    for t in host:
        if len(t) == 0: continue
        d = t.count.astype(np.uint32) if not kind else np.diff(t.set_off).astype(np.uint32)
        ds.append((t, d))
        ogs.append(O.compress_kmers(k, False, O.SPEC_MAX, t.key_hi, t.key_lo, t.exts, d))
    if not ogs: continue
    try:
        wantg = O.graph_combine(ogs).finish().compress_graph(False, O.SPEC_MAX)
    except Exception as e:
        continue
    for rep in range(4):
        try:
            gs = [dbg.compress_kmers_with_hash(False, dbg.SimpleCompress("max"), t, k=k, data=d, ctx=ctx) for t, d in ds]
            for g, og in zip(gs, ogs):
                if not graphs_equal(g.arrays(), og.arrays()):
                    print(tag, "seed", seed, "rep", rep, "compress_kmers MISMATCH", flush=True); bad += 1
            comb = dbg.combine_graphs(gs, ctx=ctx)
            got = dbg.compress_graph(False, dbg.SimpleCompress("max"), comb, ctx=ctx)
            if not graphs_equal(got.arrays(), wantg.arrays()):
                print(tag, "seed", seed, "rep", rep, "compress_graph MISMATCH", flush=True); bad += 1
        except Exception as e:
            print(tag, "seed", seed, "rep", rep, "k", k, "world", world, "kind", kind, "RAISED", repr(e)[:120], flush=True); bad += 1
print(tag, "done, bad =", bad, flush=True)
