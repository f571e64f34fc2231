"""Run under torch.distributed.run: the whole sharded pipeline through the two rank-spanning entry points of the C ABI --
dbg_shard_filter_kmers_dev (all-to-all of minimizer-bin records) and dbg_shard_compress_dev (per-rank compress_kmers_with_hash,
device-resident graphs over the transport, BaseGraph::combine + compress_graph; src/test.rs:433-470) -- checked on rank 0
against the oracle's same flow on the same per-rank tables.  Both BASELINE
shapes: k = 47 CountFilter/saturating_add (config 4's flow) and k = 51 CountFilterSet -> ScmapCompress (config 5)."""
import argparse
import ctypes as C
import importlib
import os
import pickle
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--one-device", action="store_true")
    ap.add_argument("--reads", type=int, default=20000)
    ap.add_argument("--check-table", action="store_true",
                    help="also compare the union of the per-rank tables with the oracle's filter_kmers over all reads (CPU: ~20 s per 10^6 reads)")
    ap.add_argument("--reduce", default="gather", choices=["gather", "tree"],
                    help="gather: all shard graphs to rank 0, one combine + compress_graph (the reference's flow literally: compared node "
                         "for node).  tree: pairwise combine + compress_graph up a binary tree (same unitigs in another order / strand: "
                         "compared in canonical form)")
    a = ap.parse_args()
    dbg = importlib.import_module("rust-debruijn_amd")
    capi = importlib.import_module("rust-debruijn_amd._capi")
    D = importlib.import_module("rust-debruijn_amd.distributed")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    lr = 0 if a.one_device else int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(lr)
    if a.backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
    else:
        dist.init_process_group("gloo")
    dev = torch.device("cuda", lr)
    ctx = dbg.Context(lr)
    ctx.set_option("DBG_COMPRESS", "device")
    eng = D.HipEngine(ctx, dev)
    for kind, k in ((0, 47), (1, 51)):
        per = a.reads // world
        hs = dbg.synth_reads_host(n_reads=per, read_len=150, genome_len=a.reads * 150 // 30, error_rate=0.001, stranded=False,
                                  n_colours=5, first_read=rank * per)
        ss, keep = dbg.upload_seqs(hs if kind else dbg.HostSeqs(hs.words, hs.start, hs.length), lr)
        tab, total, n_local, n_recs = D.sharded_filter_kmers(eng, ss, k, False, kind, 2)
        h = capi.KmerTable()
        ctx.check(ctx.lib.dbg_table_to_host(ctx.h, C.byref(tab), C.byref(h)))
        th = dbg._table_from_c(h, k)
        ctx.lib.dbg_free_table(ctx.h, C.byref(h))
        spec = dbg.ScmapCompress() if kind else dbg.SimpleCompress("saturating_add")
        spec2 = spec if kind else dbg.SimpleCompress("max")
        final, local = D.shard_compress_c(ctx, D.transport_for(eng), tab, k, False, spec, spec2, reduce=1 if a.reduce == "tree" else 0, root=0)
        eng.free_table(tab)
        tabs = [None] * world
        dist.all_gather_object(tabs, dict(key_hi=th.key_hi, key_lo=th.key_lo, exts=th.exts, count=th.count, set_off=th.set_off, set_val=th.set_val))
        if rank == 0:
            import oracle_lib as O
            from graph_canon import graphs_equal, canonical_nodes
            # label lists -> class ids in the job-wide numbering (ranks of the sorted distinct lists), vectorised: labels < 64, so a
            # list is a 64-bit mask, and only the few DISTINCT masks are ever turned into tuples
            def masks(t):
                n, off = len(t["key_lo"]), t["set_off"].astype(np.int64)
                m = np.zeros(n, np.uint64)
                np.bitwise_or.at(m, np.repeat(np.arange(n), np.diff(off)), np.uint64(1) << t["set_val"].astype(np.uint64))
                return m
            glob = None
            if kind:
                ms = [masks(t) for t in tabs]
                distinct = np.unique(np.concatenate(ms))
                as_tuple = lambda m: tuple(b for b in range(64) if (int(m) >> b) & 1)
                glob = sorted(as_tuple(m) for m in distinct)
                gid = {m: glob.index(as_tuple(m)) for m in distinct.tolist()}
                lut = np.array([gid[m] for m in distinct.tolist()], dtype=np.uint32)
            shard = []
            for i, t in enumerate(tabs):
                data = lut[np.searchsorted(distinct, ms[i])] if kind else t["count"]
                shard.append(O.compress_kmers(k, False, O.SPEC_SCMAP_EQ if kind else O.SPEC_SAT_ADD, t["key_hi"], t["key_lo"], t["exts"], data))
            if a.check_table:
                hs_all = dbg.synth_reads_host(n_reads=per * world, read_len=150, genome_len=a.reads * 150 // 30, error_rate=0.001,
                                              stranded=False, n_colours=5)
                so = O.SeqSet(hs_all.words, hs_all.start, hs_all.length, None, hs_all.data if kind else None, 1 if kind else 0)
                wt = O.filter_kmers(so, k, O.COUNT_FILTER_SET if kind else O.COUNT_FILTER, 2, stranded=False)
                # every k-mer lives on exactly one rank: the tables, merged by key, are the oracle's table
                hi = np.concatenate([t["key_hi"] for t in tabs]); lo = np.concatenate([t["key_lo"] for t in tabs])
                order = np.lexsort((lo, hi))
                assert len(order) == wt.n and np.array_equal(hi[order], wt.key_hi) and np.array_equal(lo[order], wt.key_lo)
                assert np.array_equal(np.concatenate([t["exts"] for t in tabs])[order], wt.exts)
                if kind:
                    wm = masks(dict(key_lo=wt.key_lo, set_off=wt.set_off, set_val=wt.set_val))
                    assert np.array_equal(np.concatenate(ms)[order], wm)
                else:
                    assert np.array_equal(np.concatenate([t["count"] for t in tabs])[order], wt.count)
            want = O.graph_combine(shard).finish().compress_graph(False, O.SPEC_SCMAP_EQ if kind else O.SPEC_MAX)
            assert final is not None
            if a.reduce == "gather":
                assert graphs_equal(final.arrays(), want.arrays()), "sharded compress differs from the oracle (kind %d)" % kind
            else:
                assert len(final) == len(want.arrays()["start"]), (len(final), len(want.arrays()["start"]))
                assert canonical_nodes(final.arrays(), k, False) == canonical_nodes(want.arrays(), k, False), "tree merge: other unitigs than the oracle (kind %d)" % kind
            assert (final.classes == glob) if kind else final.classes is None
            assert sum(len(t["key_lo"]) for t in tabs) > 1000 and len(final) > 10
            print("kind %d: %d ranks, %d valid k-mers, %d unitigs after the second stage" % (kind, world, sum(len(t["key_lo"]) for t in tabs), len(final)), flush=True)
        else:
            assert final is None
        dist.barrier()
    if rank == 0:
        print("sharded compress ok", flush=True)
    D.close_transports()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
