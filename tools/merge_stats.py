"""Sender-side duplicate merge (dbg_shard_plan.merge_dups): records a rank sends with and without it, against the number of
distinct records (the best any merge can do), and the kernel's time.  usage: python tools/merge_stats.py [reads_per_rank] [world ...]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from pkg import dbg  # noqa: E402

D = importlib.import_module("rust-debruijn_amd.distributed")


def main():
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    worlds = [int(x) for x in sys.argv[2:]] or [2, 8]
    ctx = dbg.Context(0)
    ctx.enable_timing(True)
    eng = D.HipEngine(ctx, torch.device("cuda", 0))
    for k in (47, 63):
        for world in worlds:
            n_reads = per * world                      # rank 0's share of a `world`-rank job over 30x reads
            p = dbg.synth_params(n_reads=per, read_len=150, genome_len=n_reads * 150 // 30, error_rate=0.001, stranded=False,
                                 n_colours=4, first_read=0)
            nw = ctx.lib.dbg_synth_words(C.byref(p))
            dev = eng.device
            words = torch.empty(nw, dtype=torch.int64, device=dev)
            start = torch.empty(per, dtype=torch.int64, device=dev)
            length = torch.empty(per, dtype=torch.int32, device=dev)
            data = torch.empty(per, dtype=torch.uint8, device=dev)
            ctx.check(ctx.lib.dbg_synth_reads_dev(ctx.h, C.byref(p), words.data_ptr(), start.data_ptr(), length.data_ptr(), data.data_ptr()))
            ss = eng.seqset(words, start, length, None, 0)
            total = eng.count_instances(ss, k) * world
            out = {}
            for merge in (False, True):
                plan = eng.plan(k, False, 0, 2, total, 0, merge_dups=merge)
                bin_off, n = eng.scan(ss, plan)
                t = {e["name"]: e["ms"] for e in ctx.timings()}
                recs = eng.scatter(plan, bin_off, n)
                out[merge] = (n, t.get("slab_merge", 0.0), t.get("sk_scan", 0.0))
                if not merge:
                    r = recs.view(-1, plan.rec_words)
                    distinct = int(torch.unique(r, dim=0).shape[0]) if r.shape[0] < 60_000_000 else -1
                del recs
            n0, n1 = out[False][0], out[True][0]
            print(f"k={k} world={world} reads/rank={per}: records {n0} -> {n1} ({n1 / n0:.3f}); distinct {distinct} ({distinct / n0:.3f}); "
                  f"slab_merge {out[True][1]:.2f} ms (scan {out[True][2]:.2f} ms)", flush=True)
            del words, start, length, data
            ctx.trim()


if __name__ == "__main__":
    main()
