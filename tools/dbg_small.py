import importlib, sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
dbg = importlib.import_module("rust-debruijn_amd")
ctx = dbg.Context(0)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 47
hs = dbg.synth_reads_host(n_reads=2000, read_len=150, error_rate=0.001, stranded=False, n_colours=4)
print("calling filter", flush=True)
t, _ = dbg.filter_kmers(hs, dbg.CountFilter(2), False, False, 4, k=k, ctx=ctx)
print("n", len(t), flush=True)
