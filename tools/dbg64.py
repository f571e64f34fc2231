import os, sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import oracle_lib as O
from pkg import dbg
from test_gpu_filter import random_reads, to_host_seqs
os.environ["DBG_PATH"] = "fast"; os.environ["DBG_DEBUG"] = "1"
ctx = dbg.Context(0)
for k in (63, 64):
    for n in (5, 50, 600):
        rng = np.random.default_rng(1000 + 64 * 3)
        seqs = random_reads(rng, n, 4000, 150, False)
        ss = O.SeqSet.from_byte_seqs(seqs)
        t0 = time.time()
        print("k", k, "n", n, flush=True)
        got, _ = dbg.filter_kmers(to_host_seqs(ss), dbg.CountFilter(1), False, False, 4, k=k, ctx=ctx)
        print("  ok", len(got), time.time() - t0, flush=True)
