"""Mid-size differential check (between the unit tests' thousands of reads and the full-size property suite): filter_kmers on
1.5e5 .. 2e6 synthetic reads, the whole table compared with the CPU oracle (15 s of CPU per 1e6 reads).  usage: python tools/hunt_midsize.py [n_cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from pkg import dbg

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = dbg.Context(0)
bad = 0
for i in range(n_cases):
    n = int(rng.choice([150_000, 400_000, 1_000_000, 2_000_000]))
    k = int(rng.choice([20, 31, 33, 47, 48, 49, 55, 63, 64, 13]))
    kind = int(rng.integers(0, 2))
    colours = int(rng.choice([4, 24, 40, 100, 200, 3000, 50000])) if kind else 4      # > 64: label lists (fast_labellists.hpp)
    cov = int(rng.choice([3, 30, 100]))
    err = float(rng.choice([0.0, 0.001, 0.01]))
    stranded = bool(rng.integers(0, 2))
    min_obs = int(rng.choice([1, 2, 3]))
    hs = dbg.synth_reads_host(n_reads=n, read_len=150, genome_len=n * 150 // cov, error_rate=err, stranded=stranded, n_colours=min(colours, 255))
    lab, width = hs.data, 1
    if colours > 255:                                                    # u32 labels spread over [0, 2^24), one per read at random
        alphabet = np.unique(rng.integers(0, 1 << 24, size=colours, dtype=np.uint64)).astype(np.uint32)
        lab, width = alphabet[rng.integers(0, len(alphabet), size=n)], 4
        hs = dbg.HostSeqs(hs.words, hs.start, hs.length, None, lab, 4)
    t0 = time.time()
    summ = (dbg.CountFilterSet if kind else dbg.CountFilter)(min_obs)
    got, _ = dbg.filter_kmers(hs if kind else dbg.HostSeqs(hs.words, hs.start, hs.length), summ, stranded, False, 4, k=k, ctx=ctx)
    t1 = time.time()
    want = O.filter_kmers(O.SeqSet(hs.words, hs.start, hs.length, None, lab if kind else None, width if kind else 0), k, kind, min_obs, stranded=stranded)
    t2 = time.time()
    ok = (len(got) == want.n and np.array_equal(got.key_hi, want.key_hi) and np.array_equal(got.key_lo, want.key_lo) and np.array_equal(got.exts, want.exts)
          and (np.array_equal(got.count, want.count) if not kind else (np.array_equal(got.set_off, want.set_off) and np.array_equal(got.set_val, want.set_val))))
    bad += not ok
    print("case %d reads=%d k=%d kind=%d colours=%d cov=%d err=%g stranded=%d min_obs=%d valid=%d gpu %.2fs cpu %.1fs %s"
          % (i, n, k, kind, colours, cov, err, stranded, min_obs, want.n, t1 - t0, t2 - t1, "ok" if ok else "MISMATCH"), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
