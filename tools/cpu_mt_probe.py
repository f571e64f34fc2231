"""scratch: thread scaling of the many-core CPU baseline on the GPU box's host"""
import sys, os, time, importlib
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, oracle_lib as O
dbg = importlib.import_module("rust-debruijn_amd")
print("cpus", os.cpu_count(), len(os.sched_getaffinity(0)))
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cpu.max", e)
n_s, L, k = 2000000, 150, 47
hs = dbg.synth_reads_host(n_reads=n_s, read_len=L, genome_len=n_s * L // 30, error_rate=0.001, stranded=False, n_colours=4)
so = O.SeqSet(hs.words, hs.start, hs.length, None, hs.data, 1)
for nt in (8, 32, 64, 128, 256):
    for ns in (4 * nt, 16 * nt):
        t, nv = O.time_filter_kmers_sharded_mt(so, k, O.COUNT_FILTER_SET, 2, False, nt, ns)
        print(nt, ns, "%.2f s (msp %.2f)" % (t, O._PH1.value), nv, "%.4f Gkmer/s" % (n_s * (L - k + 1) / t / 1e9), flush=True)
