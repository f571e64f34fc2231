"""Times dbg_compress_kmers_with_hash (host arrays in, host BaseGraph out) on the valid-k-mer table of a
synthetic read set: filter (GPU, CountFilter(2)) -> table to host -> compress."""
import ctypes as C, importlib, sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
dbg = importlib.import_module("rust-debruijn_amd"); capi = importlib.import_module("rust-debruijn_amd._capi")
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
err = float(os.environ.get("DBG_BENCH_ERR", 0.001))          # 0: no branch points, the genome is one chain
k = 47
ctx = dbg.Context(0); lib = ctx.lib; dev = torch.device("cuda", 0)
p = dbg.synth_params(n_reads=n_reads, read_len=150, genome_len=n_reads * 150 // 30, error_rate=err, stranded=False, n_colours=0)
nw = lib.dbg_synth_words(C.byref(p))
words = torch.empty(nw, dtype=torch.int64, device=dev); start = torch.empty(n_reads, dtype=torch.int64, device=dev)
length = torch.empty(n_reads, dtype=torch.int32, device=dev)
ctx.check(lib.dbg_synth_reads_dev(ctx.h, C.byref(p), words.data_ptr(), start.data_ptr(), length.data_ptr(), None))
ss = capi.SeqSet(words.data_ptr(), nw, start.data_ptr(), length.data_ptr(), None, None, 0, n_reads)
fp = capi.FilterParams(k, 0, 0, 2, 0, 4)
t = capi.KmerTable(); ctx.check(lib.dbg_filter_kmers_dev(ctx.h, C.byref(ss), C.byref(fp), C.byref(t)))
for rep in range(2):                     # device-resident index
    ctx.enable_timing(True)
    g = capi.Graph(); torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.check(lib.dbg_compress_kmers_with_hash_dev(ctx.h, k, 0, 0, t.n, t.key_hi, t.key_lo, t.exts, None, t.count, C.byref(g)))
    dt = time.perf_counter() - t0
    print("dev-index nodes", g.n_nodes, "time %.3f s" % dt, "unitigs/s %.3e" % (g.n_nodes / dt), "kmers/s %.3e" % (t.n / dt),
          {x["name"]: round(x["ms"], 1) for x in ctx.timings()}, flush=True)
    lib.dbg_free_graph(ctx.h, C.byref(g))
h = capi.KmerTable(); ctx.check(lib.dbg_table_to_host(ctx.h, C.byref(t), C.byref(h))); lib.dbg_free_table(ctx.h, C.byref(t))
n = h.n
print("valid kmers", n, flush=True)
data = np.ctypeslib.as_array(C.cast(h.count, C.POINTER(C.c_uint16)), shape=(n,)).astype(np.uint32)
for mode in (sys.argv[2:] or ["device"]):
    if mode == "none":
        break
    ctx.set_option("DBG_COMPRESS", mode)
    ctx.enable_timing(True)
    g = capi.Graph()
    t0 = time.perf_counter()
    ctx.check(lib.dbg_compress_kmers_with_hash(ctx.h, k, 0, 0, n, h.key_hi, h.key_lo, h.exts, data.ctypes.data_as(C.c_void_p), None, C.byref(g)))
    dt = time.perf_counter() - t0
    print(mode, "nodes", g.n_nodes, "bases", g.seq_len_bases, "time %.3f s" % dt, "unitigs/s %.3e" % (g.n_nodes / dt), "kmers/s %.3e" % (n / dt),
          {x["name"]: round(x["ms"], 1) for x in ctx.timings()}, flush=True)
    lib.dbg_free_graph(ctx.h, C.byref(g))
