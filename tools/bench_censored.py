"""The pipeline real callers run at BASELINE config 3's size: dbg_filter_kmers_dev(CountFilter(2)) -> dbg_remove_censored_exts ->
dbg_compress_kmers_with_hash_dev, three times, with the kernels' HIP-event times (DBG_DEBUG=1 in the environment: the routes' own lines).
    python tools/bench_censored.py [reads] [k]"""
import importlib, ctypes as C, time, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dbg = importlib.import_module("rust-debruijn_amd"); capi = importlib.import_module("rust-debruijn_amd._capi")
ctx = dbg.Context(0); lib = ctx.lib; dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 47
p = dbg.synth_params(n_reads=N, read_len=150, genome_len=N * 150 // 30, error_rate=0.001, stranded=False, n_colours=4, first_read=0)
nw = lib.dbg_synth_words(C.byref(p))
words = torch.empty(nw, dtype=torch.int64, device=dev); start = torch.empty(N, dtype=torch.int64, device=dev)
length = torch.empty(N, dtype=torch.int32, device=dev); colour = torch.empty(N, dtype=torch.uint8, device=dev)
ctx.check(lib.dbg_synth_reads_dev(ctx.h, C.byref(p), words.data_ptr(), start.data_ptr(), length.data_ptr(), colour.data_ptr()))
ss = capi.SeqSet(words.data_ptr(), nw, start.data_ptr(), length.data_ptr(), None, None, 0, N)
fp = capi.FilterParams(K, 0, 0, 2, 0, 4)
t = capi.KmerTable(); ctx.check(lib.dbg_filter_kmers_dev(ctx.h, C.byref(ss), C.byref(fp), C.byref(t)))
ctx.check(lib.dbg_remove_censored_exts(ctx.h, K, 0, C.byref(t), 0))
for it in range(3):
    g = capi.Graph()
    ctx.enable_timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.check(lib.dbg_compress_kmers_with_hash_dev(ctx.h, K, 0, 0, t.n, t.key_hi, t.key_lo, t.exts, None, t.count, C.byref(g)))
    dt = time.perf_counter() - t0
    kt = {x["name"]: round(x["ms"], 2) for x in ctx.timings()}
    print("compress call %d: %.1f ms, %d unitigs from %d k-mers; kernels %s (sum %.1f ms)" % (it, dt * 1e3, g.n_nodes, t.n, kt, sum(kt.values())), flush=True)
    lib.dbg_free_graph(ctx.h, C.byref(g))
