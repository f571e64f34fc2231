"""First-call cost of dbg_compress_kmers_with_hash_dev in a fresh ctx against later calls (config 3: 5e8 k-mers -> 1.9e7 unitigs):
where the 2 s of the first call go (pinned result blocks? pooled device blocks?)."""
import importlib, ctypes as C, time, torch, sys, os
sys.path.insert(0, os.getcwd())
dbg = importlib.import_module("rust-debruijn_amd"); capi = importlib.import_module("rust-debruijn_amd._capi")
ctx = dbg.Context(0); lib = ctx.lib; dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
p = dbg.synth_params(n_reads=N, read_len=150, genome_len=N * 150 // 30, error_rate=0.001, stranded=False, n_colours=4, first_read=0)
nw = lib.dbg_synth_words(C.byref(p))
words = torch.empty(nw, dtype=torch.int64, device=dev); start = torch.empty(N, dtype=torch.int64, device=dev)
length = torch.empty(N, dtype=torch.int32, device=dev); colour = torch.empty(N, dtype=torch.uint8, device=dev)
ctx.check(lib.dbg_synth_reads_dev(ctx.h, C.byref(p), words.data_ptr(), start.data_ptr(), length.data_ptr(), colour.data_ptr()))
ss = capi.SeqSet(words.data_ptr(), nw, start.data_ptr(), length.data_ptr(), None, None, 0, N)
fp = capi.FilterParams(47, 0, 0, 2, 0, 4)
t = capi.KmerTable(); ctx.check(lib.dbg_filter_kmers_dev(ctx.h, C.byref(ss), C.byref(fp), C.byref(t)))
ctx2 = dbg.Context(0)                      # a fresh ctx: nothing pooled, nothing pinned
for it in range(3):
    g = capi.Graph()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx2.check(lib.dbg_compress_kmers_with_hash_dev(ctx2.h, 47, 0, 0, t.n, t.key_hi, t.key_lo, t.exts, None, t.count, C.byref(g)))
    dt = time.perf_counter() - t0
    print("compress call %d in a fresh ctx: %.1f ms (%d unitigs)" % (it, dt * 1e3, g.n_nodes), flush=True)
    lib.dbg_free_graph(ctx2.h, C.byref(g))
# what pinning costs by itself
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
for mb in (64, 680):
    ptr = ctypes.c_void_p()
    t0 = time.perf_counter(); rc = hip.hipHostMalloc(ctypes.byref(ptr), ctypes.c_size_t(mb << 20), 0); dt = time.perf_counter() - t0
    print("hipHostMalloc(%d MB): rc %d, %.1f ms" % (mb, rc, dt * 1e3), flush=True)
    hip.hipHostFree(ptr)
for gb in (1, 8):
    ptr = ctypes.c_void_p()
    t0 = time.perf_counter(); rc = hip.hipMalloc(ctypes.byref(ptr), ctypes.c_size_t(gb << 30)); dt = time.perf_counter() - t0
    print("hipMalloc(%d GB): rc %d, %.1f ms" % (gb, rc, dt * 1e3), flush=True)
    hip.hipFree(ptr)
