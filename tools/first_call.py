import importlib, ctypes as C, time, torch, sys, os
sys.path.insert(0, os.getcwd())
dbg = importlib.import_module("rust-debruijn_amd"); capi = importlib.import_module("rust-debruijn_amd._capi")
ctx = dbg.Context(0); lib = ctx.lib; dev = torch.device("cuda", 0)
N = 100_000_000
p = dbg.synth_params(n_reads=N, read_len=150, genome_len=N * 150 // 30, error_rate=0.001, stranded=False, n_colours=4, first_read=0)
nw = lib.dbg_synth_words(C.byref(p))
words = torch.empty(nw, dtype=torch.int64, device=dev); start = torch.empty(N, dtype=torch.int64, device=dev)
length = torch.empty(N, dtype=torch.int32, device=dev); colour = torch.empty(N, dtype=torch.uint8, device=dev)
ctx.check(lib.dbg_synth_reads_dev(ctx.h, C.byref(p), words.data_ptr(), start.data_ptr(), length.data_ptr(), colour.data_ptr()))
ss = capi.SeqSet(words.data_ptr(), nw, start.data_ptr(), length.data_ptr(), None, colour.data_ptr(), 1, N)
fp = capi.FilterParams(47, 0, 1, 2, 0, 4)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t = capi.KmerTable(); ctx.check(lib.dbg_filter_kmers_dev(ctx.h, C.byref(ss), C.byref(fp), C.byref(t)))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    lib.dbg_free_table(ctx.h, C.byref(t))
    print("call %d: %.1f ms" % (it, dt * 1e3), flush=True)
