"""filter_kmers(CountFilterSet) at BASELINE configs[1] size with C colours per read set: `python tools/bench_colours.py 4 24 40 64`
(<= 24 colours: one mask word next to the Exts; 25..64: the WIDE layout of the counting kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib, ctypes as C, time, torch, numpy as np, sys
dbg = importlib.import_module("rust-debruijn_amd"); capi = importlib.import_module("rust-debruijn_amd._capi")
ctx = dbg.Context(0); lib = ctx.lib; dev = torch.device("cuda", 0)
N = 100_000_000
ctx.enable_timing(True)
for ncol in [int(x) for x in sys.argv[1:]]:
    p = dbg.synth_params(n_reads=N, read_len=150, genome_len=N * 150 // 30, error_rate=0.001, stranded=False, n_colours=ncol, first_read=0)
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
        n, nsv = t.n, t.n_set_val
        kt = {x["name"]: round(x["ms"], 1) for x in ctx.timings()}
        lib.dbg_free_table(ctx.h, C.byref(t))
    print("colours", ncol, "valid", n, "set_val", nsv, "ms %.1f" % (dt * 1e3), "Gkmer/s %.1f" % (N * 104 / dt / 1e9), kt, flush=True)
    del words, start, length, colour
