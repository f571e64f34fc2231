"""filter_kmers(CountFilterSet) at BASELINE configs[1] shape with L distinct u32 labels drawn at random per read:
`python tools/bench_labels.py [--reads N] [--k K] [--lists 0|1|auto] L...`  (L > 64: label groups up to 1024, label lists beyond --
or everywhere with --lists 1; fast_manylabels.hpp / fast_labellists.hpp).  Labels are spread over [0, 2^24)."""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib, ctypes as C, time, torch

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=100_000_000)
ap.add_argument("--k", type=int, default=47)
ap.add_argument("--lists", default="auto")
ap.add_argument("--iters", type=int, default=6, help="calls per alphabet (the first four are the slab tournament of the shape); the last one is reported")
ap.add_argument("--target", default=None, help="DBG_FAST_TARGET: k-mer instances per bin")
ap.add_argument("--small", action="store_true", help="labels below 65536 (label groups apply for 65..1024 of them)")
ap.add_argument("labels", type=int, nargs="+")
a = ap.parse_args()
dbg = importlib.import_module("rust-debruijn_amd"); capi = importlib.import_module("rust-debruijn_amd._capi")
ctx = dbg.Context(0); lib = ctx.lib; dev = torch.device("cuda", 0)
if a.lists != "auto":
    ctx.set_option("DBG_LABEL_LISTS", a.lists)
if a.target:
    ctx.set_option("DBG_FAST_TARGET", a.target)
N = a.reads
ctx.enable_timing(True)
p = dbg.synth_params(n_reads=N, read_len=150, genome_len=N * 150 // 30, error_rate=0.001, stranded=False, n_colours=4, first_read=0)
nw = lib.dbg_synth_words(C.byref(p))
words = torch.empty(nw, dtype=torch.int64, device=dev); start = torch.empty(N, dtype=torch.int64, device=dev)
length = torch.empty(N, dtype=torch.int32, device=dev); colour = torch.empty(N, dtype=torch.uint8, device=dev)
ctx.check(lib.dbg_synth_reads_dev(ctx.h, C.byref(p), words.data_ptr(), start.data_ptr(), length.data_ptr(), colour.data_ptr()))
del colour
g = torch.Generator(device=dev); g.manual_seed(7)
for L in a.labels:
    alphabet = torch.randperm(65536 if a.small else 1 << 24, device=dev, generator=g)[:L].to(torch.int32)
    lab = alphabet[torch.randint(0, L, (N,), device=dev, generator=g)].contiguous()
    ss = capi.SeqSet(words.data_ptr(), nw, start.data_ptr(), length.data_ptr(), None, lab.data_ptr(), 4, N)
    fp = capi.FilterParams(a.k, 0, 1, 2, 0, 4)
    for it in range(a.iters):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        t = capi.KmerTable(); ctx.check(lib.dbg_filter_kmers_dev(ctx.h, C.byref(ss), C.byref(fp), C.byref(t)))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        n, nsv = t.n, t.n_set_val
        kt = {}
        for x in ctx.timings():
            kt[x["name"]] = round(kt.get(x["name"], 0) + x["ms"], 1)
        lib.dbg_free_table(ctx.h, C.byref(t))
    nk = N * (150 - a.k + 1)
    print("labels", L, "lists", a.lists, "k", a.k, "reads", N, "valid", n, "set_val", nsv, "ms %.1f" % (dt * 1e3), "Gkmer/s %.2f" % (nk / dt / 1e9), kt, flush=True)
    del lab, alphabet
