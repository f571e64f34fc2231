"""BASELINE config 5's per-GPU share on one GPU: 7.5*10^7 reads x 150 bp, k = 51, CountFilterSet<u8> (4 colours) -> label-list classes ->
compress_kmers_with_hash with ScmapCompress, the index kept in HBM (dbg_filter_kmers_dev -> dbg_compress_table_dev).  Prints one JSON line."""
import ctypes as C, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
dbg = importlib.import_module("rust-debruijn_amd"); capi = importlib.import_module("rust-debruijn_amd._capi")
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 75_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 51
ctx = dbg.Context(0); lib = ctx.lib; dev = torch.device("cuda", 0)
p = dbg.synth_params(n_reads=n_reads, read_len=150, genome_len=n_reads * 150 // 30, error_rate=0.001, stranded=False, n_colours=4)
nw = lib.dbg_synth_words(C.byref(p))
words = torch.empty(nw, dtype=torch.int64, device=dev); start = torch.empty(n_reads, dtype=torch.int64, device=dev)
length = torch.empty(n_reads, dtype=torch.int32, device=dev); colour = torch.empty(n_reads, dtype=torch.uint8, device=dev)
ctx.check(lib.dbg_synth_reads_dev(ctx.h, C.byref(p), words.data_ptr(), start.data_ptr(), length.data_ptr(), colour.data_ptr()))
ss = capi.SeqSet(words.data_ptr(), nw, start.data_ptr(), length.data_ptr(), None, colour.data_ptr(), 1, n_reads)
fp = capi.FilterParams(k, 0, 1, 2, 0, 4)
out = {}
for rep in range(2):
    t = capi.KmerTable(); torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.check(lib.dbg_filter_kmers_dev(ctx.h, C.byref(ss), C.byref(fp), C.byref(t)))
    tf = time.perf_counter() - t0
    ctx.enable_timing(True)
    g = capi.Graph(); cl = capi.LabelClasses(); t0 = time.perf_counter()
    ctx.check(lib.dbg_compress_table_dev(ctx.h, k, 0, 3, C.byref(t), C.byref(g), C.byref(cl)))
    tc = time.perf_counter() - t0
    out = dict(reads=n_reads, k=k, kmer_instances=int(t.n_kmer_instances), valid_kmers=int(t.n), label_classes=int(cl.n_classes), unitigs=int(g.n_nodes),
               filter_s=round(tf, 4), filter_gkmer_per_s=round(t.n_kmer_instances / tf / 1e9, 2), compress_s=round(tc, 4),
               unitigs_per_s=round(g.n_nodes / tc, 1), compress_phases_ms={x["name"]: round(x["ms"], 1) for x in ctx.timings()})
    ctx.enable_timing(False)
    lib.dbg_free_graph(ctx.h, C.byref(g)); lib.dbg_free_label_classes(C.byref(cl)); lib.dbg_free_table(ctx.h, C.byref(t))
print(json.dumps(out))
