"""times the super-k-mer scan alone on the default workload (DBG_LIB selects an experimental build of the library)"""
import sys, os, importlib, ctypes as C, json
sys.path.insert(0, ".")
import torch
dbg = importlib.import_module("rust-debruijn_amd")
capi = importlib.import_module("rust-debruijn_amd._capi")
D = importlib.import_module("rust-debruijn_amd.distributed")
lib = capi.load(); ctx = dbg.Context(0); dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 47
p = dbg.synth_params(n_reads=n, read_len=150, genome_len=n * 5, error_rate=0.001, stranded=False, n_colours=4)
nw = lib.dbg_synth_words(C.byref(p))
words = torch.empty(nw, dtype=torch.int64, device=dev); start = torch.empty(n, dtype=torch.int64, device=dev)
length = torch.empty(n, dtype=torch.int32, device=dev); colour = torch.empty(n, dtype=torch.uint8, device=dev)
ctx.check(lib.dbg_synth_reads_dev(ctx.h, C.byref(p), words.data_ptr(), start.data_ptr(), length.data_ptr(), colour.data_ptr()))
ss = capi.SeqSet(words.data_ptr(), nw, start.data_ptr(), length.data_ptr(), None, colour.data_ptr(), 1, n)
eng = D.HipEngine(ctx, dev)
tot = eng.count_instances(ss, k)
plan = eng.plan(k, False, 1, 2, tot)
import time
ctx.enable_timing(True)
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bin_off, n_recs = eng.scan(ss, plan)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    kt = {x["name"]: round(x["ms"], 2) for x in ctx.timings()}
    print(os.path.basename(os.environ.get("DBG_LIB", "default")), "rep", rep, "scan+compact %.2f ms" % (dt * 1e3), "kernels", kt, "records", n_recs, flush=True)
