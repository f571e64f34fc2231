"""Where the time of the sharded route goes on one GPU: sections of distributed.sharded_filter_kmers timed with the device
synchronised in between.  usage: time_sharded.py [reads] [--pg]   (--pg: 1-rank nccl group, exchange route forced)"""
import ctypes as C, importlib, os, sys, time, faulthandler
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
faulthandler.dump_traceback_later(100, exit=True)
dbg = importlib.import_module("rust-debruijn_amd"); capi = importlib.import_module("rust-debruijn_amd._capi")
D = importlib.import_module("rust-debruijn_amd.distributed")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
pg = "--pg" in sys.argv
ctx = dbg.Context(0); lib = ctx.lib; dev = torch.device("cuda", 0); torch.cuda.set_device(0)
if pg:
    import torch.distributed as dist
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29777", rank=0, world_size=1, device_id=dev)
p = dbg.synth_params(n_reads=n, read_len=150, genome_len=n * 5, error_rate=0.001, stranded=False, n_colours=4)
nw = lib.dbg_synth_words(C.byref(p))
words = torch.empty(nw, dtype=torch.int64, device=dev); start = torch.empty(n, dtype=torch.int64, device=dev)
length = torch.empty(n, dtype=torch.int32, device=dev); colour = torch.empty(n, dtype=torch.uint8, device=dev)
ctx.check(lib.dbg_synth_reads_dev(ctx.h, C.byref(p), words.data_ptr(), start.data_ptr(), length.data_ptr(), colour.data_ptr()))
ss = capi.SeqSet(words.data_ptr(), nw, start.data_ptr(), length.data_ptr(), None, colour.data_ptr(), 1, n)
eng = D.HipEngine(ctx, dev)
T = {}
_orig = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); T[name] = T.get(name, 0) + time.perf_counter() - t0
        return r
    setattr(obj, name, g)
for nm in ("count_instances", "plan", "scan", "scatter", "count_begin", "count_bins", "count_finish"):
    wrap(eng, nm)
for nm in ("send_layout", "exchange_geometry", "_all_to_all"):
    wrap(D, nm)
for rep in range(3):
    T.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = {}
    tab, total, n_local, n_recs = D.sharded_filter_kmers(eng, ss, 47, False, 1, 2, stats=st, force_exchange=pg)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    eng.free_table(tab)
    print("rep", rep, "total %.1f ms" % (dt * 1e3), {k: round(v * 1e3, 1) for k, v in T.items()}, "other %.1f" % ((dt - sum(T.values())) * 1e3), st, flush=True)
