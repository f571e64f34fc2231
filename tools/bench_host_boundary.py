"""filter_kmers at the reference's own boundary: reads in host memory in, table in host memory out (PCIe included).
The reads are generated on the device and copied to ordinary (pageable) numpy arrays first."""
import sys, importlib, ctypes as C, time
sys.path.insert(0, ".")
import numpy as np, torch
dbg = importlib.import_module("rust-debruijn_amd"); capi = importlib.import_module("rust-debruijn_amd._capi")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 47
ctx = dbg.Context(0); lib = ctx.lib; dev = torch.device("cuda", 0)
p = dbg.synth_params(n_reads=n, read_len=150, genome_len=n * 5, error_rate=0.001, stranded=False, n_colours=4)
nw = lib.dbg_synth_words(C.byref(p))
words = torch.empty(nw, dtype=torch.int64, device=dev); start = torch.empty(n, dtype=torch.int64, device=dev)
length = torch.empty(n, dtype=torch.int32, device=dev); colour = torch.empty(n, dtype=torch.uint8, device=dev)
ctx.check(lib.dbg_synth_reads_dev(ctx.h, C.byref(p), words.data_ptr(), start.data_ptr(), length.data_ptr(), colour.data_ptr()))
hs = dbg.HostSeqs(words.cpu().numpy().view(np.uint64), start.cpu().numpy().view(np.uint64), length.cpu().numpy().view(np.uint32), None,
                  colour.cpu().numpy(), 1)
del words, start, length, colour
torch.cuda.empty_cache()
inst = n * (150 - k + 1)
for rep in range(2):
    t0 = time.perf_counter()
    tab, _ = dbg.filter_kmers(hs, dbg.CountFilterSet(2), False, False, 4, k=k, ctx=ctx)
    dt = time.perf_counter() - t0
    out_b = tab.key_lo.nbytes + (tab.key_hi.nbytes if tab.key_hi is not None else 0) + tab.exts.nbytes
    print("host boundary rep %d: %.3f s = %.2f Gkmer/s (%d valid k-mers; in %.2f GB, out >= %.2f GB)" %
          (rep, dt, inst / dt / 1e9, len(tab), (hs.words.nbytes + hs.start.nbytes + hs.length.nbytes + n) / 1e9, out_b / 1e9), flush=True)
    del tab
