"""The reference's own boundary (host arrays in, host arrays out; filter.rs:139-148) at BASELINE configs[1] size:
dbg_filter_kmers with CountFilter / CountFilterSet on pageable numpy arrays -> host table.  Prints Gkmer/s, bytes moved,
and the PCIe rate reached.  usage: bench_host_boundary.py [reads] [count|set] [reps]"""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dbg = importlib.import_module("rust-debruijn_amd")
capi = importlib.import_module("rust-debruijn_amd._capi")


def run(ctx, n, kind, k=47, reps=3):
    import torch
    lib = ctx.lib
    dev = torch.device("cuda", 0)
    p = dbg.synth_params(n_reads=n, read_len=150, genome_len=n * 5, error_rate=0.001, stranded=False, n_colours=4)
    nw = lib.dbg_synth_words(C.byref(p))
    words = torch.empty(nw, dtype=torch.int64, device=dev); start = torch.empty(n, dtype=torch.int64, device=dev)
    length = torch.empty(n, dtype=torch.int32, device=dev); colour = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.check(lib.dbg_synth_reads_dev(ctx.h, C.byref(p), words.data_ptr(), start.data_ptr(), length.data_ptr(), colour.data_ptr()))
    hw, hst, hl, hc = words.cpu().numpy(), start.cpu().numpy(), length.cpu().numpy(), colour.cpu().numpy()    # pageable host arrays
    del words, start, length, colour
    torch.cuda.empty_cache()
    is_set = kind == "set"
    ss = capi.SeqSet(hw.ctypes.data, nw, hst.ctypes.data, hl.ctypes.data, None, hc.ctypes.data if is_set else None, 1 if is_set else 0, n)
    fp = capi.FilterParams(k, 0, 1 if is_set else 0, 2, 0, 4)
    out = []
    for rep in range(reps):
        t = capi.KmerTable()
        t0 = time.perf_counter()
        ctx.check(lib.dbg_filter_kmers(ctx.h, C.byref(ss), C.byref(fp), C.byref(t)))
        dt = time.perf_counter() - t0
        b_in = hw.nbytes + hst.nbytes + hl.nbytes + (hc.nbytes if is_set else 0)
        b_out = t.n * (8 + 8 + 1) + (t.n * 2 if not is_set else (t.n + 1) * 8 + t.n_set_val * 4)
        out.append(dict(seconds=round(dt, 4), gkmer_per_s=round(t.n_kmer_instances / dt / 1e9, 3), valid=int(t.n), gb_in=round(b_in / 1e9, 2),
                        gb_out=round(b_out / 1e9, 2), pcie_gb_per_s=round((b_in + b_out) / dt / 1e9, 1)))
        lib.dbg_free_table(ctx.h, C.byref(t))
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    kind = sys.argv[2] if len(sys.argv) > 2 else "count"
    ctx = dbg.Context(0)
    for r in run(ctx, n, kind, reps=int(sys.argv[3]) if len(sys.argv) > 3 else 3):
        print(kind, n, r, flush=True)
    ctx.close()
