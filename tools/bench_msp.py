#!/usr/bin/env python3
"""Rate of the reference-exact MSP scanner (dbg_msp_sequence_dev, msp.rs:207-324) on device-resident reads."""
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

dbg = importlib.import_module("rust-debruijn_amd")
capi = importlib.import_module("rust-debruijn_amd._capi")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
k, p_, L = 47, 8, 150
lmer_words = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = dbg.Context(0)
lib = ctx.lib
dev = torch.device("cuda", 0)
sp = dbg.synth_params(n_reads=n, read_len=L, genome_len=n * L // 30, error_rate=0.001, stranded=False, n_colours=0)
nw = lib.dbg_synth_words(C.byref(sp))
words = torch.empty(nw, dtype=torch.int64, device=dev)
start = torch.empty(n, dtype=torch.int64, device=dev)
length = torch.empty(n, dtype=torch.int32, device=dev)
ctx.check(lib.dbg_synth_reads_dev(ctx.h, C.byref(sp), words.data_ptr(), start.data_ptr(), length.data_ptr(), None))
ss = capi.SeqSet(words.data_ptr(), nw, start.data_ptr(), length.data_ptr(), None, None, 0, n)
mp = capi.MspParams(k, p_, None, 1, lmer_words)
res = {}
for rep in range(3):
    pc = capi.MspPieces()
    ctx.enable_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.check(lib.dbg_msp_sequence_dev(ctx.h, C.byref(ss), C.byref(mp), C.byref(pc)))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kt = {t["name"]: round(t["ms"], 3) for t in ctx.timings()}
    ctx.enable_timing(False)
    res = dict(reads=n, k=k, p=p_, lmer_words=lmer_words, pieces=int(pc.n_pieces), seconds=round(dt, 4),
               reads_per_s=round(n / dt, 1), gbases_per_s=round(n * L / dt / 1e9, 3), kernels_ms=kt)
    lib.dbg_free_pieces(ctx.h, C.byref(pc))
print(json.dumps(res))
