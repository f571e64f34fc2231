#!/usr/bin/env python3
"""Streaming rate of the ASCII <-> 2-bit kernels (dbg_pack_acgt_dev / dbg_unpack_acgt_dev) on device-resident data.
Algorithmic bytes: pack = 1 B in + 0.25 B out per base; unpack = 0.25 B in + 1 B out."""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 15_000_000_000      # configs[1]: 1e8 reads x 150 bp
ctx = dbg.Context(0)
lib = ctx.lib
dev = torch.device("cuda", 0)
lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
ascii_ = torch.empty(n, dtype=torch.uint8, device=dev)
step = 1 << 30
for o in range(0, n, step):
    m = min(step, n - o)
    ascii_[o:o + m] = lut[torch.randint(0, 4, (m,), device=dev)]
words = torch.empty((n + 31) // 32, dtype=torch.int64, device=dev)
back = torch.empty(n, dtype=torch.uint8, device=dev)
res = {}
for name, fn in (("pack_acgt", lambda: lib.dbg_pack_acgt_dev(ctx.h, ascii_.data_ptr(), n, words.data_ptr(), None)),
                 ("unpack_acgt", lambda: lib.dbg_unpack_acgt_dev(ctx.h, words.data_ptr(), 0, n, back.data_ptr()))):
    ctx.check(fn())
    torch.cuda.synchronize()
    ctx.enable_timing(True)
    t0 = time.perf_counter()
    reps = 5
    ms = 0.0
    for _ in range(reps):
        ctx.check(fn())
        torch.cuda.synchronize()
        ms = sum(t["ms"] for t in ctx.timings())          # cumulative since enable_timing
    dt = (time.perf_counter() - t0) / reps
    ms /= reps
    ctx.enable_timing(False)
    res[name] = dict(bases=n, wall_ms=round(dt * 1e3, 3), kernel_ms=round(ms, 3), gbases_per_s=round(n / (ms * 1e-3) / 1e9, 1),
                     alg_GBps=round(1.25 * n / (ms * 1e-3) / 1e9, 1), frac_of_8TBps=round(1.25 * n / (ms * 1e-3) / 8e12, 3))
ok = True
for o in range(0, n, step):
    ok = ok and bool((back[o:o + step] == ascii_[o:o + step]).all().item())
assert ok
print(json.dumps(res))
