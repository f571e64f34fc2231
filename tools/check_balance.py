"""Run under torch.distributed.run (N ranks; --one-device --backend gloo lets them share one GPU): ownership of the bin space on a
LOW-COMPLEXITY stream.  5 % of the reads are poly-A / dinucleotide / trinucleotide repeats (with a sprinkle of substitutions),
whose k-mers all share a handful of minimizers: with equal bin ranges the owners of those bins receive far more records than the
others; with ownership cut from the all-reduced record histogram of the scan (dbg_shard_params.balance = 1, the default;
dbg_shard_owner_bounds) every rank receives the same number to within one bin.  Both runs must give the table of the single call
over all reads (order-independent digest), so the check is also a parity check of the histogram-driven layout.

Prints one JSON line on rank 0: records_owned max / mean for both settings, and the digests."""
import argparse
import ctypes as C
import importlib
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_reads(n, seed, genome, low_frac=0.05):
    """[n, 150] bases: 30x-style reads of `genome` (both strands, 0.1 % substitutions) with low_frac low-complexity reads mixed in"""
    rng = np.random.default_rng(seed)
    L = 150
    pos = rng.integers(0, len(genome) - L, n)
    reads = genome[pos[:, None] + np.arange(L)[None, :]]
    flip = rng.random(n) < 0.5
    reads[flip] = 3 - reads[flip][:, ::-1]
    low = rng.random(n) < low_frac
    kind = rng.integers(0, 4, n)
    units = [np.array([0]), np.array([0, 1]), np.array([3, 2]), np.array([0, 2, 3])]
    for u_i, u in enumerate(units):
        sel = low & (kind == u_i)
        phase = rng.integers(0, len(u), int(sel.sum()))
        reads[sel] = u[(phase[:, None] + np.arange(L)[None, :]) % len(u)]
    err = rng.random((n, L)) < 0.001
    reads[err] = (reads[err] + rng.integers(1, 4, int(err.sum()))) % 4
    return reads.astype(np.uint8), int(low.sum())


def pack(reads):
    n, L = reads.shape
    flat = reads.reshape(-1).astype(np.uint64)
    pad = (-len(flat)) % 32
    flat = np.concatenate([flat, np.zeros(pad + 64, np.uint64)]).reshape(-1, 32)
    shifts = (62 - 2 * np.arange(32)).astype(np.uint64)
    words = (flat << shifts[None, :]).sum(axis=1, dtype=np.uint64)
    start = (np.arange(n, dtype=np.uint64) * np.uint64(L))
    length = np.full(n, L, np.uint32)
    return words, start, length


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--one-device", action="store_true")
    ap.add_argument("--reads", type=int, default=400000, help="reads per rank")
    ap.add_argument("--k", type=int, default=47)
    a = ap.parse_args()
    dbg = importlib.import_module("rust-debruijn_amd")
    capi = importlib.import_module("rust-debruijn_amd._capi")
    D = importlib.import_module("rust-debruijn_amd.distributed")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    lr = 0 if a.one_device else int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(lr)
    if a.backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
    else:
        dist.init_process_group("gloo")
    dev = torch.device("cuda", lr)
    ctx = dbg.Context(lr)
    eng = D.HipEngine(ctx, dev)
    tr = D.transport_for(eng)
    genome = np.random.default_rng(1234).integers(0, 4, a.reads * world * 150 // 30).astype(np.uint8)
    reads, n_low = make_reads(a.reads, 1000 + rank, genome)
    w, s, l = pack(reads)
    ss, keep = dbg.upload_seqs(dbg.HostSeqs(w, s, l), lr)
    out = {}
    for balance in (1, 0):
        # (merge_dups pinned: left to the library it would turn itself on for the second run -- the host-staged transport of a
        #  shared GPU exposes all of the exchange -- and the two runs would not move the same records)
        tab, st = D.shard_filter_kmers_c(ctx, tr, ss, a.k, False, 0, 2, merge_dups=0, balance=bool(balance))
        dg, n_valid = D.table_digest(tab, dev), int(tab.n)
        ctx.lib.dbg_free_table(ctx.h, C.byref(tab))
        info = [None] * world
        dist.all_gather_object(info, dict(owned=int(st.records_owned), scanned=int(st.records_scanned), digest=dg, bins=int(st.owned_hi - st.owned_lo),
                                          valid=n_valid))
        own = np.array([i["owned"] for i in info], dtype=np.float64)
        out["balanced" if balance else "equal_bins"] = dict(records_owned_max_over_mean=round(float(own.max() / own.mean()), 4),
                                                            records_owned=[int(x) for x in own], bins_owned=[i["bins"] for i in info],
                                                            digest="%016x" % (sum(i["digest"] for i in info) & ((1 << 64) - 1)),
                                                            valid_kmers=sum(i["valid"] for i in info))
    if rank == 0:
        # the single call over all reads (every rank's reads are a function of its seed)
        allr = np.concatenate([make_reads(a.reads, 1000 + r, genome)[0] for r in range(world)])
        w, s, l = pack(allr)
        ss1, keep1 = dbg.upload_seqs(dbg.HostSeqs(w, s, l), lr)
        t = dbg.filter_kmers_dev(ss1, dbg.CountFilter(2), False, a.k, ctx=ctx)
        out["single_call"] = dict(digest="%016x" % D.table_digest(t, dev), valid_kmers=int(t.n))
        ctx.lib.dbg_free_table(ctx.h, C.byref(t))
        out.update(world=world, reads_per_rank=a.reads, k=a.k, low_complexity_reads_rank0=n_low)
        ok = (out["balanced"]["digest"] == out["equal_bins"]["digest"] == out["single_call"]["digest"]
              and out["balanced"]["valid_kmers"] == out["single_call"]["valid_kmers"])
        out["ok"] = bool(ok)
        print(json.dumps(out), flush=True)
    dist.barrier()
    D.close_transports()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
