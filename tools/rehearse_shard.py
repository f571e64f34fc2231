#!/usr/bin/env python3
"""Full-size rehearsal of the two rank-spanning C entry points on ONE GPU (round 5; VERDICT r04 item 1 b/c).

N thread-ranks (the library's in-process transport; one dbg_ctx each on cuda:0) share a stream of synthetic reads the way N GPUs
would: rank r generates reads [r * per, (r + 1) * per) of the stream in HBM, all ranks call dbg_shard_filter_kmers_dev (N owners,
N rounds by default), then dbg_shard_compress_dev in gather and in tree mode.  Checked against the single-GPU calls over the same
reads:
  * the per-rank table digests add up to the digest of dbg_filter_kmers_dev's table (every k-mer lives on exactly one rank),
  * the root's final graph has the digest of the graph dbg_compress_kmers_with_hash_dev builds from the whole table: an
    order- and strand-independent sum over nodes of a hash of {canonical first k-mer, canonical last k-mer}, length and data
    (both graphs are maximal compressions of the same k-mer set, src/compression.rs:291-349, so equal node sets <=> equal digests;
    the single-GPU side censors the table first -- remove_censored_exts, src/filter.rs:238-306 -- because compress_graph's fix_exts
    drops the extensions towards filtered k-mers on the sharded side),
and the per-phase host times of the second stage (DBG_DEBUG lines of dbg_shard_compress_dev) are collected:
per-shard compress, transfer, combine, compress_graph -- the reference's flow is src/test.rs:459-470, src/graph.rs:71-100.

    python tools/rehearse_shard.py --ranks 8 --reads-per-rank 12500000 --k 47 [--summarizer count] [--out profiles/r05_second_stage.txt]
"""
import argparse
import ctypes as C
import importlib
import io
import json
import os
import re
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

M64 = (1 << 64) - 1


def s64(x):
    x &= M64
    return x - (1 << 64) if x >> 63 else x


def lshr(x, n):
    """logical right shift of int64 tensors by a python int 1..63"""
    return (x >> n) & ((1 << (64 - n)) - 1)


def rev2_64(x):
    """reverse the 32 two-bit groups of every int64"""
    for sh, m in ((2, 0x3333333333333333), (4, 0x0F0F0F0F0F0F0F0F), (8, 0x00FF00FF00FF00FF), (16, 0x0000FFFF0000FFFF)):
        x = (lshr(x, sh) & m) | ((x & m) << sh)
    return lshr(x, 32) | (x << 32)


def graph_digest(torch, dev, words, start, length, data, k):
    """order- and strand-independent digest of a BaseGraph's node set (see the module docstring)"""
    n = len(length)
    if n == 0:
        return 0
    w = torch.from_numpy(words.view("int64")).to(dev)
    nw = w.numel()
    st = torch.from_numpy(start.view("int64")).to(dev)
    ln = torch.from_numpy(length.astype("int64")).to(dev)
    da = torch.from_numpy(data.astype("int64")).to(dev)
    SIGN = -(1 << 63)

    def kmer_at(pos):
        wi = pos >> 5
        sh = (pos & 31) * 2
        idx = lambda d: torch.clamp(wi + d, max=nw - 1)
        w0, w1, w2 = w[idx(0)], w[idx(1)], w[idx(2)]
        inv = 64 - sh                                              # 2..64
        z = sh == 0
        r1 = torch.where(z, torch.zeros_like(w1), (w1 >> torch.clamp(inv, max=63)) & ((torch.ones_like(sh) << sh) - 1))
        r2 = torch.where(z, torch.zeros_like(w2), (w2 >> torch.clamp(inv, max=63)) & ((torch.ones_like(sh) << sh) - 1))
        hi = (w0 << sh) | r1
        lo = (w1 << sh) | r2
        if 2 * k <= 64:
            hi = hi & s64(M64 << (64 - 2 * k))
            lo = torch.zeros_like(lo)
        elif 2 * k < 128:
            lo = lo & s64(M64 << (128 - 2 * k))
        # reverse complement, left-aligned again
        chi, clo = rev2_64(~lo), rev2_64(~hi)                      # 128-bit reversal of the complement: the k-mer's rc sits in the LOW 2k bits
        sft = 128 - 2 * k
        if sft >= 64:
            rhi, rlo = clo << (sft - 64) if sft > 64 else clo, torch.zeros_like(clo)
        elif sft == 0:
            rhi, rlo = chi, clo
        else:
            rhi, rlo = (chi << sft) | lshr(clo, 64 - sft), clo << sft
        less = ((rhi ^ SIGN) < (hi ^ SIGN)) | ((rhi == hi) & ((rlo ^ SIGN) < (lo ^ SIGN)))
        return torch.where(less, rhi, hi), torch.where(less, rlo, lo)

    def mix(a, b):
        x = a * s64(0x9E3779B97F4A7C15) + b * s64(0xC2B2AE3D27D4EB4F)
        x = x ^ lshr(x, 29)
        x = x * s64(0xD6E8FEB86659FD93)
        return x ^ lshr(x, 32)
    fh, fl = kmer_at(st)
    lh, ll = kmer_at(st + ln - k)
    ends = mix(fh, fl) + mix(lh, ll)                              # commutative: a node and its reverse complement agree
    node = mix(ends + ln * s64(0x9FB21C651E98DF25), da + 1)
    return int(node.sum().item()) & M64


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--reads-per-rank", type=int, default=12_500_000)
    ap.add_argument("--k", type=int, default=47)
    ap.add_argument("--summarizer", default="count", choices=["count", "set"])
    ap.add_argument("--rounds", type=int, default=0, help="exchange rounds (0 = the library's default: 8 from four ranks on)")
    ap.add_argument("--colours", type=int, default=4, help="distinct D1 labels of the synthetic reads (read index mod colours; at most 255)")
    ap.add_argument("--no-label-groups", action="store_true", help="DBG_NO_LABEL_GROUPS=1 on every ctx: 65..1024 labels take the key-range route")
    ap.add_argument("--labels", type=int, default=0, help="CountFilterSet with this many distinct u32 labels spread over [0, 2^24), one per read by a hash of its index (overrides --colours)")
    ap.add_argument("--lists", default=None, help="DBG_LABEL_LISTS on every ctx (0: label groups / the key-range route beyond 64 colours)")
    ap.add_argument("--no-compress", action="store_true")
    ap.add_argument("--per-device", action="store_true", help="thread-rank r drives device r %% device_count (a multi-GPU node: real peer copies)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    os.environ.setdefault("DBG_INPROC_TIMEOUT_S", "600")
    import numpy as np
    import torch
    dbg = importlib.import_module("rust-debruijn_amd")
    capi = importlib.import_module("rust-debruijn_amd._capi")
    D = importlib.import_module("rust-debruijn_amd.distributed")
    lib = capi.load()
    dev = torch.device("cuda", 0)
    n_dev = torch.cuda.device_count() if args.per_device else 1
    rank_dev = lambda r: torch.device("cuda", r % n_dev)
    W, per, k, L = args.ranks, args.reads_per_rank, args.k, 150
    is_set = args.summarizer == "set"
    total_reads = W * per
    genome_len = total_reads * L // 30
    log = io.StringIO()

    def say(*a):
        line = " ".join(str(x) for x in a)
        print(line, flush=True)
        log.write(line + "\n")

    def synth(ctx, n, first, dev=dev):
        p = dbg.synth_params(n_reads=n, read_len=L, genome_len=genome_len, error_rate=0.001, stranded=False, n_colours=args.colours, first_read=first)
        nw = lib.dbg_synth_words(C.byref(p))
        t = dict(words=torch.empty(nw, dtype=torch.int64, device=dev), start=torch.empty(n, dtype=torch.int64, device=dev),
                 length=torch.empty(n, dtype=torch.int32, device=dev), colour=torch.empty(n, dtype=torch.uint8, device=dev))
        ctx.check(lib.dbg_synth_reads_dev(ctx.h, C.byref(p), t["words"].data_ptr(), t["start"].data_ptr(), t["length"].data_ptr(), t["colour"].data_ptr()))
        torch.cuda.synchronize()
        if args.labels and is_set:
            g = torch.Generator(device=dev); g.manual_seed(11)
            alphabet = torch.randperm(1 << 24, device=dev, generator=g)[:args.labels].to(torch.int32)
            idx = torch.arange(first, first + n, device=dev, dtype=torch.int64)
            t["label"] = alphabet[((idx * 2654435761) % 4294967291) % args.labels].contiguous()
            del idx
            torch.cuda.synchronize()
            return capi.SeqSet(t["words"].data_ptr(), nw, t["start"].data_ptr(), t["length"].data_ptr(), None, t["label"].data_ptr(), 4, n), t
        ss = capi.SeqSet(t["words"].data_ptr(), nw, t["start"].data_ptr(), t["length"].data_ptr(), None,
                         t["colour"].data_ptr() if is_set else None, 1 if is_set else 0, n)
        return ss, t

    spec = dbg.SimpleCompress("saturating_add")
    say("# rehearse_shard: %d thread-ranks x %d reads (%d in all), k = %d, %s(2), %d labels%s, one GPU, in-process transport"
        % (W, per, total_reads, k, "CountFilterSet" if is_set else "CountFilter", args.labels or args.colours,
           (" (label groups off)" if args.no_label_groups else "") + (" DBG_LABEL_LISTS=%s" % args.lists if args.lists is not None else "")))

    # ---- the single-GPU calls over the same reads ----
    ctx0 = dbg.Context(0)
    if args.lists is not None:
        ctx0.set_option("DBG_LABEL_LISTS", args.lists)
    ss, keep = synth(ctx0, total_reads, 0)
    fp = capi.FilterParams(k, 0, 1 if is_set else 0, 2, 0, 4)
    t1 = capi.KmerTable()
    t0 = time.perf_counter()
    ctx0.check(lib.dbg_filter_kmers_dev(ctx0.h, C.byref(ss), C.byref(fp), C.byref(t1)))
    single_filter_s = time.perf_counter() - t0
    single_digest, single_valid = D.table_digest(t1, dev), int(t1.n)
    say("single call: %d valid k-mers, table digest %016x, %.3f s (first call of the ctx)" % (single_valid, single_digest, single_filter_s))
    single_graph_digest = None
    if not args.no_compress and not is_set:
        # The sharded flow ends in compress_graph, whose fix_exts (src/graph.rs:337-377, src/compression.rs:309, :331) drops every
        # extension that leads to no node -- the hanging Exts towards k-mers the filter removed.  The single-GPU counterpart of that
        # is the pipeline real callers run: filter_kmers -> remove_censored_exts (src/filter.rs:238-306) -> compress_kmers_with_hash.
        # (Without the censoring step the plain compress stops at every such hanging extension: 1.9e7 unitigs instead of 1.3e5 here.)
        t0 = time.perf_counter()
        ctx0.check(lib.dbg_remove_censored_exts(ctx0.h, k, 0, C.byref(t1), 0))
        censor_s = time.perf_counter() - t0
        g = capi.Graph()
        t0 = time.perf_counter()
        ctx0.check(lib.dbg_compress_kmers_with_hash_dev(ctx0.h, k, 0, spec.kind, t1.n, t1.key_hi, t1.key_lo, t1.exts, None, t1.count, C.byref(g)))
        single_compress_s = time.perf_counter() - t0
        arr = lambda ptr, n_, ty: np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ty)), shape=(max(int(n_), 1),))[:int(n_)]
        single_graph_digest = graph_digest(torch, dev, arr(g.seq_words, g.n_seq_words, C.c_uint64), arr(g.start, g.n_nodes, C.c_uint64),
                                           arr(g.length, g.n_nodes, C.c_uint32), arr(g.data, g.n_nodes, C.c_uint32), k)
        say("single remove_censored_exts %.3f s + compress_kmers_with_hash (index in HBM): %d unitigs, %d bases, graph digest %016x, %.3f s"
            % (censor_s, g.n_nodes, g.seq_len_bases, single_graph_digest, single_compress_s))
        single_nodes = int(g.n_nodes)
        lib.dbg_free_graph(ctx0.h, C.byref(g))
    lib.dbg_free_table(ctx0.h, C.byref(t1))
    del keep, ss
    ctx0.close()
    torch.cuda.empty_cache()

    # ---- the same over W thread-ranks ----
    arr_t = (C.POINTER(capi.Transport) * W)()
    assert lib.dbg_transport_inprocess_create(W, arr_t) == 0
    ctxs = [dbg.Context(r % n_dev) for r in range(W)]

    def account(what):
        """the ctxs' allocation accounts (dbg_ctx_get_stats) since the last call of this function: where wall time that no kernel
        explains went -- driver allocations, pool trims after a failed allocation (eight ctx pools share ONE device here)"""
        tot = {}
        for c_ in ctxs:
            st_ = c_.stats()
            last = getattr(c_, "_acct", {})
            for kk in ("s_hipmalloc", "s_free", "n_hipmalloc", "n_trims", "n_oom_retries", "n_raw_free"):
                tot[kk] = tot.get(kk, 0) + st_[kk] - last.get(kk, 0)
            tot["pooled_high_water_max"] = max(tot.get("pooled_high_water_max", 0), st_["pooled_high_water"])
            c_._acct = st_
        say("  allocation account of %s, summed over the %d ctxs: hipMalloc %.3f s (%d calls), hipFree %.3f s (%d blocks), %d pool trims after %d failed "
            "allocations; largest pool of a ctx %.1f GB" % (what, W, tot["s_hipmalloc"], tot["n_hipmalloc"], tot["s_free"], tot["n_raw_free"], tot["n_trims"],
                                                           tot["n_oom_retries"], tot["pooled_high_water_max"] / 1e9))
    if args.no_label_groups:
        for c_ in ctxs:
            c_.set_option("DBG_NO_LABEL_GROUPS", "1")
    if args.lists is not None:
        for c_ in ctxs:
            c_.set_option("DBG_LABEL_LISTS", args.lists)
    res = [None] * W
    err = [None] * W

    def run(fn):
        def body(r):
            try:
                res[r] = fn(r)
            except BaseException as e:                           # noqa: BLE001
                err[r] = e
        th = [threading.Thread(target=body, args=(r,)) for r in range(W)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        for r, e in enumerate(err):
            if e is not None:
                raise SystemExit("rank %d: %r" % (r, e))
        return time.perf_counter() - t0

    reads = [synth(ctxs[r], per, r * per, rank_dev(r)) for r in range(W)]

    def do_filter(r):
        p = capi.ShardParams(k, 0, 1 if is_set else 0, 2, args.rounds, -1, 1, 0)
        tab, st = capi.KmerTable(), capi.ShardStats()
        ctxs[r].check(lib.dbg_shard_filter_kmers_dev(ctxs[r].h, arr_t[r], C.byref(reads[r][0]), C.byref(p), C.byref(tab), C.byref(st)))
        return tab, st
    secs = run(do_filter)
    # once more on the warm ctxs (the pools hold the first call's blocks): what is left of the wall time is kernels + transfers
    first = [x[0] for x in res]
    secs_warm = run(do_filter)
    for r_, t_ in enumerate(first):
        lib.dbg_free_table(ctxs[r_].h, C.byref(t_))
    tabs = [x[0] for x in res]
    stats = [x[1] for x in res]
    dsum = sum(D.table_digest(t, rank_dev(r_)) for r_, t in enumerate(tabs)) & M64
    vsum = sum(int(t.n) for t in tabs)
    owned = [int(s.records_owned) for s in stats]
    say("dbg_shard_filter_kmers_dev x %d ranks: first call %.3f s wall, second call %.3f s (all ranks on one GPU), rounds %d, sender merge %d, valid %d, digest sum %016x -> %s"
        % (W, secs, secs_warm, int(stats[0].n_rounds), int(stats[0].merge_dups), vsum, dsum, "EQUAL to the single call" if (dsum == single_digest and vsum == single_valid) else "MISMATCH"))
    say("  records owned per rank: min %d max %d (max / mean %.4f); bytes sent per rank: %s"
        % (min(owned), max(owned), max(owned) / (sum(owned) / W), [int(s.bytes_sent) for s in stats]))
    ok = dsum == single_digest and vsum == single_valid
    account("dbg_shard_filter_kmers_dev")
    say("  kernel + transfer time per rank inside the call (HIP events; setup / exposed exchange): setup %s ms, exposed %s ms"
        % ([round(float(s.setup_ms), 1) for s in stats], [round(float(s.exposed_ms), 1) for s in stats]))
    del reads
    torch.cuda.empty_cache()

    if not args.no_compress and not is_set:
        for mode, name in ((0, "gather"), (1, "tree")):
            for c in ctxs:
                c.set_option("DBG_DEBUG", "1")
            # the library writes its phase lines to stderr: capture the process's fd 2 for the duration of the call
            sys.stderr.flush()
            tmp = tempfile.TemporaryFile(mode="w+b")
            saved = os.dup(2)
            os.dup2(tmp.fileno(), 2)

            def do_compress(r, mode=mode):
                fin, loc, cl = capi.Graph(), capi.Graph(), capi.LabelClasses()
                ctxs[r].check(lib.dbg_shard_compress_dev(ctxs[r].h, arr_t[r], k, 0, spec.kind, spec.kind, C.byref(tabs[r]), mode, 0, C.byref(fin), None, C.byref(cl)))
                return fin
            ctxs[0].enable_timing(True)
            try:
                secs = run(do_compress)
            finally:
                sys.stderr.flush()
                os.dup2(saved, 2)
                os.close(saved)
            for c in ctxs:
                c.set_option("DBG_DEBUG", None)
            tmp.seek(0)
            lines = [l for l in tmp.read().decode(errors="replace").splitlines() if l.startswith("[shard_compress]")]
            tmp.close()
            fin = res[0]
            arr = lambda ptr, n_, ty: np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ty)), shape=(max(int(n_), 1),))[:int(n_)]
            gd = graph_digest(torch, dev, arr(fin.seq_words, fin.n_seq_words, C.c_uint64), arr(fin.start, fin.n_nodes, C.c_uint64),
                              arr(fin.length, fin.n_nodes, C.c_uint32), arr(fin.data, fin.n_nodes, C.c_uint32), k)
            same = gd == single_graph_digest and int(fin.n_nodes) == single_nodes
            ok = ok and same
            say("dbg_shard_compress_dev %s x %d ranks: %.3f s wall, %d unitigs, %d bases, graph digest %016x -> %s"
                % (name, W, secs, fin.n_nodes, fin.seq_len_bases, gd, "EQUAL to the single-GPU graph" if same else "MISMATCH"))
            ph = {}
            for l in lines:
                m = re.match(r"\[shard_compress\] rank=(\d+) phase=(\S+) ms=([\d.]+) nodes=(\d+)", l)
                if m:
                    ph.setdefault(m.group(2), []).append((int(m.group(1)), float(m.group(3)), int(m.group(4))))
            for name2, v in ph.items():
                root = [x for x in v if x[0] == 0]
                say("    %-30s ranks %2d  max %9.2f ms  root %9.2f ms  nodes (root) %d" % (name2, len(v), max(x[1] for x in v), root[-1][1] if root else 0.0, root[-1][2] if root else 0))
            kt = sorted(ctxs[0].timings(), key=lambda t: -t["ms"])
            ctxs[0].enable_timing(False)
            say("    root's kernels (HIP events, ms): " + ", ".join("%s %.1f" % (t["name"], t["ms"]) for t in kt[:14]))
            account("dbg_shard_compress_dev %s" % name)
            for r in range(W):
                lib.dbg_free_graph(ctxs[r].h, C.byref(res[r]))
    for r in range(W):
        lib.dbg_free_table(ctxs[r].h, C.byref(tabs[r]))
        ctxs[r].close()
        lib.dbg_transport_destroy(arr_t[r])
    say("rehearsal %s" % ("ok" if ok else "FAILED"))
    if args.out:
        with open(args.out, "a") as f:
            f.write(log.getvalue() + "\n")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
