"""Sharded k-mer counting across GPUs (one rank per GPU, torch.distributed).

The reference scales by MSP sharding: each shard's k-mers are counted independently
(src/msp.rs:279-324, src/filter.rs:121-124, src/test.rs:433-456).  Here the shards are the minimizer
bins of the fast path: every rank scans its own reads into ONE global bin space, rank r owns the
contiguous bin range [r*n_bins/W, (r+1)*n_bins/W), and a single variable-size all-to-all of
super-k-mer records (RCCL over xGMI on GPUs) moves every bin to its owner.  After the exchange there
is no further communication: each rank counts and filters the bins it owns.

The orchestration below is engine-agnostic so that it can be exercised on CPU with the gloo backend
(tests/ plug a CPU checker engine in); the product engine is HipEngine (C ABI -> HIP kernels).
"""
import ctypes as C

import numpy as np

from . import _capi


class HipEngine:
    """Stages of the fast path on one GPU, through the C ABI (include/dbg_mi355x.h, dbg_shard_*)."""

    def __init__(self, ctx, device):
        import torch
        self.torch = torch
        self.ctx = ctx
        self.lib = ctx.lib
        self.device = device

    def seqset(self, words, start, length, data=None, data_width=0):
        """torch cuda tensors -> dbg_seqset (device pointers).  The library runs on the ctx's own stream: whatever torch's
        current stream still has in flight for these tensors (an H2D copy, a generator kernel) is waited for here."""
        self.sync()
        self._keep = (words, start, length, data)
        return _capi.SeqSet(words.data_ptr(), words.numel(), start.data_ptr(), length.data_ptr(), None,
                            data.data_ptr() if data is not None else None, data_width if data is not None else 0,
                            start.numel())

    def count_instances(self, ss, k):
        n = C.c_uint64()
        self.ctx.check(self.lib.dbg_count_kmer_instances_dev(self.ctx.h, C.byref(ss), k, C.byref(n)))
        return n.value

    def plan(self, k, stranded, summarizer_kind, min_obs, total_kmers):
        p = _capi.ShardPlan(k, int(bool(stranded)), summarizer_kind, min_obs, total_kmers, 0, 0, 0)
        self.ctx.check(self.lib.dbg_shard_plan_make(self.ctx.h, C.byref(p)))
        return p

    def scan(self, ss, plan):
        t = self.torch
        bin_off = t.empty(plan.n_bins + 1, dtype=t.int64, device=self.device)
        n = C.c_uint64()
        self.ctx.check(self.lib.dbg_shard_scan_dev(self.ctx.h, C.byref(ss), C.byref(plan), C.byref(n), bin_off.data_ptr()))
        return bin_off, n.value

    def scatter(self, plan, bin_off, n_recs):
        t = self.torch
        recs = t.empty(max(n_recs * plan.rec_words, 1), dtype=t.int64, device=self.device)
        self.ctx.check(self.lib.dbg_shard_scatter_dev(self.ctx.h, bin_off.data_ptr(), recs.data_ptr()))
        return recs[:n_recs * plan.rec_words]

    def count(self, plan, recs, seg_off, n_src, n_bins_local, n_kmers_hint):
        """-> device KmerTable (caller frees with free_table)."""
        self.torch.cuda.synchronize(self.device)
        tab = _capi.KmerTable()
        seg_off = seg_off.contiguous()
        self._keep2 = (recs, seg_off)
        self.ctx.check(self.lib.dbg_shard_count_dev(self.ctx.h, C.byref(plan), recs.data_ptr(), seg_off.data_ptr(), n_src,
                                                    n_bins_local, n_kmers_hint, C.byref(tab)))
        return tab

    # chunked counting: count a range of the owned bins while the next range is still on the wire; one sort at the end
    def count_begin(self, plan, n_kmers_hint):
        self.ctx.check(self.lib.dbg_shard_count_begin(self.ctx.h, C.byref(plan), n_kmers_hint))

    def count_bins(self, plan, recs, seg_off, n_src, n_bins_chunk, n_kmers_units=0):
        """recs / seg_off must be complete in device memory (the caller has synchronised the stream that produced them)."""
        seg_off = seg_off.contiguous()
        self._keep3 = (recs, seg_off)
        self.ctx.check(self.lib.dbg_shard_count_bins_dev(self.ctx.h, recs.data_ptr(), seg_off.data_ptr(), n_src, n_bins_chunk,
                                                         n_kmers_units))

    def count_finish(self, plan):
        tab = _capi.KmerTable()
        self.ctx.check(self.lib.dbg_shard_count_finish(self.ctx.h, C.byref(tab)))
        return tab

    def sync(self):
        """wait for everything queued on torch's current stream (collectives that were wait()-ed, tensor ops)"""
        self.torch.cuda.current_stream(self.device).synchronize()

    def free_table(self, tab):
        self.lib.dbg_free_table(self.ctx.h, C.byref(tab))


def owner_bounds(n_bins, world, group=1):
    """rank r owns bins [bounds[r], bounds[r+1]); boundaries are multiples of `group` (a bin's length
    classes stay together)."""
    return [(r * (n_bins // group) // world) * group for r in range(world + 1)]


class _Done:
    def wait(self):
        return True


def _all_to_all(out, inp, out_splits, in_splits, group, async_op=False):
    """all_to_all_single; with the gloo backend and device tensors (a debugging set-up: several ranks sharing one GPU)
    the payload is staged through host memory, because gloo moves host buffers only."""
    import torch.distributed as dist
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        h_out = out.new_empty(out.shape, device="cpu")
        dist.all_to_all_single(h_out, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
        out.copy_(h_out)
        return _Done()
    w = dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group, async_op=async_op)
    return w if async_op else _Done()


def chunk_bounds(n_local_bins, n_chunks, group=1):
    """split a rank's owned bins into n_chunks contiguous ranges (boundaries multiples of `group`)"""
    return [(c * (n_local_bins // group) // n_chunks) * group for c in range(n_chunks + 1)]


def exchange_and_count(engine, plan, bin_off, recs, n_local_kmers, group=None, n_chunks=None):
    """All-to-all of bin-ordered super-k-mer slabs, then count the owned bins.  Works for any
    world size (including 1) and any torch.distributed backend that implements all_to_all_single.

    The owned bin range of every rank is cut into n_chunks ranges; the records of range c+1 are exchanged
    (asynchronous all-to-all) while range c is being counted, and the table is sorted once at the end."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    rw = plan.rec_words
    grp = getattr(plan, "bin_group", 1) or 1
    bounds = owner_bounds(plan.n_bins, world, grp)
    nb_local = bounds[rank + 1] - bounds[rank]
    hint = max(n_local_kmers, 1)
    hist = (bin_off[1:] - bin_off[:-1]).contiguous()                         # records per bin (local reads)
    if world == 1:
        seg_off = torch.zeros(1, nb_local + 1, dtype=torch.int64, device=hist.device)
        seg_off[0, 1:] = torch.cumsum(hist.to(torch.int64), 0)
        engine.sync()
        engine.count_begin(plan, hint)
        engine.count_bins(plan, recs, seg_off, 1, nb_local, hint)
        return engine.count_finish(plan)
    if n_chunks is None:
        n_chunks = 4
    # the number of exchange rounds and every cut must be the same on all ranks: derive them from the smallest owned range
    n_chunks = max(1, min(n_chunks, max(min(bounds[d + 1] - bounds[d] for d in range(world)) // grp, 1)))
    # 1) per-bin record counts of my bins from every source rank
    recv_hist = torch.empty(world * nb_local, dtype=hist.dtype, device=hist.device)
    _all_to_all(recv_hist, hist, [nb_local] * world, [bounds[d + 1] - bounds[d] for d in range(world)], group)
    recv_hist = recv_hist.view(world, nb_local).to(torch.int64)
    # 2) chunk geometry.  Chunk c of destination d = d's owned bins [cb_d[c], cb_d[c+1]); every rank computes the same cuts.
    cuts = [chunk_bounds(bounds[d + 1] - bounds[d], n_chunks, grp) for d in range(world)]
    gidx = torch.tensor([[bounds[d] + cuts[d][c] for c in range(n_chunks + 1)] for d in range(world)], device=bin_off.device)
    send_edge = bin_off[gidx].tolist()                                       # [d][c] record offset in my bin-ordered `recs`
    my = cuts[rank]
    csum = torch.zeros(world, nb_local + 1, dtype=torch.int64, device=recv_hist.device)
    csum[:, 1:] = torch.cumsum(recv_hist, dim=1)
    recv_edge = csum[:, torch.tensor(my, device=csum.device)].tolist()       # [s][c] records source s holds before my chunk c

    def launch(c):
        """start the exchange of chunk c; returns (work, recv tensor, per-source record counts)"""
        parts = [recs[send_edge[d][c] * rw: send_edge[d][c + 1] * rw] for d in range(world)]
        send = torch.cat(parts) if sum(p.numel() for p in parts) else torch.zeros(0, dtype=recs.dtype, device=recs.device)
        in_split = [p.numel() for p in parts]
        cnt = [recv_edge[s][c + 1] - recv_edge[s][c] for s in range(world)]
        recv = torch.empty(max(sum(cnt) * rw, 1), dtype=recs.dtype, device=recs.device)
        work = _all_to_all(recv[:sum(cnt) * rw], send, [x * rw for x in cnt], in_split, group, async_op=True)
        return work, recv, cnt, send

    engine.sync()
    engine.count_begin(plan, hint)
    pending = launch(0)
    for c in range(n_chunks):
        work, recv, cnt, send = pending
        work.wait()
        engine.sync()                                                        # chunk c is complete in device memory
        if c + 1 < n_chunks:
            pending = launch(c + 1)                                          # goes on the wire while chunk c is counted
        lo, hi = my[c], my[c + 1]
        # segment table of the chunk: records of its bin b from source s = [seg[s, b], seg[s, b+1])
        seg = csum[:, lo:hi + 1] - csum[:, lo:lo + 1]
        base = torch.zeros(world, dtype=torch.int64, device=seg.device)
        if world > 1:
            base[1:] = torch.cumsum(torch.tensor(cnt[:-1], dtype=torch.int64, device=seg.device), 0)
        seg = (seg + base[:, None]).contiguous()
        engine.sync()
        engine.count_bins(plan, recv, seg, world, hi - lo, hint // n_chunks)
        del send
    return engine.count_finish(plan)


def sharded_filter_kmers(engine, ss, k, stranded, summarizer_kind, min_obs, group=None, n_chunks=None):
    """Distributed filter_kmers: returns this rank's table (the valid k-mers of the bins it owns,
    ascending by key) and the global k-mer instance count."""
    import torch
    import torch.distributed as dist
    engine.sync()                                  # the reads may still be in flight on torch's current stream
    n_local = engine.count_instances(ss, k)
    total = n_local
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        t = torch.tensor([n_local], dtype=torch.int64, device="cpu" if dist.get_backend(group) == "gloo" else engine.device)
        dist.all_reduce(t, group=group)
        total = int(t.item())
    plan = engine.plan(k, stranded, summarizer_kind, min_obs, total)
    bin_off, n_recs = engine.scan(ss, plan)
    recs = engine.scatter(plan, bin_off, n_recs)
    tab = exchange_and_count(engine, plan, bin_off, recs, n_local, group, n_chunks)
    return tab, total, n_local, n_recs
