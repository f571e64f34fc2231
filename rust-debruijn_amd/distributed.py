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
        """torch cuda tensors -> dbg_seqset (device pointers)."""
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

    def free_table(self, tab):
        self.lib.dbg_free_table(self.ctx.h, C.byref(tab))


def owner_bounds(n_bins, world, group=1):
    """rank r owns bins [bounds[r], bounds[r+1]); boundaries are multiples of `group` (a bin's length
    classes stay together)."""
    return [(r * (n_bins // group) // world) * group for r in range(world + 1)]


def exchange_and_count(engine, plan, bin_off, recs, n_local_kmers, group=None):
    """All-to-all of bin-ordered super-k-mer slabs, then count the owned bins.  Works for any
    world size (including 1) and any torch.distributed backend that implements all_to_all."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    rw = plan.rec_words
    bounds = owner_bounds(plan.n_bins, world, getattr(plan, "bin_group", 1) or 1)
    nb_local = bounds[rank + 1] - bounds[rank]
    hist = (bin_off[1:] - bin_off[:-1]).contiguous()                         # records per bin (local reads)
    edge = bin_off[torch.tensor(bounds, device=bin_off.device)].tolist()     # record offset at each owner boundary
    send_counts = [edge[d + 1] - edge[d] for d in range(world)]
    if world == 1:
        recv_hist, recv = hist.view(1, nb_local), recs
        recv_counts = [send_counts[0]]
    else:
        # 1) per-bin record counts of my bins from every source rank
        recv_hist = torch.empty(world * nb_local, dtype=hist.dtype, device=hist.device)
        dist.all_to_all_single(recv_hist, hist, output_split_sizes=[nb_local] * world,
                               input_split_sizes=[bounds[d + 1] - bounds[d] for d in range(world)], group=group)
        recv_hist = recv_hist.view(world, nb_local)
        recv_counts = recv_hist.sum(dim=1).tolist()
        # 2) the records themselves: one contiguous slab per (source, destination) pair
        recv = torch.empty(max(sum(recv_counts) * rw, 1), dtype=recs.dtype, device=recs.device)
        dist.all_to_all_single(recv[:sum(recv_counts) * rw], recs, output_split_sizes=[c * rw for c in recv_counts],
                               input_split_sizes=[c * rw for c in send_counts], group=group)
    # segment table: records of local bin b from source s = [seg_off[s, b], seg_off[s, b+1])
    base = torch.zeros(world, dtype=torch.int64, device=recv_hist.device)
    if world > 1:
        base[1:] = torch.cumsum(torch.tensor(recv_counts[:-1], dtype=torch.int64, device=base.device), 0)
    seg_off = torch.zeros(world, nb_local + 1, dtype=torch.int64, device=recv_hist.device)
    seg_off[:, 1:] = torch.cumsum(recv_hist.to(torch.int64), dim=1)
    seg_off += base[:, None]
    # hint for the output buffer: this rank's share of the instances
    hint = max(n_local_kmers, 1)
    return engine.count(plan, recv, seg_off, world, nb_local, hint)


def sharded_filter_kmers(engine, ss, k, stranded, summarizer_kind, min_obs, group=None):
    """Distributed filter_kmers: returns this rank's table (the valid k-mers of the bins it owns,
    ascending by key) and the global k-mer instance count."""
    import torch
    import torch.distributed as dist
    n_local = engine.count_instances(ss, k)
    total = n_local
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        t = torch.tensor([n_local], dtype=torch.int64, device=engine.device)
        dist.all_reduce(t, group=group)
        total = int(t.item())
    plan = engine.plan(k, stranded, summarizer_kind, min_obs, total)
    bin_off, n_recs = engine.scan(ss, plan)
    recs = engine.scatter(plan, bin_off, n_recs)
    tab = exchange_and_count(engine, plan, bin_off, recs, n_local, group)
    return tab, total, n_local, n_recs
