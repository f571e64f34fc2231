"""Sharded k-mer counting across GPUs (one rank per GPU, torch.distributed).

The reference scales by MSP sharding: each shard's k-mers are counted independently
(src/msp.rs:279-324, src/filter.rs:121-124, src/test.rs:433-456).  Here the shards are the minimizer
bins of the fast path: every rank scans its own reads into ONE global bin space, rank r owns the
contiguous bin range [r*n_bins/W, (r+1)*n_bins/W), and a single variable-size all-to-all of
super-k-mer records (RCCL over xGMI on GPUs) moves every bin to its owner.  After the exchange there
is no further communication: each rank counts and filters the bins it owns.

Since round 4 the product flow runs INSIDE the library, behind the C ABI: dbg_shard_filter_kmers_dev (scan, ownership from
the all-reduced record histogram, layout, pipelined exchange rounds ordered by events, counting, sort) and
dbg_shard_compress_dev (per-shard compress, device-resident graphs over the transport, combine + compress_graph), over a
dbg_transport (rust-debruijn_amd/transport.py: the library's RCCL table, or torch.distributed callbacks).  sharded_filter_kmers /
sharded_compress below are thin callers of those two entry points when the engine is HipEngine.

The engine-agnostic Python model of the same flow (the rounds-2/3 orchestration over torch.distributed, which the CPU tests run
with the oracle as engine under gloo) is test scaffolding and lives in tests/model_orchestration.py since round 5; what stays here
is the ctypes convenience over the dbg_shard_* calls (HipEngine), the wrappers of the library's host geometry functions
(owner_bounds, exchange_geometry) and the digest used to compare sharded with single-GPU tables.
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

    def max_label(self, ss):
        m = C.c_uint32()
        self.ctx.check(self.lib.dbg_seqset_max_label_dev(self.ctx.h, C.byref(ss), C.byref(m)))
        return m.value

    def label_presence(self, ss):
        """-> uint8[65537]: 1 where the D1 label occurs in ss ([65536]: some label >= 65536 does); a max-reduction over the
        ranks gives the alphabet of the whole job"""
        bm = (C.c_uint32 * 2049)()
        self.ctx.check(self.lib.dbg_seqset_label_bitmap_dev(self.ctx.h, C.byref(ss), bm))
        words = np.frombuffer(bm, dtype=np.uint32)
        pres = np.zeros(65537, np.uint8)
        pres[:65536] = np.unpackbits(words[:2048].view(np.uint8), bitorder="little")
        pres[65536] = 1 if words[2048] else 0
        return pres

    def plan(self, k, stranded, summarizer_kind, min_obs, total_kmers, max_label=0, merge_dups=True, labels=None):
        """merge_dups: this rank merges its identical super-k-mer records before the exchange (dbg_shard_plan.merge_dups);
        labels: CountFilterSet with labels >= 64 -- the ascending list of ALL ranks' distinct labels (at most 64)"""
        labels = [int(x) for x in labels] if labels is not None else []
        p = _capi.ShardPlan(k, int(bool(stranded)), summarizer_kind, min_obs, total_kmers, 0, 0, 0, max_label,
                            int(bool(merge_dups)), len(labels), (C.c_uint32 * 64)(*labels[:64]))
        if len(labels) > 64:
            p.n_labels = len(labels)                 # (refused by the library with its own message)
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

    # ---- the rank-spanning end of the pipeline (test.rs:459-470): per-shard compress, combine, compress_graph ----
    def compress_table(self, tab, k, stranded, spec):
        """compress_kmers_with_hash on this rank's device-resident table -> host BaseGraph (+ .classes for label sets)"""
        from . import compress_table_dev
        return compress_table_dev(stranded, spec, tab, k, ctx=self.ctx)

    def combine(self, graphs):
        from . import combine_graphs
        return combine_graphs(graphs, ctx=self.ctx)

    def compress_graph(self, stranded, spec, graph):
        from . import compress_graph
        return compress_graph(stranded, spec, graph, ctx=self.ctx)


def owner_bounds(n_bins, world, group=1, group_records=None):
    """rank r owns bins [bounds[r], bounds[r+1]); boundaries are multiples of `group`.  group_records (records per bin group
    over ALL ranks) -> ranges of nearly equal record count; None -> equal numbers of bins.  The library's own arithmetic
    (dbg_shard_owner_bounds: host code, no GPU needed)."""
    lib = _capi.load()
    out = (C.c_uint32 * (world + 1))()
    gr = None
    if group_records is not None:
        gr = np.ascontiguousarray(group_records, np.uint64)
        if len(gr) != n_bins // group:
            raise ValueError("group_records must hold n_bins / group entries")
    if lib.dbg_shard_owner_bounds(gr.ctypes.data if gr is not None else None, n_bins, group, world, out):
        raise ValueError("dbg_shard_owner_bounds: bad arguments")
    return list(out)


def chunk_bounds(n_local_bins, n_chunks, group=1):
    """split a rank's owned bins into n_chunks contiguous ranges (boundaries multiples of `group`)"""
    return [(c * (n_local_bins // group) // n_chunks) * group for c in range(n_chunks + 1)]


def exchange_geometry(n_bins, world, grp=1, n_chunks=None, force=False):
    """Ownership and pipeline cuts of the exchange, from globally known values only (every rank computes the same):
    bounds[r]..bounds[r+1] = bins rank r owns; the owned range of every rank is cut into the same number of chunks
    (cuts[d][c] relative to bounds[d]); chunk c of all destinations travels in exchange round c."""
    bounds = owner_bounds(n_bins, world, grp)
    if n_chunks is None:
        n_chunks = 4
    if world == 1 and not force:
        n_chunks = 1
    # (dbg_shard_round_cuts: the library's host arithmetic; clamps the rounds to what the smallest owner can be cut into)
    lib = _capi.load()
    nr = C.c_uint32(max(1, n_chunks))
    stride = nr.value + 1
    cbuf = (C.c_uint32 * (world * stride))()
    if lib.dbg_shard_round_cuts((C.c_uint32 * (world + 1))(*bounds), world, grp, C.byref(nr), cbuf):
        raise ValueError("dbg_shard_round_cuts: bad arguments")
    n_chunks = nr.value
    cuts = [[cbuf[d * stride + c] for c in range(n_chunks + 1)] for d in range(world)]
    return bounds, n_chunks, cuts


_TRANSPORTS = {}


def transport_for(engine, group=None):
    """the dbg_transport of this (device, process group): made once, kept for the life of the process"""
    import torch.distributed as dist
    if not dist.is_initialized():
        return None
    key = (str(engine.device), id(group))
    if key not in _TRANSPORTS:
        from . import transport as T
        _TRANSPORTS[key] = T.make_transport(engine.device, group)
    return _TRANSPORTS[key]


def close_transports():
    for t in _TRANSPORTS.values():
        t.close()
    _TRANSPORTS.clear()


def shard_filter_kmers_c(ctx, transport, ss, k, stranded, summarizer_kind, min_obs, n_rounds=0, merge_dups=-1, balance=True,
                         force_exchange=False):
    """dbg_shard_filter_kmers_dev: the whole rank-spanning counting flow inside the library.  -> (device KmerTable, ShardStats)"""
    p = _capi.ShardParams(k, int(bool(stranded)), summarizer_kind, min_obs, n_rounds or 0, merge_dups, int(bool(balance)),
                          int(bool(force_exchange)))
    tab, st = _capi.KmerTable(), _capi.ShardStats()
    ctx.check(ctx.lib.dbg_shard_filter_kmers_dev(ctx.h, transport.ptr if transport is not None else None, C.byref(ss), C.byref(p),
                                                 C.byref(tab), C.byref(st)))
    return tab, st


def shard_compress_c(ctx, transport, tab, k, stranded, spec, second_spec=None, reduce=0, root=0):
    """dbg_shard_compress_dev.  -> (final BaseGraph on root / None elsewhere, this rank's shard graph); label-set tables:
    both carry `.classes` = the global class table, data = global class ids"""
    from . import _graph_from_c, _classes_from_c
    fin, loc, cl = _capi.Graph(), _capi.Graph(), _capi.LabelClasses()
    ctx.check(ctx.lib.dbg_shard_compress_dev(ctx.h, transport.ptr if transport is not None else None, k, int(bool(stranded)), spec.kind,
                                             (second_spec or spec).kind, C.byref(tab), reduce, root, C.byref(fin), C.byref(loc),
                                             C.byref(cl)))
    rank = transport.rank if transport is not None else 0
    classes = _classes_from_c(cl) if tab.set_off else None
    local = _graph_from_c(ctx, loc, k)
    local.classes = classes
    final = None
    if rank == root:
        final = _graph_from_c(ctx, fin, k)
        final.classes = classes
    return final, local


def sharded_filter_kmers(engine, ss, k, stranded, summarizer_kind, min_obs, group=None, n_chunks=None, stats=None, force_exchange=False,
                         merge_dups=None):
    """Distributed filter_kmers: returns this rank's table (the valid k-mers of the bins it owns,
    ascending by key) and the global k-mer instance count.  stats (a dict, optional) receives the exchange volume and the
    exchange time the counting kernels could not hide.  merge_dups: merge identical records on the sending rank before the
    exchange (default: at 2 ranks, the one shape where a link carries more than the counting kernels can hide -- DESIGN.md
    section 5; DBG_SHARD_MERGE=0 / 1 forces it off / on)."""
    engine.sync()                                  # the reads may still be in flight on torch's current stream
    if isinstance(engine, HipEngine):
        # one call into the library
        tr = transport_for(engine, group)
        md = -1 if merge_dups is None else int(bool(merge_dups))
        tab, st = shard_filter_kmers_c(engine.ctx, tr, ss, k, stranded, summarizer_kind, min_obs, n_rounds=n_chunks or 0, merge_dups=md,
                                       force_exchange=force_exchange and tr is not None)
        if stats is not None:
            nr = int(st.n_rounds)
            stats.update(exchange_bytes_sent=int(st.bytes_sent), exchange_exposed_ms=float(st.exposed_ms), exchange_rounds=nr,
                         records_owned=int(st.records_owned), exchange_exposed_ms_by_round=[float(st.exposed_ms_round[i]) for i in range(nr)],
                         merge_dups=bool(st.merge_dups), balanced=bool(st.balanced), setup_ms=float(st.setup_ms),
                         owned_bins=int(st.owned_hi - st.owned_lo), n_bins=int(st.n_bins),
                         transport=getattr(tr, "name", None), transport_fallback=getattr(tr, "fallback_reason", None))
        return tab, int(st.total_kmers), int(st.local_kmers), int(st.records_scanned)
    raise TypeError("distributed.sharded_filter_kmers drives the library (HipEngine); the engine-agnostic Python model of the flow "
                    "is test scaffolding: tests/model_orchestration.py")


def sharded_compress(engine, tab, k, stranded, spec, group=None, dst=0, second_spec=None):
    """The compress stage of the sharded pipeline: every rank compresses the table of the bins it owns
    (compress_kmers_with_hash, index resident in HBM), the per-rank unitig graphs travel to rank `dst`, which runs
    BaseGraph::combine + compress_graph across the shard boundaries.  Returns (final graph on dst / None elsewhere,
    this rank's own shard graph)."""
    import os
    if isinstance(engine, HipEngine):
        tr = transport_for(engine, group)
        mode = 1 if os.environ.get("DBG_SHARD_REDUCE") == "tree" else 0
        return shard_compress_c(engine.ctx, tr, tab, k, stranded, spec, second_spec, reduce=mode, root=dst)
    raise TypeError("distributed.sharded_compress drives the library (HipEngine); the engine-agnostic Python model of the flow is "
                    "test scaffolding: tests/model_orchestration.py")


# ------------------------------------------------------------------------------------------------
# Order-independent digest of a device-resident table: a sum over rows of a hash of (key, Exts, count | label list).
# Every k-mer lives on exactly one rank, so the per-rank digests of a sharded run add up (mod 2^64) to the digest of
# the single-GPU table over the same reads -- a whole-table comparison that needs no gather.
# ------------------------------------------------------------------------------------------------
class _DevArray:
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def _dev_tensor(ptr, n, typestr, device):
    import torch
    if not ptr or n == 0:
        return torch.zeros(0, dtype={"<i8": torch.int64, "<i4": torch.int32, "<i2": torch.int16, "|u1": torch.uint8}[typestr], device=device)
    return torch.as_tensor(_DevArray(ptr, n, typestr), device=device)


def _mix(x, c):
    x = x * c
    return x ^ (x >> 29)


def table_digest(tab, device):
    """tab: _capi.KmerTable with device pointers -> python int in [0, 2^64)"""
    import torch
    n = int(tab.n)
    lo = _dev_tensor(tab.key_lo, n, "<i8", device)
    hi = _dev_tensor(tab.key_hi, n, "<i8", device)
    ex = _dev_tensor(tab.exts, n, "|u1", device).to(torch.int64)
    row = _mix(lo, -7046029254386353131) + _mix(hi + 1, -4417276706812531889) + _mix(ex + 5, 1609587929392839161)
    if tab.count:
        cnt = _dev_tensor(tab.count, n, "<i2", device).to(torch.int64) & 0xFFFF
        row = row + _mix(cnt + 11, -3750763034362895579)
    d = int(row.sum().item()) if n else 0
    if tab.set_off and n:
        off = _dev_tensor(tab.set_off, n + 1, "<i8", device)
        val32 = _dev_tensor(tab.set_val, int(tab.n_set_val), "<i4", device)
        own = _mix(lo, 6364136223846793005) + hi
        # in stretches of keys (label lists hold ~10^10 labels at full size: one repeat_interleave of all of them is 80 GB)
        step = 1 << 24
        for a in range(0, n, step):
            b = min(a + step, n)
            o0, o1 = int(off[a].item()), int(off[b].item())
            owner = torch.repeat_interleave(own[a:b], off[a + 1:b + 1] - off[a:b])
            val = val32[o0:o1].to(torch.int64)
            d += int(_mix(owner + (val + 1) * 1442695040888963407, -7046029254386353131).sum().item())
            del owner, val
    return d & ((1 << 64) - 1)
