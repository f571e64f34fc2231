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

The Python orchestration that remains (exchange_and_count, second_stage, ...) is the engine-agnostic model of the same flow:
tests/ plug a CPU checker engine in and run it with the gloo backend at world size 2 on hosts without a GPU; its geometry
(owner_bounds, chunk_bounds) comes from the library's own host functions (dbg_shard_owner_bounds, dbg_shard_round_cuts), so
that the CPU tests exercise the arithmetic the product uses.
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


def send_layout(bin_off, bounds, cuts, n_chunks):
    """Record offsets for the scatter such that everything exchange round c sends is ONE contiguous range, ordered by
    destination: layout order = (chunk, destination, bin).  The all-to-all of round c then reads straight from the
    scattered buffer with per-destination split sizes -- no gather copy of the send data.
    -> (per-bin offsets [n_bins + 1 entries, last = total], round edges [n_chunks + 1], in-round splits [c][d])"""
    import torch
    world = len(bounds) - 1
    hist = (bin_off[1:] - bin_off[:-1]).to(torch.int64)
    order = []
    for c in range(n_chunks):
        for d in range(world):
            order.append(torch.arange(bounds[d] + cuts[d][c], bounds[d] + cuts[d][c + 1], device=hist.device))
    perm = torch.cat(order) if order else torch.zeros(0, dtype=torch.int64, device=hist.device)
    csum = torch.zeros(len(perm) + 1, dtype=torch.int64, device=hist.device)
    csum[1:] = torch.cumsum(hist[perm], 0)
    off = torch.zeros(len(hist) + 1, dtype=torch.int64, device=hist.device)
    off[perm] = csum[:-1]
    off[-1] = csum[-1]
    # edges of the (chunk, destination) blocks in the permuted order
    sizes = [cuts[d][c + 1] - cuts[d][c] for c in range(n_chunks) for d in range(world)]
    pos = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    edges = csum[torch.from_numpy(pos).to(csum.device)].tolist()
    round_edge = [edges[c * world] for c in range(n_chunks)] + [edges[-1]]
    splits = [[edges[c * world + d + 1] - edges[c * world + d] for d in range(world)] for c in range(n_chunks)]
    return off, round_edge, splits


def exchange_and_count(engine, plan, bin_off, recs, n_local_kmers, group=None, n_chunks=None, layout=None, stats=None, force=False):
    """All-to-all of super-k-mer records, then count the owned bins.  Works for any world size (including 1) and any
    torch.distributed backend that implements all_to_all_single.

    bin_off = natural per-bin record offsets of the local scan (only its differences -- records per bin -- are used);
    recs = the records laid out by send_layout (layout = its result; None: natural bin order, world 1 only).
    The owned bin range of every rank is cut into n_chunks ranges; the records of range c+1 are exchanged
    (asynchronous all-to-all) while range c is being counted, and the table is sorted once at the end."""
    import time
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    rw = plan.rec_words
    grp = getattr(plan, "bin_group", 1) or 1
    force = force and dist.is_initialized()          # force: take the collective route even at world size 1 (a functional check)
    bounds, n_chunks, cuts = exchange_geometry(plan.n_bins, world, grp, n_chunks, force)
    nb_local = bounds[rank + 1] - bounds[rank]
    hint = max(n_local_kmers, 1)
    hist = (bin_off[1:] - bin_off[:-1]).contiguous()                         # records per bin (local reads)
    if stats is not None:
        stats.update(exchange_bytes_sent=0, exchange_exposed_ms=0.0, exchange_rounds=n_chunks if (world > 1 or force) else 0,
                     records_owned=0, exchange_exposed_ms_by_round=[0.0] * (n_chunks if (world > 1 or force) else 0))
    if world == 1 and not force:
        if stats is not None:
            stats["records_owned"] = int(recs.numel() // max(rw, 1))
        seg_off = torch.zeros(1, nb_local + 1, dtype=torch.int64, device=hist.device)
        seg_off[0, 1:] = torch.cumsum(hist.to(torch.int64), 0)
        engine.sync()
        engine.count_begin(plan, hint)
        engine.count_bins(plan, recs, seg_off, 1, nb_local, hint)
        return engine.count_finish(plan)
    if layout is None:
        raise ValueError("the collective route needs the records in send_layout order")
    _, round_edge, splits = layout
    # 1) per-bin record counts of my bins from every source rank
    recv_hist = torch.empty(world * nb_local, dtype=hist.dtype, device=hist.device)
    _all_to_all(recv_hist, hist, [nb_local] * world, [bounds[d + 1] - bounds[d] for d in range(world)], group)
    recv_hist = recv_hist.view(world, nb_local).to(torch.int64)
    my = cuts[rank]
    csum = torch.zeros(world, nb_local + 1, dtype=torch.int64, device=recv_hist.device)
    csum[:, 1:] = torch.cumsum(recv_hist, dim=1)
    recv_edge = csum[:, torch.tensor(my, device=csum.device)].tolist()       # [s][c] records source s holds before my chunk c

    def launch(c):
        """start exchange round c straight from the scattered buffer; returns (work, recv tensor, per-source record counts)"""
        send = recs[round_edge[c] * rw: round_edge[c + 1] * rw]
        in_split = [x * rw for x in splits[c]]
        cnt = [recv_edge[s][c + 1] - recv_edge[s][c] for s in range(world)]
        recv = torch.empty(max(sum(cnt) * rw, 1), dtype=recs.dtype, device=recs.device)
        work = _all_to_all(recv[:sum(cnt) * rw], send, [x * rw for x in cnt], in_split, group, async_op=True)
        if stats is not None:
            stats["exchange_bytes_sent"] += (sum(in_split) - in_split[rank]) * 8
        return work, recv, cnt

    engine.sync()
    engine.count_begin(plan, hint)
    pending = launch(0)
    for c in range(n_chunks):
        work, recv, cnt = pending
        t0 = time.perf_counter()
        work.wait()
        engine.sync()                                                        # chunk c is complete in device memory
        if stats is not None:
            ex_ms = (time.perf_counter() - t0) * 1e3                          # time the counting kernels could not hide
            stats["exchange_exposed_ms"] += ex_ms
            stats["exchange_exposed_ms_by_round"][c] += ex_ms
            stats["records_owned"] += sum(cnt)
        if c + 1 < n_chunks:
            pending = launch(c + 1)                                          # goes on the wire while chunk c is counted
        lo, hi = my[c], my[c + 1]
        # segment table of the chunk: records of its bin b from source s = [seg[s, b], seg[s, b+1])
        seg = csum[:, lo:hi + 1] - csum[:, lo:lo + 1]
        base = torch.zeros(world, dtype=torch.int64, device=seg.device)
        base[1:] = torch.cumsum(torch.tensor(cnt[:-1], dtype=torch.int64, device=seg.device), 0)
        seg = (seg + base[:, None]).contiguous()
        engine.sync()
        engine.count_bins(plan, recv, seg, world, hi - lo, hint // n_chunks)
    return engine.count_finish(plan)


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
    import os
    import torch
    import torch.distributed as dist
    engine.sync()                                  # the reads may still be in flight on torch's current stream
    if isinstance(engine, HipEngine) and not os.environ.get("DBG_PY_ORCHESTRATION"):
        # the product route: one call into the library
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
    n_local = engine.count_instances(ss, k)
    total = n_max = n_local
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world > 1:
        rdev = "cpu" if dist.get_backend(group) == "gloo" else engine.device
        t = torch.tensor([n_local], dtype=torch.int64, device=rdev)
        dist.all_reduce(t, group=group)
        total = int(t.item())
        t = torch.tensor([n_local], dtype=torch.int64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        n_max = int(t.item())
    max_label = 0
    if summarizer_kind == 1:                       # CountFilterSet: every rank must pick the same colour layout
        max_label = engine.max_label(ss)
        if world > 1:
            t = torch.tensor([max_label], dtype=torch.int64, device=rdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            max_label = int(t.item())
    labels = None
    if summarizer_kind == 1 and max_label >= 64:
        # labels beyond the 64 colours of the counting kernel: a sparse alphabet is mapped to colour indices, the same way on
        # every rank -- the union of the ranks' label sets (a max-reduction of presence flags; NCCL has no bitwise OR)
        pres = torch.from_numpy(engine.label_presence(ss))
        if world > 1:
            pres = pres.to(rdev)
            dist.all_reduce(pres, op=dist.ReduceOp.MAX, group=group)
            pres = pres.cpu()
        if int(pres[65536]):
            raise ValueError("sharded CountFilterSet: labels must be < 65536")
        labels = torch.nonzero(pres[:65536]).flatten().tolist()
        if len(labels) > 64:
            raise ValueError("sharded CountFilterSet: %d distinct labels over all ranks; the sharded path holds 64" % len(labels))
    if merge_dups is None:
        env = os.environ.get("DBG_SHARD_MERGE")
        merge_dups = (world == 2) if env is None else env != "0"
    if stats is not None:
        stats["merge_dups"] = bool(merge_dups)
    plan = engine.plan(k, stranded, summarizer_kind, min_obs, total, max_label, merge_dups=merge_dups, labels=labels)
    bin_off, n_recs = engine.scan(ss, plan)
    layout = None
    force = force_exchange and dist.is_initialized()
    if n_chunks is None:
        # One message (one peer, one round) stays under 1 GiB: RCCL transfers of 2 GiB and more were seen to arrive incomplete.
        # Records are at most ~2.5 bytes per k-mer instance (24-byte records of >= 10 k-mers at k = 47; denser for small k),
        # and every rank must arrive at the same number of rounds, hence the estimate from the largest rank.
        est = max(n_max, 1) * 4 // max(world, 1)
        n_chunks = max(4 if world < 4 else 8, -(-est // (1 << 30)))
    if world > 1 or force:
        bounds, nch, cuts = exchange_geometry(plan.n_bins, world, getattr(plan, "bin_group", 1) or 1, n_chunks, force)
        layout = send_layout(bin_off, bounds, cuts, nch)
    recs = engine.scatter(plan, layout[0] if layout else bin_off, n_recs)
    tab = exchange_and_count(engine, plan, bin_off, recs, n_local, group, n_chunks, layout, stats, force)
    return tab, total, n_local, n_recs


# ------------------------------------------------------------------------------------------------
# Rank-spanning end of the pipeline.  The reference's sharded flow ends with per-shard compress_kmers_with_hash,
# BaseGraph::combine over the shard graphs and compress_graph across the shard boundaries
# (src/test.rs:459-470, src/graph.rs:71-100, src/compression.rs:291-349).  Here a shard = the bins a rank owns.
# ------------------------------------------------------------------------------------------------
def unify_classes(class_tables):
    """Per-rank label-list class tables (lists of label tuples, rank-local ids) -> (global table, one remap array per
    rank).  Global ids are ranks in the sorted order of the distinct label tuples, so every rank -- and a checker --
    arrives at the same ids without looking at the k-mers."""
    glob = sorted(set(t for tab in class_tables for t in tab))
    pos = {t: i for i, t in enumerate(glob)}
    return glob, [np.array([pos[t] for t in tab], dtype=np.uint32) for tab in class_tables]


def _graph_payload(g):
    a = g.arrays()
    return dict(k=g.k, stranded=g.stranded, n_bases=a["n_bases"], classes=g.classes,
                **{n: np.ascontiguousarray(a[n]) for n in ("words", "start", "length", "exts", "data")})


def _graph_from_payload(p):
    from . import BaseGraph, PackedDnaStringSet
    g = BaseGraph(p["k"], PackedDnaStringSet(p["words"], p["start"], p["length"], p["n_bases"]), p["exts"], p["data"], p["stranded"])
    g.classes = p["classes"]
    return g


def _gather_payloads(payload, dst, group):
    """every rank's graph arrays to rank dst: sizes by all_gather_object, the arrays as point-to-point tensor transfers
    (device tensors with the nccl backend = RCCL over xGMI, host tensors with gloo)"""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    names = ("words", "start", "length", "exts", "data")
    meta = {n: (payload[n].dtype.str, len(payload[n])) for n in names}
    small = {k_: v for k_, v in payload.items() if k_ not in names}
    metas = [None] * world
    dist.all_gather_object(metas, (meta, small), group=group)
    on_dev = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_dev else torch.device("cpu")
    if rank != dst:
        reqs = []
        for n in names:
            if len(payload[n]):
                t = torch.from_numpy(payload[n].view(np.uint8)).to(dev)
                reqs.append(dist.isend(t, dist.get_global_rank(group, dst) if group is not None else dst, group=group))
        for r in reqs:
            r.wait()
        return None
    out = []
    for src in range(world):
        m, sm = metas[src]
        if src == rank:
            out.append(payload)
            continue
        p = dict(sm)
        for n in names:
            dt, ln = np.dtype(m[n][0]), m[n][1]
            if ln:
                t = torch.empty(ln * dt.itemsize, dtype=torch.uint8, device=dev)
                dist.recv(t, dist.get_global_rank(group, src) if group is not None else src, group=group)
                p[n] = t.cpu().numpy().view(dt)
            else:
                p[n] = np.zeros(0, dt)
        out.append(p)
    return out


def second_stage(engine, graphs, stranded, spec):
    """BaseGraph::combine + compress_graph over per-shard graphs (test.rs:468-470); label-list classes are first brought
    to one global numbering.  -> the final graph (classes attached when the data are label-list classes)."""
    glob = None
    if graphs and graphs[0].classes is not None:
        from . import BaseGraph
        glob, remaps = unify_classes([g.classes for g in graphs])
        renumbered = []                                # the callers' graphs keep their rank-local ids and class tables
        for g, m in zip(graphs, remaps):
            h = BaseGraph(g.k, g.sequences, g.exts, m[np.asarray(g.data, dtype=np.int64)] if len(g.data) else np.zeros(0, np.uint32),
                          g.stranded)
            h.classes = glob
            renumbered.append(h)
        graphs = renumbered
    out = engine.compress_graph(stranded, spec, engine.combine(graphs))
    out.classes = glob
    return out


def sharded_compress(engine, tab, k, stranded, spec, group=None, dst=0, second_spec=None):
    """The compress stage of the sharded pipeline: every rank compresses the table of the bins it owns
    (compress_kmers_with_hash, index resident in HBM), the per-rank unitig graphs travel to rank `dst`, which runs
    BaseGraph::combine + compress_graph across the shard boundaries.  Returns (final graph on dst / None elsewhere,
    this rank's own shard graph)."""
    import os
    import torch.distributed as dist
    if isinstance(engine, HipEngine) and not os.environ.get("DBG_PY_ORCHESTRATION"):
        tr = transport_for(engine, group)
        mode = 1 if os.environ.get("DBG_SHARD_REDUCE") == "tree" else 0
        return shard_compress_c(engine.ctx, tr, tab, k, stranded, spec, second_spec, reduce=mode, root=dst)
    local = engine.compress_table(tab, k, stranded, spec)
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return second_stage(engine, [local], stranded, second_spec or spec), local
    rank = dist.get_rank(group)
    payloads = _gather_payloads(_graph_payload(local), dst, group)
    if rank != dst:
        return None, local
    graphs = [_graph_from_payload(p) for p in payloads]
    return second_stage(engine, graphs, stranded, second_spec or spec), local


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
        val = _dev_tensor(tab.set_val, int(tab.n_set_val), "<i4", device).to(torch.int64)
        owner = torch.repeat_interleave(_mix(lo, 6364136223846793005) + hi, off[1:] - off[:-1])
        d += int(_mix(owner + (val + 1) * 1442695040888963407, -7046029254386353131).sum().item())
    return d & ((1 << 64) - 1)
