"""Engine-agnostic Python model of the rank-spanning flow (test scaffolding; moved out of the product package in round 5).

The product runs this flow inside the library, behind dbg_shard_filter_kmers_dev / dbg_shard_compress_dev
(rust-debruijn_amd/distributed.py is the thin caller).  What lives here is the rounds-2/3 orchestration over torch.distributed --
all-reduces, send_layout, the pipelined all_to_all_single rounds, per-shard compress -> gather -> combine -> compress_graph -- kept
because it runs with ANY engine: the CPU tests plug the oracle in (tests/oracle_engine.py) and run it under gloo at world size 2
on hosts without a GPU, and the GPU tests run it with HipEngine as a second opinion on the library's own flow.  Its geometry
(owner_bounds, exchange_geometry) comes from the library's host functions, so the arithmetic exercised is the product's.
Reference flow: src/test.rs:433-470 (msp shards -> per-shard filter_kmers -> per-shard compress -> combine -> compress_graph).
"""
import importlib
import os

import numpy as np

_pkg = importlib.import_module("rust-debruijn_amd")
D = importlib.import_module("rust-debruijn_amd.distributed")
owner_bounds, exchange_geometry, chunk_bounds = D.owner_bounds, D.exchange_geometry, D.chunk_bounds


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


def sharded_filter_kmers(engine, ss, k, stranded, summarizer_kind, min_obs, group=None, n_chunks=None, stats=None, force_exchange=False,
                         merge_dups=None):
    """the model of dbg_shard_filter_kmers_dev: this rank's table (the valid k-mers of the bins it owns, ascending by key), the
    global k-mer instance count, the local one, the records scanned"""
    import torch
    import torch.distributed as dist
    engine.sync()
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
    BaseGraph, PackedDnaStringSet = _pkg.BaseGraph, _pkg.PackedDnaStringSet
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
        BaseGraph = _pkg.BaseGraph
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
    """the model of dbg_shard_compress_dev (gather mode): (final graph on dst / None elsewhere, this rank's own shard graph)"""
    import torch.distributed as dist
    local = engine.compress_table(tab, k, stranded, spec)
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return second_stage(engine, [local], stranded, second_spec or spec), local
    rank = dist.get_rank(group)
    payloads = _gather_payloads(_graph_payload(local), dst, group)
    if rank != dst:
        return None, local
    graphs = [_graph_from_payload(p) for p in payloads]
    return second_stage(engine, graphs, stranded, second_spec or spec), local
