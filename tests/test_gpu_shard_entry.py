"""GPU: the rank-spanning entry points of the C ABI (round 4) -- dbg_shard_filter_kmers_dev and dbg_shard_compress_dev over a
dbg_transport -- on the one GPU of the test box:

  * the library's RCCL transport (dbg_transport_rccl_create on a communicator made with dbg_rccl_comm_create) with ONE rank:
    every operation of the table through its function pointers (ncclAllReduce, ncclAllGather, a ncclSend/ncclRecv group), and
    the whole counting flow with the exchange route forced;
  * the device-resident second stage (compress -> combine -> compress_graph with the graph in HBM throughout) against the same
    flow composed from the host-boundary calls, which tests/test_gpu_graph.py pins against the oracle
    (src/test.rs:459-470, src/graph.rs:71-100, src/compression.rs:291-349).

N > 1 ranks sharing the GPU (torch.distributed callbacks as the transport) run in tests/test_gpu_multirank.py."""
import ctypes as C
import importlib

import numpy as np
import pytest

import oracle_lib as O
from graph_canon import graphs_equal
from pkg import dbg, capi, D

pytestmark = pytest.mark.gpu
T = importlib.import_module("rust-debruijn_amd.transport")


@pytest.fixture(scope="module")
def ctx():
    c = dbg.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def rccl1():
    """a one-rank communicator of our own (no process group: the unique id needs no hand-over)"""
    tr = T.RcclTransport(0, rank=0, world=1, bootstrap=lambda x: x)
    yield tr
    tr.close()


def test_rccl_table_operations_world1(rccl1):
    import torch
    dev = torch.device("cuda", 0)
    tab = rccl1.table
    assert tab.rank == 0 and tab.world == 1
    a = torch.arange(1000, dtype=torch.int64, device=dev) * 7 + 3
    want = a.clone()
    torch.cuda.synchronize()
    assert tab.all_reduce_u64(tab.self, a.data_ptr(), 1000, 0, None) == 0
    assert tab.all_reduce_u64(tab.self, a.data_ptr(), 1000, 1, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(a, want)                                  # one rank: sum and max are the identity
    src = torch.randint(0, 255, (4096,), dtype=torch.uint8, device=dev)
    dst = torch.zeros(4096, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    assert tab.all_gather(tab.self, src.data_ptr(), dst.data_ptr(), 4096, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(src, dst)
    # a send/recv group with this rank's own traffic: 1000 bytes from offset 96 land at offset 8
    dst.zero_()
    u = C.c_uint64 * 1
    torch.cuda.synchronize()
    assert tab.all_to_allv(tab.self, src.data_ptr(), u(96), u(1000), dst.data_ptr(), u(8), u(1000), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(dst[8:1008], src[96:1096]) and int(dst[:8].sum()) == 0 and int(dst[1008:].sum()) == 0


def test_rccl_table_pieces_above_one_gib_world1(rccl1):
    """a pair's message of more than 1 GiB is cut into <= 1 GiB pieces, one send/recv group per piece (transport.hip:
    rccl_all_to_allv, rccl_send / rccl_recv): 2.5 GiB from an odd offset to an odd offset, every byte checked"""
    import torch
    dev = torch.device("cuda", 0)
    tab = rccl1.table
    n = (5 << 29) + 12345                                        # 2.5 GiB and a bit: three pieces
    tot = (n + 4096 + 7) // 8 * 8
    src = (torch.arange(tot // 8, dtype=torch.int64, device=dev) * -7046029254386353131).view(torch.uint8)
    dst = torch.zeros(tot, dtype=torch.uint8, device=dev)
    u = C.c_uint64 * 1
    torch.cuda.synchronize()
    assert tab.all_to_allv(tab.self, src.data_ptr(), u(40), u(n), dst.data_ptr(), u(1000), u(n), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(dst[1000:1000 + n], src[40:40 + n]) and int(dst[:1000].sum()) == 0 and int(dst[1000 + n:].sum()) == 0
    del src, dst
    torch.cuda.empty_cache()


def _table(ctx, t, k):
    h = capi.KmerTable()
    ctx.check(ctx.lib.dbg_table_to_host(ctx.h, C.byref(t), C.byref(h)))
    out = dbg._table_from_c(h, k)
    ctx.lib.dbg_free_table(ctx.h, C.byref(h))
    return out


def _same_table(a, b):
    return (np.array_equal(a.key_hi, b.key_hi) and np.array_equal(a.key_lo, b.key_lo) and np.array_equal(a.exts, b.exts)
            and np.array_equal(a.count, b.count) and np.array_equal(a.set_off, b.set_off) and np.array_equal(a.set_val, b.set_val))


@pytest.mark.parametrize("k,kind", [(47, 1), (47, 0), (31, 0), (63, 1)])
def test_shard_filter_kmers_entry_world1(ctx, rccl1, k, kind):
    """no transport / one-rank RCCL transport with the collective route forced: the table of dbg_filter_kmers_dev"""
    hs = dbg.synth_reads_host(n_reads=30000, read_len=150, error_rate=0.002, stranded=False, n_colours=5)
    if not kind:
        hs = dbg.HostSeqs(hs.words, hs.start, hs.length)
    ss, keep = dbg.upload_seqs(hs, 0)
    summ = dbg.CountFilterSet(2) if kind else dbg.CountFilter(2)
    ref = dbg.filter_kmers_dev(ss, summ, False, k, ctx=ctx)
    want = _table(ctx, ref, k)
    ctx.lib.dbg_free_table(ctx.h, C.byref(ref))
    assert len(want) > 10000
    for tr, force, rounds in ((None, False, 0), (rccl1, True, 0), (rccl1, True, 3), (rccl1, True, 1)):
        t, st = D.shard_filter_kmers_c(ctx, tr, ss, k, False, kind, 2, n_rounds=rounds, force_exchange=force)
        got = _table(ctx, t, k)
        ctx.lib.dbg_free_table(ctx.h, C.byref(t))
        assert _same_table(got, want)
        assert st.total_kmers == 30000 * (150 - k + 1) == st.local_kmers
        if force:
            assert st.n_rounds == (rounds or 4) and st.records_owned == st.records_scanned > 0 and st.bytes_sent == 0
            assert st.owned_lo == 0 and st.owned_hi == st.n_bins


def _host_flow(ctx, tab, k, spec, spec2):
    """the same second stage composed from the host-boundary calls (each pinned against the oracle elsewhere)"""
    g = dbg.compress_table_dev(False, spec, tab, k, ctx=ctx)
    comb = dbg.combine_graphs([g], ctx=ctx)
    return dbg.compress_graph(False, spec2, comb, ctx=ctx), g


@pytest.mark.parametrize("k,kind,mode", [(47, 0, 0), (47, 0, 1), (51, 1, 0), (31, 0, 0), (63, 1, 1)])
def test_shard_compress_entry_world1(ctx, rccl1, k, kind, mode):
    old = ctx.set_option("DBG_COMPRESS", "device")
    try:
        hs = dbg.synth_reads_host(n_reads=20000, read_len=150, error_rate=0.001, stranded=False, n_colours=5)
        if not kind:
            hs = dbg.HostSeqs(hs.words, hs.start, hs.length)
        ss, keep = dbg.upload_seqs(hs, 0)
        tab = dbg.filter_kmers_dev(ss, dbg.CountFilterSet(2) if kind else dbg.CountFilter(2), False, k, ctx=ctx)
        spec = dbg.ScmapCompress() if kind else dbg.SimpleCompress("saturating_add")
        spec2 = spec if kind else dbg.SimpleCompress("max")
        want, want_local = _host_flow(ctx, tab, k, spec, spec2)
        for tr in (None, rccl1):
            final, local = D.shard_compress_c(ctx, tr, tab, k, False, spec, spec2, reduce=mode, root=0)
            # label-set data are class ids in the job-wide numbering (ranks of the sorted distinct label lists); the host flow
            # numbers them per table
            fa, la, wf, wl = final.arrays(), local.arrays(), want.arrays(), want_local.arrays()
            if kind:
                assert final.classes == sorted(set(want_local.classes)) == local.classes
                remap = np.array([final.classes.index(c) for c in want_local.classes], dtype=np.uint32)
                wf, wl = dict(wf, data=remap[wf["data"]]), dict(wl, data=remap[wl["data"]])
            assert len(final) > 10 and graphs_equal(fa, wf)
            assert graphs_equal(la, wl)
        ctx.lib.dbg_free_table(ctx.h, C.byref(tab))
    finally:
        ctx.set_option("DBG_COMPRESS", old)


def test_rccl_table_poll_and_abort_world1(ctx):
    """round 5: the RCCL table's failure handling on a real communicator -- poll is ncclCommGetAsyncError (healthy: 0), abort is
    ncclCommAbort: afterwards every operation of the table fails at once, dbg_transport_aborted says so (the host must not destroy
    the ncclComm_t again), and a rank-spanning call over the dead table returns an error instead of hanging"""
    import torch
    tr = T.RcclTransport(0, rank=0, world=1, bootstrap=lambda x: x)
    try:
        tab = tr.table
        assert tab.poll(tab.self) == 0 and ctx.lib.dbg_transport_aborted(tr.ptr) == 0
        a = torch.arange(16, dtype=torch.int64, device=torch.device("cuda", 0))
        torch.cuda.synchronize()
        assert tab.all_reduce_u64(tab.self, a.data_ptr(), 16, 0, None) == 0
        torch.cuda.synchronize()
        tab.abort(tab.self)
        assert ctx.lib.dbg_transport_aborted(tr.ptr) == 1 and tab.poll(tab.self) != 0
        assert tab.all_reduce_u64(tab.self, a.data_ptr(), 16, 0, None) != 0
        tab.abort(tab.self)                                       # idempotent
        hs = dbg.synth_reads_host(n_reads=2000, read_len=150, error_rate=0.002, stranded=False, n_colours=3)
        dev, hc = capi.SeqSet(), hs.c_struct()
        ctx.check(ctx.lib.dbg_seqset_to_device(ctx.h, C.byref(hc), C.byref(dev)))
        p = capi.ShardParams(47, 0, 0, 2, 2, 0, 1, 1)             # force_exchange: the collective route over the dead table
        tab_out, st = capi.KmerTable(), capi.ShardStats()
        rc = ctx.lib.dbg_shard_filter_kmers_dev(ctx.h, tr.ptr, C.byref(dev), C.byref(p), C.byref(tab_out), C.byref(st))
        ctx.lib.dbg_seqset_free_device(ctx.h, C.byref(dev))
        assert rc != 0 and tab_out.n == 0
    finally:
        tr.close()                                                # (skips ncclCommDestroy: the communicator is gone)
