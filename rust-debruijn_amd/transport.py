"""Transports for the rank-spanning entry points of the C ABI (dbg_transport in include/dbg_mi355x.h).

The flow itself (ownership, layout, rounds, counting, graph merge) runs inside the library
(dbg_shard_filter_kmers_dev, dbg_shard_compress_dev); a transport only moves device buffers between ranks:

  RcclTransport   the product transport: an ncclComm_t of this process's own (created through the library's
                  dbg_rccl_* helpers, the unique id handed round with whatever process group exists) and the library's
                  RCCL table (dbg_transport_rccl_create): ncclSend / ncclRecv groups on the stream the library names,
                  asynchronous to the host.  One rank per GPU.
  TorchTransport  the same table filled with Python callbacks over torch.distributed.  With the gloo backend every payload
                  is staged through host memory, so N ranks can share ONE GPU: the functional check of the N > 1 path on a
                  one-GPU box (and on CPU-only hosts for the pure-host operations).  Synchronous: an operation drains the
                  device, moves the data, and returns.

The reference has no transport: its sharded flow is composed by the caller from per-shard calls (src/test.rs:433-470).
"""
import ctypes as C
import os
import sys
import traceback

from . import _capi


class _DevBytes:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class TorchTransport:
    """dbg_transport over torch.distributed (group: a process group or None = the default one)."""

    def __init__(self, device, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group, self.device = torch, dist, group, device
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.staged = dist.get_backend(group) == "gloo"          # gloo moves host memory only
        self.name = "torch.distributed/%s%s" % (dist.get_backend(group), " (host-staged)" if self.staged else "")
        self._cbs = (_capi.TR_ALL_REDUCE(self._all_reduce), _capi.TR_ALL_GATHER(self._all_gather),
                     _capi.TR_ALL_TO_ALLV(self._all_to_allv), _capi.TR_SEND(self._send), _capi.TR_RECV(self._recv))
        # (poll / abort stay NULL: torch.distributed has its own watchdog; the library then waits without a health check)
        self.table = _capi.Transport(None, self.rank, self.world, *self._cbs)

    @property
    def ptr(self):
        return C.byref(self.table)

    def close(self):
        pass

    # -- helpers --
    def _peer(self, r):
        return self.dist.get_global_rank(self.group, r) if self.group is not None else r

    def _view(self, ptr, nbytes):
        t = self.torch
        if not nbytes:
            return t.zeros(0, dtype=t.uint8, device=self.device)
        return t.as_tensor(_DevBytes(ptr, nbytes), device=self.device)

    def _guard(self, fn):
        try:
            self.torch.cuda.synchronize(self.device)              # everything the library queued before the call is complete
            fn()
            self.torch.cuda.synchronize(self.device)
            return 0
        except BaseException:                                    # an exception must not unwind through the C caller
            traceback.print_exc(file=sys.stderr)
            return 1

    # -- the operations --
    def _all_reduce(self, _self, buf, n, op, _stream):
        def run():
            t = self._view(buf, n * 8).view(self.torch.int64)
            rop = self.dist.ReduceOp.MAX if op == 1 else self.dist.ReduceOp.SUM
            if self.staged:
                h = t.cpu()
                self.dist.all_reduce(h, op=rop, group=self.group)
                t.copy_(h)
            else:
                self.dist.all_reduce(t, op=rop, group=self.group)
        return self._guard(run)

    def _all_gather(self, _self, send, recv, nbytes, _stream):
        def run():
            s, r = self._view(send, nbytes), self._view(recv, nbytes * self.world)
            if self.staged:
                hs = s.cpu()
                parts = [self.torch.empty_like(hs) for _ in range(self.world)]
                self.dist.all_gather(parts, hs, group=self.group)
                r.copy_(self.torch.cat(parts))
            else:
                self.dist.all_gather_into_tensor(r, s, group=self.group)
        return self._guard(run)

    def _all_to_allv(self, _self, send, soff, sbytes, recv, roff, rbytes, _stream):
        def run():
            W, me = self.world, self.rank
            so, sb = [int(soff[d]) for d in range(W)], [int(sbytes[d]) for d in range(W)]
            ro, rb = [int(roff[d]) for d in range(W)], [int(rbytes[d]) for d in range(W)]
            ops, keep, land = [], [], []
            for i in range(W):
                to, frm = (me + i) % W, (me - i) % W
                if sb[to]:
                    t = self._view(send + so[to], sb[to])
                    if self.staged:
                        t = t.cpu()
                    keep.append(t)
                    if to == me:                                  # self traffic (the library does not generate any)
                        land.append((self._view(recv + ro[me], rb[me]), t))
                        continue
                    ops.append(self.dist.P2POp(self.dist.isend, t, self._peer(to), group=self.group))
                if rb[frm] and frm != me:
                    dst = self._view(recv + ro[frm], rb[frm])
                    if self.staged:
                        h = self.torch.empty(rb[frm], dtype=self.torch.uint8)
                        land.append((dst, h))
                        dst = h
                    keep.append(dst)
                    ops.append(self.dist.P2POp(self.dist.irecv, dst, self._peer(frm), group=self.group))
            if ops:
                for w in self.dist.batch_isend_irecv(ops):
                    w.wait()
            for dst, src in land:
                dst.copy_(src)
        return self._guard(run)

    def _send(self, _self, buf, nbytes, peer, _stream):
        def run():
            t = self._view(buf, nbytes)
            self.dist.send(t.cpu() if self.staged else t, self._peer(peer), group=self.group)
        return self._guard(run)

    def _recv(self, _self, buf, nbytes, peer, _stream):
        def run():
            t = self._view(buf, nbytes)
            if self.staged:
                h = self.torch.empty(nbytes, dtype=self.torch.uint8)
                self.dist.recv(h, self._peer(peer), group=self.group)
                t.copy_(h)
            else:
                self.dist.recv(t, self._peer(peer), group=self.group)
        return self._guard(run)


def _torch_librccl():
    """the librccl torch loaded into this process (a second copy next to it would be a second RCCL runtime)"""
    try:
        import torch
        p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        return p if os.path.exists(p) else None
    except ImportError:
        return None


class RcclTransport:
    """The library's RCCL transport on a communicator of this process's own.  `bootstrap(obj_or_None) -> obj` hands rank 0's
    128-byte unique id to every rank (default: torch.distributed.broadcast_object_list on `group`); rank / world default to
    the group's.  One rank per GPU (RCCL refuses two ranks on one device)."""

    def __init__(self, device_index, rank=None, world=None, group=None, librccl=None, bootstrap=None):
        lib = _capi.load()
        self.lib = lib
        self.path = (librccl or os.environ.get("DBG_LIBRCCL") or _torch_librccl() or "").encode() or None
        if rank is None or world is None:
            import torch.distributed as dist
            rank = dist.get_rank(group) if rank is None else rank
            world = dist.get_world_size(group) if world is None else world
        if bootstrap is None:
            import torch.distributed as dist

            def bootstrap(obj, _g=group):
                box = [obj]
                dist.broadcast_object_list(box, src=dist.get_global_rank(_g, 0) if _g is not None else 0, group=_g)
                return box[0]
        self.rank, self.world = rank, world
        err = C.create_string_buffer(512)
        uid = None
        if rank == 0:
            buf = (C.c_uint8 * 128)()
            if lib.dbg_rccl_unique_id(self.path, buf, err, 512):
                bootstrap(None)                                   # the other ranks must not wait for an id that will not come
                raise RuntimeError("dbg_rccl_unique_id: " + err.value.decode())
            uid = bytes(buf)
        uid = bootstrap(uid)
        if uid is None:
            raise RuntimeError("RcclTransport: rank 0 could not create a unique id")
        self.comm = C.c_void_p()
        idbuf = (C.c_uint8 * 128).from_buffer_copy(uid)
        if lib.dbg_rccl_comm_create(self.path, idbuf, world, rank, device_index, C.byref(self.comm), err, 512):
            raise RuntimeError("dbg_rccl_comm_create: " + err.value.decode())
        tp = C.POINTER(_capi.Transport)()
        if lib.dbg_transport_rccl_create(self.comm, rank, world, self.path, C.byref(tp), err, 512):
            lib.dbg_rccl_comm_destroy(self.path, self.comm)
            raise RuntimeError("dbg_transport_rccl_create: " + err.value.decode())
        self._tp = tp
        self.name = "rccl (library transport, own communicator, %s)" % (self.path.decode() if self.path else "librccl.so.1")

    @property
    def ptr(self):
        return self._tp

    @property
    def table(self):
        return self._tp.contents

    def close(self):
        aborted = bool(getattr(self, "_tp", None)) and bool(self.lib.dbg_transport_aborted(self._tp))
        if getattr(self, "_tp", None):
            self.lib.dbg_transport_destroy(self._tp)
            self._tp = None
        if getattr(self, "comm", None):
            if not aborted:                                   # (ncclCommAbort has already freed an aborted communicator)
                self.lib.dbg_rccl_comm_destroy(self.path, self.comm)
            self.comm = None


def make_transport(device, group=None, prefer=None):
    """The transport for this process group: RCCL through the library when the group's backend is nccl (one rank per GPU),
    the torch.distributed callbacks otherwise (gloo: host-staged, ranks may share a GPU).  prefer = "torch" forces the callbacks;
    DBG_TRANSPORT=torch does the same from the environment.  If the library transport cannot be set up, every rank falls back
    together (the outcome is all-reduced), and the reason is kept in `.fallback_reason`."""
    import torch
    import torch.distributed as dist
    prefer = prefer or os.environ.get("DBG_TRANSPORT")
    dev = torch.device(device)
    if dist.get_backend(group) != "nccl" or prefer == "torch":
        return TorchTransport(dev, group)
    tr, why = None, None
    try:
        tr = RcclTransport(dev.index if dev.index is not None else torch.cuda.current_device(), group=group)
    except Exception as e:                                       # noqa: BLE001 -- reported, and agreed on below
        why = "%s: %s" % (type(e).__name__, e)
    ok = torch.tensor([1 if tr is not None else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if int(ok.item()):
        return tr
    if tr is not None:
        tr.close()
    t = TorchTransport(dev, group)
    t.fallback_reason = why or "another rank could not set up the library's RCCL transport"
    return t
