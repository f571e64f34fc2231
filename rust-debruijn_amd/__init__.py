"""MI355X-native k-mer hot path of the `debruijn` crate (10XGenomics/rust-debruijn).

Host-side mirror of the reference's interface for this path, over the C ABI in
include/dbg_mi355x.h (hand-written HIP kernels for gfx950 behind it):

    filter_kmers(seqs, summarizer, stranded, report_all_kmers, memory_size)   src/filter.rs:139-148
    msp_sequence(k, seq, permutation, rc)                                      src/msp.rs:279-288
    compress_kmers_with_hash(stranded, spec, index)                            src/compression.rs:588-594

plus the plain-data types those functions exchange (Exts, PackedDnaStringSet, BaseGraph, the
BoomHashMap2 contents as sorted struct-of-arrays).  K (k) and P (p) are type parameters in the
reference and ordinary arguments here.  Everything computes on the GPU: there is no CPU fallback,
and importing works without a GPU only so that the boundary itself can be inspected.
"""
import ctypes as C
import numpy as np

from . import _capi
from ._capi import SeqSet as _SeqSet

__all__ = ["Context", "Exts", "CountFilter", "CountFilterSet", "SimpleCompress", "ScmapCompress",
           "PackedDnaStringSet", "KmerTable", "BaseGraph", "filter_kmers", "msp_sequence",
           "compress_kmers_with_hash", "compress_graph", "combine_graphs", "remove_censored_exts", "remove_censored_exts_sharded",
           "synth_reads_host", "pack_bases", "unpack_bases", "DbgError", "LEFT", "RIGHT"]

LEFT, RIGHT = 0, 1
M64 = (1 << 64) - 1


class DbgError(RuntimeError):
    """Raised where the reference would panic (assert!/panic!) or on a device error."""


# ------------------------------------------------------------------------------------------------
# plain-data mirrors
# ------------------------------------------------------------------------------------------------
class Exts:
    """1-byte neighbour bitmap (src/lib.rs:577-749): low nibble = left A,C,G,T; high nibble = right."""
    __slots__ = ("val",)

    def __init__(self, val=0):
        self.val = int(val) & 0xFF

    @staticmethod
    def empty():
        return Exts(0)

    def has_ext(self, direction, base):
        return bool((self.val >> (4 * direction + base)) & 1)

    def get(self, direction):
        return [b for b in range(4) if self.has_ext(direction, b)]

    def num_ext_dir(self, direction):
        return bin((self.val >> (4 * direction)) & 0xF).count("1")

    def rc(self):
        l, r = self.val & 0xF, self.val >> 4
        rev4 = lambda x: ((x & 1) << 3) | ((x & 2) << 1) | ((x & 4) >> 1) | ((x & 8) >> 3)
        return Exts((rev4(l) << 4) | rev4(r))

    def __eq__(self, o):
        return isinstance(o, Exts) and o.val == self.val

    def __hash__(self):
        return self.val

    def __int__(self):
        return self.val

    def __repr__(self):
        s = lambda d: "".join("ACGT"[b] for b in self.get(d))
        return "%s|%s" % (s(LEFT), s(RIGHT))


class CountFilter:
    """KmerSummarizer: u16 saturating count, valid iff count >= min_kmer_obs (src/filter.rs:40-63)."""
    kind = 0

    def __init__(self, min_kmer_obs):
        self.min_kmer_obs = int(min_kmer_obs)


class CountFilterSet:
    """KmerSummarizer: sorted de-duplicated label list, valid iff nobs >= min_kmer_obs (src/filter.rs:68-101)."""
    kind = 1

    def __init__(self, min_kmer_obs):
        self.min_kmer_obs = int(min_kmer_obs)


class SimpleCompress:
    """CompressionSpec with join_test = true and a reduce closure from a closed set
    (src/compression.rs:40-65): 'saturating_add', 'add_mod_65535', 'max', 'wrapping_add'."""
    _KINDS = {"saturating_add": 0, "add_mod_65535": 1, "max": 2, "wrapping_add": 4}

    def __init__(self, func="saturating_add"):
        if func not in self._KINDS:
            raise ValueError("reduce must be one of %s" % sorted(self._KINDS))
        self.kind = self._KINDS[func]


class ScmapCompress:
    """CompressionSpec joining only equal data (src/compression.rs:68-98)."""
    kind = 3


def pack_bases(b):
    """0-3 bases -> packed u64 words, base i at bits [63-2(i%32), 62-2(i%32)] of word i/32
    (src/dna_string.rs:383-399)."""
    b = np.asarray(b, dtype=np.uint64)
    n = len(b)
    nw = (n + 31) // 32
    pad = np.zeros(nw * 32, dtype=np.uint64)
    pad[:n] = b
    shifts = (62 - 2 * np.arange(32)).astype(np.uint64)
    if nw == 0:
        return np.zeros(0, dtype=np.uint64)
    return np.bitwise_or.reduce(pad.reshape(nw, 32) << shifts, axis=1).astype(np.uint64)


def unpack_bases(words, start, length):
    idx = np.arange(start, start + length, dtype=np.uint64)
    w = np.asarray(words, dtype=np.uint64)[(idx >> np.uint64(5)).astype(np.int64)]
    sh = (np.uint64(62) - np.uint64(2) * (idx & np.uint64(31))).astype(np.uint64)
    return ((w >> sh) & np.uint64(3)).astype(np.uint8)


class PackedDnaStringSet:
    """Concatenated 2-bit store: sequence words + start (base offsets) + length
    (src/dna_string.rs:762-822)."""

    def __init__(self, words=None, start=None, length=None, n_bases=None):
        self.words = np.zeros(0, np.uint64) if words is None else np.ascontiguousarray(words, np.uint64)
        self.start = np.zeros(0, np.uint64) if start is None else np.ascontiguousarray(start, np.uint64)
        self.length = np.zeros(0, np.uint32) if length is None else np.ascontiguousarray(length, np.uint32)
        self.n_bases = int(n_bases) if n_bases is not None else (
            int(self.start[-1] + self.length[-1]) if len(self.start) else 0)

    @staticmethod
    def from_seqs(seqs):
        """PackedDnaStringSet::add for each sequence (back-to-back, not word aligned)."""
        starts, lens, pos, parts = [], [], 0, []
        for s in seqs:
            s = np.asarray(s, dtype=np.uint8)
            starts.append(pos)
            lens.append(len(s))
            parts.append(s)
            pos += len(s)
        cat = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
        return PackedDnaStringSet(pack_bases(cat), starts, lens, pos)

    def __len__(self):
        return len(self.start)

    def get(self, i):
        return unpack_bases(self.words, int(self.start[i]), int(self.length[i]))


class KmerTable:
    """The BoomHashMap2<K, Exts, DS> contents as ascending-key struct-of-arrays (the vectors the
    reference hands to BoomHashMap2::new, src/filter.rs:227-230)."""

    def __init__(self, k):
        self.k = k
        self.key_hi = self.key_lo = self.exts = self.count = self.set_off = self.set_val = None
        self.n_kmer_instances = 0
        self.n_passes = 0

    def __len__(self):
        return len(self.key_lo)

    def keys(self):
        return [(int(h) << 64) | int(l) for h, l in zip(self.key_hi, self.key_lo)]

    def data(self, i):
        if self.count is not None:
            return int(self.count[i])
        return [int(x) for x in self.set_val[int(self.set_off[i]):int(self.set_off[i + 1])]]

    def __iter__(self):
        for i, kmer in enumerate(self.keys()):
            yield kmer, Exts(self.exts[i]), self.data(i)


class BaseGraph:
    """BaseGraph<K, D> (src/graph.rs:43-50)."""

    def __init__(self, k, sequences, exts, data, stranded):
        self.k = k
        self.sequences = sequences
        self.exts = exts
        self.data = data
        self.stranded = stranded
        self.classes = None         # class id -> label tuple when data holds label-list classes (compress_table_dev)

    def __len__(self):
        return len(self.sequences)

    def arrays(self):
        return dict(words=self.sequences.words, start=self.sequences.start, length=self.sequences.length,
                    exts=self.exts, data=self.data, n_bases=self.sequences.n_bases)

    def edges(self, ctx=None):
        """Node::l_edges / r_edges (src/graph.rs:1041-1049) of every node: ([l_edges], [r_edges]) per node, each
        edge (target_node, incoming_dir 'L'|'R', flipped) in the reference's order."""
        return graph_edges(self, ctx)

    def write_gfa(self, ctx=None):
        """DebruijnGraph::write_gfa (src/graph.rs:603-616) -> bytes."""
        return graph_to_gfa(self, ctx)

    def to_gfa(self, path, ctx=None):
        """DebruijnGraph::to_gfa (src/graph.rs:598-601)."""
        ctx = ctx or default_context()
        cg, keep = _graph_to_c(self)
        ctx.check(ctx.lib.dbg_graph_write_gfa(ctx.h, self.k, C.byref(cg), str(path).encode()))


# ------------------------------------------------------------------------------------------------
# context
# ------------------------------------------------------------------------------------------------
class Context:
    """One dbg_ctx: device, stream, pooled scratch.  One per host thread."""

    def __init__(self, device=0):
        self.lib = _capi.load()
        h = C.c_void_p()
        r = self.lib.dbg_ctx_create(device, C.byref(h))
        if r:
            raise DbgError(self.lib.dbg_last_error(None).decode())
        self.h = h
        self.device = device
        import os
        # the knobs as the library saw them in dbg_ctx_create (it never reads the environment again): what set_option restores to
        self._opts = {n: v for n, v in os.environ.items() if n.startswith("DBG_")}

    def set_option(self, name, value):
        """dbg_ctx_set_option: one diagnostic knob of this ctx (the library reads the environment only in dbg_ctx_create).
        value None = unset.  Returns the value the knob had (as far as this wrapper knows: the environment at creation or
        an earlier set_option)."""
        old = self._opts.get(name)
        self.check(self.lib.dbg_ctx_set_option(self.h, name.encode(), None if value is None else str(value).encode()))
        self._opts[name] = value
        return old

    def options(self, **kw):
        """context manager: set knobs for the duration of a with-block"""
        import contextlib

        @contextlib.contextmanager
        def cm():
            old = {n: self.set_option(n, v) for n, v in kw.items()}
            try:
                yield self
            finally:
                for n, v in old.items():
                    self.set_option(n, v)
        return cm()

    def trim(self):
        """dbg_ctx_trim: give the pooled scratch of finished calls back to the driver -> device bytes released"""
        n = C.c_uint64()
        self.check(self.lib.dbg_ctx_trim(self.h, C.byref(n)))
        return n.value

    def close(self):
        if getattr(self, "h", None):
            self.lib.dbg_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, r):
        if r:
            raise DbgError(self.lib.dbg_last_error(self.h).decode())

    def enable_timing(self, on=True):
        self.check(self.lib.dbg_ctx_enable_timing(self.h, int(on)))

    def timings(self):
        arr = (_capi.KernelTime * 64)()
        n = C.c_uint32()
        self.check(self.lib.dbg_ctx_get_timings(self.h, arr, 64, C.byref(n)))
        return [dict(name=arr[i].name.decode(), ms=arr[i].ms, launches=arr[i].launches, units=arr[i].units)
                for i in range(min(n.value, 64))]

    def set_stream(self, hip_stream_ptr):
        self.check(self.lib.dbg_ctx_set_stream(self.h, C.c_void_p(hip_stream_ptr)))

    def stats(self):
        """dbg_ctx_get_stats: allocation account of the ctx (what a cold call pays) and how the scan's slab is backed"""
        st = _capi.CtxStats()
        st.struct_size = C.sizeof(st)
        self.check(self.lib.dbg_ctx_get_stats(self.h, C.byref(st)))
        d = {n: getattr(st, n) for n, _ in st._fields_ if n not in ("struct_size", "slab_note", "slab_trial_ms") and not n.startswith("reserved")}
        d["slab_trial_ms"] = [round(float(x), 3) for x in st.slab_trial_ms[:min(st.slab_trials_done, 8)]]
        d["slab_note"] = st.slab_note.decode(errors="replace")
        d["slab_backing_name"] = _capi.SLAB_BACKING_NAMES.get(st.slab_backing, "?")
        return d

    def probe_slab(self, n_writes=0):
        """dbg_ctx_probe_slab: random record-sized writes into the pooled slab -> (ms, writes per launch); None if there is none"""
        ms, nw = C.c_float(), C.c_uint64()
        if self.lib.dbg_ctx_probe_slab(self.h, n_writes, C.byref(ms), C.byref(nw)):
            return None
        return ms.value, nw.value

    def warm(self, slab_bytes=0, pinned_bytes=0):
        self.check(self.lib.dbg_ctx_warm(self.h, slab_bytes, pinned_bytes))


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


# ------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------
def _np_ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class HostSeqs:
    """&[(V, Exts, D1)] flattened to the C-ABI's dbg_seqset (host arrays)."""

    def __init__(self, words, start, length, exts=None, data=None, data_width=0):
        self.words = np.ascontiguousarray(words, np.uint64)
        self.start = np.ascontiguousarray(start, np.uint64)
        self.length = np.ascontiguousarray(length, np.uint32)
        self.exts = None if exts is None else np.ascontiguousarray(exts, np.uint8)
        self.data_width = data_width if data is not None else 0
        dt = {0: None, 1: np.uint8, 2: np.uint16, 4: np.uint32}[self.data_width]
        self.data = None if data is None else np.ascontiguousarray(data, dt)

    @staticmethod
    def from_tuples(seqs, data_width=None):
        """seqs: iterable of (bases, Exts|int, d) like the reference's &[(V, Exts, D1)]; d may be None/()."""
        seqs = list(seqs)
        ps = PackedDnaStringSet.from_seqs([s[0] for s in seqs])
        exts = np.array([int(s[1]) for s in seqs], dtype=np.uint8)
        has_d = any(not (s[2] is None or s[2] == ()) for s in seqs)
        if has_d:
            dv = [int(s[2]) for s in seqs]
            if data_width is None:
                mx = max(dv) if dv else 0
                data_width = 1 if mx < 256 else (2 if mx < 65536 else 4)
            return HostSeqs(ps.words, ps.start, ps.length, exts, dv, data_width)
        return HostSeqs(ps.words, ps.start, ps.length, exts)

    def c_struct(self):
        s = _SeqSet()
        s.words = _np_ptr(self.words)
        s.n_words = len(self.words)
        s.start = _np_ptr(self.start)
        s.length = _np_ptr(self.length)
        s.exts = _np_ptr(self.exts)
        s.data = _np_ptr(self.data)
        s.data_width = self.data_width
        s.n_seqs = len(self.start)
        return s


def _copy(ptr, n, dtype):
    if not ptr or n == 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(n,)).copy()


def _table_from_c(t, k):
    out = KmerTable(k)
    n = t.n
    out.key_hi = _copy(t.key_hi, n, np.uint64) if t.key_hi else np.zeros(n, np.uint64)     # (compact flag 2: no key_hi for k <= 32)
    out.key_lo = _copy(t.key_lo, n, np.uint64)
    out.exts = _copy(t.exts, n, np.uint8)
    out.count = _copy(t.count, n, np.uint16) if t.count else None
    # (compact tables -- dbg_filter_params.compact_sets -- carry narrow CSR arrays: widened here, the mirror's tables are plain)
    ow = {0: np.uint64, 8: np.uint64, 4: np.uint32}[int(getattr(t, "set_off_width", 0))]
    vw = {0: np.uint32, 4: np.uint32, 2: np.uint16, 1: np.uint8}[int(getattr(t, "set_val_width", 0))]
    out.set_off = _copy(t.set_off, n + 1, ow).astype(np.uint64) if t.set_off else None
    out.set_val = _copy(t.set_val, t.n_set_val, vw).astype(np.uint32) if t.set_off else None
    out.all_hi = _copy(t.all_hi, t.n_all, np.uint64)
    out.all_lo = _copy(t.all_lo, t.n_all, np.uint64)
    out.n_kmer_instances = t.n_kmer_instances
    out.n_passes = t.n_passes
    return out


# ------------------------------------------------------------------------------------------------
# the reference's functions
# ------------------------------------------------------------------------------------------------
def filter_kmers(seqs, summarizer, stranded, report_all_kmers, memory_size, k, ctx=None, compact_sets=False):
    """filter_kmers::<K, V, D1, DS, S> (src/filter.rs:139-231).

    seqs: HostSeqs or an iterable of (bases, Exts, d).  Returns (KmerTable, all_kmers) where
    all_kmers is the list of every observed k-mer (ascending) iff report_all_kmers, else [].
    compact_sets: ask the library for the narrow CSR of a CountFilterSet table (dbg_filter_params.compact_sets); the mirror
    widens it again, so the result is the same table."""
    ctx = ctx or default_context()
    hs = seqs if isinstance(seqs, HostSeqs) else HostSeqs.from_tuples(seqs)
    p = _capi.FilterParams(k, int(bool(stranded)), summarizer.kind, summarizer.min_kmer_obs,
                           int(bool(report_all_kmers)), int(memory_size), int(compact_sets))
    cs = hs.c_struct()
    t = _capi.KmerTable()
    ctx.check(ctx.lib.dbg_filter_kmers(ctx.h, C.byref(cs), C.byref(p), C.byref(t)))
    try:
        out = _table_from_c(t, k)
    finally:
        ctx.lib.dbg_free_table(ctx.h, C.byref(t))
    all_kmers = [(int(h) << 64) | int(l) for h, l in zip(out.all_hi, out.all_lo)]
    return out, all_kmers


def msp_sequence(k, seq, permutation=None, rc=True, p=8, lmer_words=0, ctx=None):
    """msp_sequence::<P, V> (src/msp.rs:279-324) for one 0-3 byte sequence.
    Returns [(bucket, Exts, piece_bases)]; with lmer_words > 0 the third element is the packed
    Lmer<[u64; lmer_words]> words instead (src/vmer.rs:32-47)."""
    res = msp_sequence_batch(k, [seq], permutation, rc, p, lmer_words, ctx)
    seq = np.asarray(seq, dtype=np.uint8)
    out = []
    for i in range(len(res["bucket"])):
        s, l = int(res["start"][i]), int(res["len"][i])
        v = res["lmer"][i] if lmer_words else seq[s:s + l]
        out.append((int(res["bucket"][i]), Exts(res["exts"][i]), v))
    return out


def msp_sequence_batch(k, seqs, permutation=None, rc=True, p=8, lmer_words=0, ctx=None):
    """Batched msp_sequence over many sequences -> dict of arrays (+ piece_off per sequence)."""
    ctx = ctx or default_context()
    hs = seqs if isinstance(seqs, HostSeqs) else HostSeqs.from_tuples([(s, 0, None) for s in seqs])
    perm = None if permutation is None else np.ascontiguousarray(permutation, np.uint32)
    mp = _capi.MspParams(k, p, _np_ptr(perm), int(bool(rc)), lmer_words)
    cs = hs.c_struct()
    pc = _capi.MspPieces()
    ctx.check(ctx.lib.dbg_msp_sequence(ctx.h, C.byref(cs), C.byref(mp), C.byref(pc)))
    try:
        n = pc.n_pieces
        res = dict(piece_off=_copy(pc.piece_off, len(hs.start) + 1, np.uint64), bucket=_copy(pc.bucket, n, np.uint32),
                   exts=_copy(pc.exts, n, np.uint8), start=_copy(pc.start, n, np.uint32),
                   len=_copy(pc.len, n, np.uint16), minimizer_pos=_copy(pc.minimizer_pos, n, np.uint32))
        res["lmer"] = _copy(pc.lmer, n * lmer_words, np.uint64).reshape(n, lmer_words) if lmer_words else None
    finally:
        ctx.lib.dbg_free_pieces(ctx.h, C.byref(pc))
    return res


def compress_kmers_with_hash(stranded, spec, index, k=None, seed_order=None, data=None, ctx=None):
    """compress_kmers_with_hash::<K, D, S> (src/compression.rs:588-594).

    index: KmerTable (its count column is D unless `data` is given).  seed_order: optional
    permutation standing in for the MPHF slot order the reference iterates in
    (compression.rs:574); None = ascending key order."""
    ctx = ctx or default_context()
    k = k or index.k
    n = len(index)
    key_hi = np.ascontiguousarray(index.key_hi, np.uint64)
    key_lo = np.ascontiguousarray(index.key_lo, np.uint64)
    exts = np.ascontiguousarray(index.exts, np.uint8)
    if data is None:
        data = index.count if index.count is not None else np.zeros(n, np.uint32)
    data = np.ascontiguousarray(data, np.uint32)
    so = None if seed_order is None else np.ascontiguousarray(seed_order, np.uint64)
    g = _capi.Graph()
    ctx.check(ctx.lib.dbg_compress_kmers_with_hash(ctx.h, k, int(bool(stranded)), spec.kind, n, _np_ptr(key_hi),
                                                   _np_ptr(key_lo), _np_ptr(exts), _np_ptr(data), _np_ptr(so),
                                                   C.byref(g)))
    try:
        seqs = PackedDnaStringSet(_copy(g.seq_words, g.n_seq_words, np.uint64), _copy(g.start, g.n_nodes, np.uint64),
                                  _copy(g.length, g.n_nodes, np.uint32), g.seq_len_bases)
        out = BaseGraph(k, seqs, _copy(g.exts, g.n_nodes, np.uint8), _copy(g.data, g.n_nodes, np.uint32),
                        bool(g.stranded))
    finally:
        ctx.lib.dbg_free_graph(ctx.h, C.byref(g))
    return out


def kmer_set_exts(k, key_hi, key_lo, ctx=None):
    """the Exts compress_kmers_no_exts derives for a bare k-mer set (src/compression.rs:626-652): one bit per neighbour whose
    min_rc is in the set"""
    ctx = ctx or default_context()
    key_lo = np.ascontiguousarray(key_lo, np.uint64)
    key_hi = np.ascontiguousarray(key_hi if key_hi is not None else np.zeros(len(key_lo), np.uint64), np.uint64)
    out = np.zeros(max(len(key_lo), 1), np.uint8)
    ctx.check(ctx.lib.dbg_kmer_set_exts(ctx.h, k, len(key_lo), _np_ptr(key_hi), _np_ptr(key_lo), _np_ptr(out)))
    return out[:len(key_lo)]


def compress_kmers_no_exts(stranded, spec, k, key_hi, key_lo, data=None, seed_order=None, ctx=None):
    """compress_kmers_no_exts::<K, D, S> (src/compression.rs:619-659): kmer_exts = [(K, D)] as key columns + data."""
    ctx = ctx or default_context()
    key_lo = np.ascontiguousarray(key_lo, np.uint64)
    n = len(key_lo)
    key_hi = np.ascontiguousarray(key_hi if key_hi is not None else np.zeros(n, np.uint64), np.uint64)
    data = np.ascontiguousarray(data if data is not None else np.zeros(n, np.uint32), np.uint32)
    so = None if seed_order is None else np.ascontiguousarray(seed_order, np.uint64)
    g = _capi.Graph()
    ctx.check(ctx.lib.dbg_compress_kmers_no_exts(ctx.h, k, int(bool(stranded)), spec.kind, n, _np_ptr(key_hi), _np_ptr(key_lo),
                                                 _np_ptr(data), _np_ptr(so), C.byref(g)))
    return _graph_from_c(ctx, g, k)


def upload_seqs(hs, device=0):
    """HostSeqs -> (dbg_seqset of device pointers, tensors that must stay alive)"""
    import torch
    dev = torch.device("cuda", device)
    w = torch.from_numpy(np.concatenate([hs.words, np.zeros(2, np.uint64)]).view(np.int64)).to(dev)
    st = torch.from_numpy(hs.start.view(np.int64)).to(dev)
    ln = torch.from_numpy(hs.length.view(np.int32)).to(dev)
    ex = None if hs.exts is None else torch.from_numpy(hs.exts).to(dev)
    da = None if hs.data is None else torch.from_numpy(hs.data.view({1: np.uint8, 2: np.int16, 4: np.int32}[hs.data_width])).to(dev)
    torch.cuda.synchronize(dev)
    ss = _capi.SeqSet(w.data_ptr(), w.numel(), st.data_ptr(), ln.data_ptr(), None if ex is None else ex.data_ptr(),
                      None if da is None else da.data_ptr(), hs.data_width if da is not None else 0, len(hs.start))
    return ss, (w, st, ln, ex, da)


def filter_kmers_dev(ss, summarizer, stranded, k, report_all_kmers=False, ctx=None):
    """dbg_filter_kmers_dev: device-resident reads in, device-resident table out (a _capi.KmerTable the caller releases
    with ctx.lib.dbg_free_table)."""
    ctx = ctx or default_context()
    fp = _capi.FilterParams(k, int(bool(stranded)), summarizer.kind, summarizer.min_kmer_obs, int(bool(report_all_kmers)), 4)
    t = _capi.KmerTable()
    ctx.check(ctx.lib.dbg_filter_kmers_dev(ctx.h, C.byref(ss), C.byref(fp), C.byref(t)))
    return t


def _classes_from_c(cl):
    """dbg_label_classes -> [tuple of labels] indexed by class id; releases the C arrays"""
    try:
        off = _copy(cl.set_off, cl.n_classes + 1, np.uint64)
        val = _copy(cl.set_val, cl.n_set_val, np.uint32)
        return [tuple(int(x) for x in val[int(off[i]):int(off[i + 1])]) for i in range(int(cl.n_classes))]
    finally:
        _capi.load().dbg_free_label_classes(C.byref(cl))


def compress_table_dev(stranded, spec, table_dev, k, ctx=None):
    """compress_kmers_with_hash on a device-resident table (dbg_compress_table_dev).  CountFilter table: D = count.
    CountFilterSet table: D = class id of the k-mer's label list (what ScmapCompress<Vec<D1>> compares); the returned
    graph carries `classes` (class id -> label tuple) so that a node's Vec<D1> is graph.classes[graph.data[i]]."""
    ctx = ctx or default_context()
    g = _capi.Graph()
    cl = _capi.LabelClasses()
    ctx.check(ctx.lib.dbg_compress_table_dev(ctx.h, k, int(bool(stranded)), spec.kind, C.byref(table_dev), C.byref(g), C.byref(cl)))
    try:
        out = _graph_from_c(ctx, g, k)
    except BaseException:
        ctx.lib.dbg_free_label_classes(C.byref(cl))
        raise
    out.classes = _classes_from_c(cl) if table_dev.set_off else None
    return out


def filter_and_compress_dev(seqs, summarizer, stranded, spec, k, ctx=None):
    """filter_kmers followed by compress_kmers_with_hash with the index kept in HBM between the two calls
    (dbg_filter_kmers_dev -> dbg_compress_table_dev); data = the count column, or the label-list class for
    CountFilterSet.  -> (BaseGraph, n_valid_kmers)"""
    ctx = ctx or default_context()
    hs = seqs if isinstance(seqs, HostSeqs) else HostSeqs.from_tuples(seqs)
    if summarizer.kind == 0 and hs.data is not None:
        hs = HostSeqs(hs.words, hs.start, hs.length, hs.exts)
    ss, keep = upload_seqs(hs, ctx.device)
    t = filter_kmers_dev(ss, summarizer, stranded, k, ctx=ctx)
    try:
        n = int(t.n)
        g = compress_table_dev(stranded, spec, t, k, ctx=ctx)
    finally:
        ctx.lib.dbg_free_table(ctx.h, C.byref(t))
    return g, n


def _graph_to_c(g):
    """BaseGraph -> dbg_graph over the graph's own numpy arrays (kept alive by the returned tuple)."""
    words = np.ascontiguousarray(np.concatenate([g.sequences.words, np.zeros(2, np.uint64)]), np.uint64)
    start = np.ascontiguousarray(g.sequences.start, np.uint64)
    length = np.ascontiguousarray(g.sequences.length, np.uint32)
    exts = np.ascontiguousarray(g.exts, np.uint8)
    data = np.ascontiguousarray(g.data, np.uint32)
    cg = _capi.Graph(len(start), _np_ptr(words), len(g.sequences.words), g.sequences.n_bases, _np_ptr(start), _np_ptr(length),
                     _np_ptr(exts), _np_ptr(data), int(bool(g.stranded)))
    return cg, (words, start, length, exts, data)


def _graph_from_c(ctx, g, k):
    try:
        seqs = PackedDnaStringSet(_copy(g.seq_words, g.n_seq_words, np.uint64), _copy(g.start, g.n_nodes, np.uint64),
                                  _copy(g.length, g.n_nodes, np.uint32), g.seq_len_bases)
        return BaseGraph(k, seqs, _copy(g.exts, g.n_nodes, np.uint8), _copy(g.data, g.n_nodes, np.uint32), bool(g.stranded))
    finally:
        ctx.lib.dbg_free_graph(ctx.h, C.byref(g))


SERDE_JSON, SERDE_BINCODE = 0, 1


def graph_serialize(graph, fmt=SERDE_BINCODE, data_width=4):
    """serde form of a BaseGraph<K, D> (src/graph.rs:43-50): serde_json text or bincode bytes.  Needs no GPU."""
    lib = _capi.load()
    cg, keep = _graph_to_c(graph)
    out, ln = C.c_void_p(), C.c_uint64()
    if lib.dbg_graph_serialize(None, C.byref(cg), fmt, data_width, C.byref(out), C.byref(ln)):
        raise DbgError(lib.dbg_serde_last_error().decode())
    b = C.string_at(out, ln.value)
    lib.dbg_free_bytes(out)
    return b


def graph_deserialize(data, k, fmt=SERDE_BINCODE, data_width=4):
    """BaseGraph from its serde_json / bincode form"""
    lib = _capi.load()
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    g = _capi.Graph()
    if lib.dbg_graph_deserialize(None, buf, len(data), fmt, data_width, C.byref(g)):
        raise DbgError(lib.dbg_serde_last_error().decode())
    try:
        seqs = PackedDnaStringSet(_copy(g.seq_words, g.n_seq_words, np.uint64), _copy(g.start, g.n_nodes, np.uint64),
                                  _copy(g.length, g.n_nodes, np.uint32), g.seq_len_bases)
        return BaseGraph(k, seqs, _copy(g.exts, g.n_nodes, np.uint8), _copy(g.data, g.n_nodes, np.uint32), bool(g.stranded))
    finally:
        lib.dbg_free_graph(None, C.byref(g))


def combine_graphs(graphs, ctx=None):
    """BaseGraph::combine (src/graph.rs:71-100)."""
    ctx = ctx or default_context()
    graphs = list(graphs)
    cs = [_graph_to_c(g) for g in graphs]
    arr = (_capi.Graph * len(cs))(*[c[0] for c in cs])
    out = _capi.Graph()
    ctx.check(ctx.lib.dbg_graph_combine(ctx.h, arr, len(cs), C.byref(out)))
    return _graph_from_c(ctx, out, graphs[0].k if graphs else 0)


def compress_graph(stranded, spec, old_graph, censor_nodes=None, ctx=None):
    """compress_graph (src/compression.rs:338-349): second-stage compaction of a (partly compressed) graph;
    `old_graph.finish()` and both fix_exts passes are part of the call."""
    ctx = ctx or default_context()
    cg, keep = _graph_to_c(old_graph)
    cn = None if censor_nodes is None else np.ascontiguousarray(censor_nodes, np.uint64)
    out = _capi.Graph()
    ctx.check(ctx.lib.dbg_compress_graph(ctx.h, old_graph.k, int(bool(stranded)), spec.kind, C.byref(cg), _np_ptr(cn),
                                         0 if cn is None else len(cn), C.byref(out)))
    return _graph_from_c(ctx, out, old_graph.k)


def graph_edges(graph, ctx=None):
    ctx = ctx or default_context()
    cg, keep = _graph_to_c(graph)
    e = _capi.Edges()
    ctx.check(ctx.lib.dbg_graph_edges(ctx.h, graph.k, C.byref(cg), C.byref(e)))
    n = int(e.n_nodes)
    tgt = np.ctypeslib.as_array(C.cast(e.target, C.POINTER(C.c_uint32)), shape=(max(n * 8, 1),))[:n * 8].copy().reshape(n, 8)
    info = np.ctypeslib.as_array(C.cast(e.info, C.POINTER(C.c_uint8)), shape=(max(n * 8, 1),))[:n * 8].copy().reshape(n, 8)
    ctx.lib.dbg_free_edges(C.byref(e))
    out = []
    for i in range(n):
        sides = []
        for s in range(2):
            sides.append([(int(tgt[i, 4 * s + b]), "R" if info[i, 4 * s + b] & 1 else "L", bool(info[i, 4 * s + b] & 2))
                          for b in range(4) if tgt[i, 4 * s + b] != 0xFFFFFFFF])
        out.append(tuple(sides))
    return out


def graph_to_gfa(graph, ctx=None):
    ctx = ctx or default_context()
    cg, keep = _graph_to_c(graph)
    txt, ln = C.c_void_p(), C.c_uint64()
    ctx.check(ctx.lib.dbg_graph_to_gfa(ctx.h, graph.k, C.byref(cg), C.byref(txt), C.byref(ln)))
    b = C.string_at(txt, ln.value)
    ctx.lib.dbg_free_text(txt)
    return b


def pack_acgt(ascii_bytes, ctx=None):
    """DnaString::from_acgt_bytes (src/dna_string.rs:222-250): -> (packed u64 words, number of non-ACGT characters)."""
    ctx = ctx or default_context()
    a = np.frombuffer(bytes(ascii_bytes), np.uint8) if not isinstance(ascii_bytes, np.ndarray) else np.ascontiguousarray(ascii_bytes, np.uint8)
    words = np.zeros((len(a) + 31) // 32, np.uint64)
    bad = C.c_uint64()
    ctx.check(ctx.lib.dbg_pack_acgt(ctx.h, _np_ptr(a) if len(a) else None, len(a), _np_ptr(words) if len(words) else None, C.byref(bad)))
    return words, int(bad.value)


def pack_acgt_hashn(reads, names, ctx=None):
    """DnaString::from_acgt_bytes_hashn (src/dna_string.rs:255-278) for a batch: reads[i] packed with every non-ACGT
    character replaced by DefaultHasher(names[i], pos) % 4 -> (packed u64 words of the concatenation, start offsets,
    number of replaced characters)."""
    ctx = ctx or default_context()
    if len(reads) != len(names):
        raise ValueError("one name per read")
    seq_off = np.zeros(len(reads) + 1, np.uint64)
    name_off = np.zeros(len(reads) + 1, np.uint64)
    if len(reads):
        seq_off[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
        name_off[1:] = np.cumsum([len(r) for r in names], dtype=np.uint64)
    a = np.frombuffer(b"".join(bytes(r) for r in reads), np.uint8)
    nm = np.frombuffer(b"".join(bytes(r) for r in names), np.uint8)
    words = np.zeros((len(a) + 31) // 32, np.uint64)
    rep = C.c_uint64()
    ctx.check(ctx.lib.dbg_pack_acgt_hashn(ctx.h, _np_ptr(a) if len(a) else None, _np_ptr(seq_off), len(reads),
                                          _np_ptr(nm) if len(nm) else None, _np_ptr(name_off),
                                          _np_ptr(words) if len(words) else None, C.byref(rep)))
    return words, seq_off[:-1].copy(), int(rep.value)


def unpack_acgt(words, first_base, n, ctx=None):
    """DnaString::to_ascii_vec (src/dna_string.rs:297-299) of bases [first_base, first_base + n) -> bytes."""
    ctx = ctx or default_context()
    w = np.ascontiguousarray(words, np.uint64)
    out = np.zeros(n, np.uint8)
    if n:
        ctx.check(ctx.lib.dbg_unpack_acgt(ctx.h, _np_ptr(w), first_base, n, _np_ptr(out)))
    return out.tobytes()


def seqs_from_acgt(reads, exts=None, data=None, ctx=None, names=None):
    """A batch of ASCII reads -> HostSeqs in PackedDnaStringSet layout (dna_string.rs:811-821), packed on the GPU.
    With `names`, non-ACGT characters become the repeatable pseudo-random bases of from_acgt_bytes_hashn instead of A."""
    lens = np.array([len(r) for r in reads], np.uint32)
    start = np.zeros(len(reads), np.uint64)
    if len(reads):
        start[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
    if names is None:
        words, _ = pack_acgt(b"".join(bytes(r) for r in reads), ctx)
    else:
        words, _, _ = pack_acgt_hashn(reads, names, ctx)
    d = None if data is None else np.ascontiguousarray(data, np.uint8)
    e = None if exts is None else np.ascontiguousarray(exts, np.uint8)
    return HostSeqs(words, start, lens, e, d, 0 if d is None else 1)


def _censor(stranded, table, all_kmers, sharded, ctx):
    ctx = ctx or default_context()
    t = _capi.KmerTable()
    key_hi = np.ascontiguousarray(table.key_hi, np.uint64)
    key_lo = np.ascontiguousarray(table.key_lo, np.uint64)
    exts = np.ascontiguousarray(table.exts, np.uint8).copy()
    t.n = len(key_lo)
    t.key_hi, t.key_lo, t.exts = _np_ptr(key_hi), _np_ptr(key_lo), _np_ptr(exts)
    if sharded:
        ah = np.array([v >> 64 for v in all_kmers], dtype=np.uint64)
        al = np.array([v & M64 for v in all_kmers], dtype=np.uint64)
        t.n_all, t.all_hi, t.all_lo = len(al), _np_ptr(ah), _np_ptr(al)
    ctx.check(ctx.lib.dbg_remove_censored_exts(ctx.h, table.k, int(bool(stranded)), C.byref(t), int(sharded)))
    table.exts = exts
    return table


def remove_censored_exts_sharded(stranded, table, all_kmers, ctx=None):
    """src/filter.rs:238-276 on the sorted table (in place on table.exts)."""
    return _censor(stranded, table, all_kmers, True, ctx)


def remove_censored_exts(stranded, table, ctx=None):
    """src/filter.rs:280-306."""
    return _censor(stranded, table, None, False, ctx)


def synth_params(n_reads, read_len=150, genome_len=0, error_rate=0.001, stranded=False, n_colours=4,
                 first_read=0, genome_seed=0xDB60001, read_seed=0xDB60002):
    return _capi.SynthParams(n_reads, read_len, genome_len, genome_seed, read_seed, error_rate, int(stranded),
                             n_colours, first_read)


def synth_reads_host(**kw):
    """Deterministic synthetic reads generated on the host (no GPU): -> HostSeqs."""
    lib = _capi.load()
    p = synth_params(**kw)
    nw = lib.dbg_synth_words(C.byref(p))
    words = np.zeros(nw, np.uint64)
    start = np.zeros(p.n_reads, np.uint64)
    length = np.zeros(p.n_reads, np.uint32)
    data = np.zeros(p.n_reads, np.uint8) if p.n_colours else None
    r = lib.dbg_synth_reads_host(C.byref(p), _np_ptr(words), _np_ptr(start), _np_ptr(length), _np_ptr(data))
    if r:
        raise DbgError("synthetic genome shorter than a read")
    return HostSeqs(words, start, length, None, data, 1 if p.n_colours else 0)
