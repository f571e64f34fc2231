"""ctypes driver for the CPU oracle (oracle/_build/liboracle.so).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package.
"""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "liboracle.so")

COUNT_FILTER, COUNT_FILTER_SET = 0, 1
SPEC_SAT_ADD, SPEC_ADD_MOD, SPEC_MAX, SPEC_SCMAP_EQ, SPEC_WRAP_ADD = 0, 1, 2, 3, 4
LEFT, RIGHT = 0, 1
M64 = (1 << 64) - 1


def build(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("dbg_oracle.cpp", "dbg_oracle_c.cpp", "dbg_oracle.hpp")]
    stale = (not os.path.exists(LIB_PATH)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_last_error.restype = C.c_char_p
        for f in ("orc_filter_kmers", "orc_compress_kmers", "orc_compress_kmers_no_exts", "orc_graph_from_arrays", "orc_graph_combine",
                  "orc_compress_graph"):
            getattr(_lib, f).restype = C.c_void_p
        for f in ("orc_table_len", "orc_table_all_len", "orc_table_setval_len", "orc_graph_len",
                  "orc_graph_n_words", "orc_graph_n_bases", "orc_dnastring_pack", "orc_dnastring_push", "orc_graph_gfa"):
            getattr(_lib, f).restype = C.c_uint64
        _lib.orc_table_passes.restype = C.c_uint32
        _lib.orc_msp_scan.restype = C.c_int64
        _lib.orc_msp_sequence.restype = C.c_int64
        _lib.orc_time_filter_kmers.restype = C.c_double
        _lib.orc_siphash.restype = C.c_uint64
        _lib.orc_time_filter_kmers_sharded_mt.restype = C.c_double
        _lib.orc_exts_rc.restype = C.c_uint8
        _lib.orc_exts_complement.restype = C.c_uint8
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _err():
    return RuntimeError("reference would panic: " + lib().orc_last_error().decode())


# ---------------------------------------------------------------- building blocks
def kmer_from_ascii(s):
    if isinstance(s, str):
        s = s.encode()
    hi, lo = C.c_uint64(), C.c_uint64()
    lib().orc_kmer_from_ascii(len(s), s, C.byref(hi), C.byref(lo))
    return (hi.value << 64) | lo.value


def kmer_to_ascii(k, v):
    buf = C.create_string_buffer(k + 1)
    lib().orc_kmer_to_ascii(k, C.c_uint64(v >> 64), C.c_uint64(v & M64), buf)
    return buf.value.decode()


def _k2(fn, k, v, *extra):
    hi, lo = C.c_uint64(), C.c_uint64()
    r = fn(k, C.c_uint64(v >> 64), C.c_uint64(v & M64), *extra, C.byref(hi), C.byref(lo))
    return (hi.value << 64) | lo.value, r


def kmer_rc(k, v):
    return _k2(lib().orc_kmer_rc, k, v)[0]


def kmer_extend(k, v, base, direction):
    return _k2(lib().orc_kmer_extend, k, v, base, direction)[0]


def kmer_min_rc_flip(k, v):
    out, flip = _k2(lib().orc_kmer_min_rc_flip, k, v)
    return out, bool(flip)


def kmer_min_rc(k, v):
    return kmer_min_rc_flip(k, v)[0]


def kmer_is_palindrome(k, v):
    return bool(lib().orc_kmer_is_palindrome(k, C.c_uint64(v >> 64), C.c_uint64(v & M64)))


def exts_rc(v):
    return lib().orc_exts_rc(C.c_uint8(v))


def dnastring_pack(bases, push=False):
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    cap = (len(bases) + 31) // 32 + 1
    out = np.zeros(cap, dtype=np.uint64)
    fn = lib().orc_dnastring_push if push else lib().orc_dnastring_pack
    nw = fn(_p(bases), C.c_uint64(len(bases)), _p(out), C.c_uint64(cap))
    return out[:nw].copy()


def seq_get_kmer(words, start, length, pos, k):
    hi, lo = C.c_uint64(), C.c_uint64()
    words = np.ascontiguousarray(words, dtype=np.uint64)
    lib().orc_seq_get_kmer(_p(words), C.c_uint64(start), C.c_uint64(length), C.c_uint64(pos), k,
                           C.byref(hi), C.byref(lo))
    return (hi.value << 64) | lo.value


# ---------------------------------------------------------------- sequence sets
def pack_bases(b):
    """numpy packer: base i -> word i/32, bits [63-2(i%32), 62-2(i%32)] (dna_string.rs:383-399)."""
    b = np.asarray(b, dtype=np.uint64)
    n = len(b)
    nw = (n + 31) // 32
    pad = np.zeros(nw * 32, dtype=np.uint64)
    pad[:n] = b
    pad = pad.reshape(nw, 32)
    shifts = (62 - 2 * np.arange(32)).astype(np.uint64)
    return np.bitwise_or.reduce(pad << shifts, axis=1).astype(np.uint64)


def unpack_bases(words, start, length):
    idx = np.arange(start, start + length, dtype=np.uint64)
    w = np.asarray(words, dtype=np.uint64)[(idx >> np.uint64(5)).astype(np.int64)]
    sh = (np.uint64(62) - np.uint64(2) * (idx & np.uint64(31))).astype(np.uint64)
    return ((w >> sh) & np.uint64(3)).astype(np.uint8)


class SeqSet:
    """PackedDnaStringSet layout + per-sequence Exts byte + per-sequence data (u32-widened)."""

    def __init__(self, words, start, length, exts=None, data=None, sizeof_d1=0):
        self.words = np.ascontiguousarray(words, dtype=np.uint64)
        self.start = np.ascontiguousarray(start, dtype=np.uint64)
        self.length = np.ascontiguousarray(length, dtype=np.uint32)
        n = len(self.start)
        self.exts = np.zeros(n, dtype=np.uint8) if exts is None else np.ascontiguousarray(exts, dtype=np.uint8)
        self.data = None if data is None else np.ascontiguousarray(data, dtype=np.uint32)
        self.sizeof_d1 = sizeof_d1
        self.n = n

    @staticmethod
    def from_byte_seqs(seqs, exts=None, data=None, sizeof_d1=0):
        """Concatenate 0-3 byte sequences back-to-back (PackedDnaStringSet::add, dna_string.rs:811-821)."""
        starts, lens, allb, pos = [], [], [], 0
        for s in seqs:
            s = np.asarray(s, dtype=np.uint8)
            starts.append(pos)
            lens.append(len(s))
            allb.append(s)
            pos += len(s)
        cat = np.concatenate(allb) if allb else np.zeros(0, dtype=np.uint8)
        words = pack_bases(cat)
        # two words of tail padding so 3-word k-mer window loads never run off the end
        words = np.concatenate([words, np.zeros(2, dtype=np.uint64)])
        return SeqSet(words, starts, lens, exts, data, sizeof_d1)

    def bases(self, i):
        return unpack_bases(self.words, int(self.start[i]), int(self.length[i]))


# ---------------------------------------------------------------- filter_kmers
class Table:
    def __len__(self):
        return int(self.n)

    def key(self, i):
        return (int(self.key_hi[i]) << 64) | int(self.key_lo[i])

    def keys(self):
        return [(int(h) << 64) | int(l) for h, l in zip(self.key_hi, self.key_lo)]


def filter_kmers(ss, k, summarizer=COUNT_FILTER, min_obs=1, stranded=False, report_all=False, memory_size=4):
    L = lib()
    h = L.orc_filter_kmers(_p(ss.words), _p(ss.start), _p(ss.length), _p(ss.exts), _p(ss.data),
                           C.c_uint64(ss.n), C.c_uint32(ss.sizeof_d1), k, summarizer, C.c_uint64(min_obs),
                           int(stranded), int(report_all), C.c_uint64(memory_size))
    if not h:
        raise _err()
    h = C.c_void_p(h)
    n, na, ns = L.orc_table_len(h), L.orc_table_all_len(h), L.orc_table_setval_len(h)
    t = Table()
    t.n = n
    t.key_hi = np.zeros(n, np.uint64)
    t.key_lo = np.zeros(n, np.uint64)
    t.exts = np.zeros(n, np.uint8)
    t.count = np.zeros(n, np.uint16)
    t.set_off = np.zeros(n + 1, np.uint64)
    t.set_val = np.zeros(ns, np.uint32)
    t.all_hi = np.zeros(na, np.uint64)
    t.all_lo = np.zeros(na, np.uint64)
    t.passes = L.orc_table_passes(h)
    is_set = summarizer == COUNT_FILTER_SET
    L.orc_table_copy(h, _p(t.key_hi), _p(t.key_lo), _p(t.exts), _p(t.count),
                     _p(t.set_off) if is_set else None, _p(t.set_val) if is_set else None,
                     _p(t.all_hi), _p(t.all_lo))
    L.orc_table_free(h)
    return t


def remove_censored_exts(k, stranded, key_hi, key_lo, exts, all_hi=None, all_lo=None, sharded=False):
    e = np.ascontiguousarray(exts, dtype=np.uint8).copy()
    n_all = 0 if all_lo is None else len(all_lo)
    lib().orc_remove_censored_exts(k, int(stranded), C.c_uint64(len(key_lo)),
                                   _p(np.ascontiguousarray(key_hi, dtype=np.uint64)),
                                   _p(np.ascontiguousarray(key_lo, dtype=np.uint64)), _p(e), C.c_uint64(n_all),
                                   _p(None if all_hi is None else np.ascontiguousarray(all_hi, dtype=np.uint64)),
                                   _p(None if all_lo is None else np.ascontiguousarray(all_lo, dtype=np.uint64)),
                                   int(sharded))
    return e


def time_filter_kmers(ss, k, summarizer, min_obs, stranded, memory_size=4):
    nv = C.c_uint64()
    t = lib().orc_time_filter_kmers(_p(ss.words), _p(ss.start), _p(ss.length), _p(ss.exts), _p(ss.data),
                                    C.c_uint64(ss.n), C.c_uint32(ss.sizeof_d1), k, summarizer,
                                    C.c_uint64(min_obs), int(stranded), C.c_uint64(memory_size), C.byref(nv))
    return t, nv.value


_PH1 = C.c_double()      # seconds of the msp_sequence phase of the last sharded run


def time_filter_kmers_sharded_mt(ss, k, summarizer, min_obs, stranded, n_threads, n_shards=None, p=8):
    """msp_sequence -> shards -> filter_kmers per shard on n_threads host threads (the pipeline of the reference's
    test.rs:418-504, which is how callers parallelise the crate); returns (seconds, total valid k-mers)"""
    nv = C.c_uint64()
    n_shards = n_shards or 8 * n_threads
    t = lib().orc_time_filter_kmers_sharded_mt(_p(ss.words), _p(ss.start), _p(ss.length), _p(ss.data), C.c_uint64(ss.n),
                                               C.c_uint32(ss.sizeof_d1), k, p, summarizer, C.c_uint64(min_obs),
                                               int(stranded), C.c_uint32(n_threads), C.c_uint32(n_shards), C.byref(nv), C.byref(_PH1))
    if t < 0:
        raise _err()
    return t, nv.value


# ---------------------------------------------------------------- MSP
def msp_scan(seq, k, p, perm=None, rc=False, score_mode=0):
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    cap = max(len(seq), 1)
    mi = np.zeros(cap, np.uint64)
    st = np.zeros(cap, np.uint32)
    ln = np.zeros(cap, np.uint16)
    mp = np.zeros(cap, np.uint32)
    perm_a = None if perm is None else np.ascontiguousarray(perm, dtype=np.uint64)
    n = lib().orc_msp_scan(_p(seq), C.c_uint64(len(seq)), k, p, _p(perm_a), int(rc), score_mode,
                           _p(mi), _p(st), _p(ln), _p(mp), C.c_uint64(cap))
    if n < 0:
        raise _err()
    return mi[:n], st[:n], ln[:n], mp[:n]


def msp_sequence(seq, k, p, perm=None, rc=True, vmer_max_len=1 << 48, lmer_words=0):
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    cap = max(len(seq), 1)
    bu = np.zeros(cap, np.uint32)
    ex = np.zeros(cap, np.uint8)
    st = np.zeros(cap, np.uint32)
    ln = np.zeros(cap, np.uint16)
    lw = np.zeros(cap * max(lmer_words, 1), np.uint64)
    perm_a = None if perm is None else np.ascontiguousarray(perm, dtype=np.uint64)
    n = lib().orc_msp_sequence(_p(seq), C.c_uint64(len(seq)), k, p, _p(perm_a), int(rc), C.c_uint64(vmer_max_len),
                               _p(bu), _p(ex), _p(st), _p(ln), _p(lw) if lmer_words else None, lmer_words,
                               C.c_uint64(cap))
    if n < 0:
        raise _err()
    out = (bu[:n].copy(), ex[:n].copy(), st[:n].copy(), ln[:n].copy())
    if lmer_words:
        out = out + (lw[:n * lmer_words].reshape(n, lmer_words).copy(),)
    return out


# ---------------------------------------------------------------- compress / graph
class Graph:
    def __init__(self, handle, k):
        self.h = C.c_void_p(handle)
        self.k = k

    def __del__(self):
        try:
            lib().orc_graph_free(self.h)
        except Exception:
            pass

    def arrays(self):
        L = lib()
        n, nw = L.orc_graph_len(self.h), L.orc_graph_n_words(self.h)
        words = np.zeros(nw + 2, np.uint64)
        start = np.zeros(n, np.uint64)
        length = np.zeros(n, np.uint32)
        exts = np.zeros(n, np.uint8)
        data = np.zeros(n, np.uint32)
        L.orc_graph_copy(self.h, _p(words), _p(start), _p(length), _p(exts), _p(data))
        return dict(words=words, n_words=nw, start=start, length=length, exts=exts, data=data,
                    n_bases=L.orc_graph_n_bases(self.h))

    def __len__(self):
        return lib().orc_graph_len(self.h)

    def is_compressed(self, spec):
        a, b = C.c_uint64(), C.c_uint64()
        ok = lib().orc_graph_is_compressed(self.h, spec, C.byref(a), C.byref(b))
        return None if ok else (a.value, b.value)

    def finish(self):
        lib().orc_graph_finish(self.h)
        return self

    def write_gfa(self):
        """DebruijnGraph::write_gfa (graph.rs:603-616) -> bytes"""
        L = lib()
        n = L.orc_graph_gfa(self.h, None, C.c_uint64(0))
        buf = C.create_string_buffer(max(n, 1))
        L.orc_graph_gfa(self.h, buf, C.c_uint64(n))
        return buf.raw[:n]

    def compress_graph(self, stranded, spec, censor=None):
        c = None if censor is None else np.ascontiguousarray(censor, dtype=np.uint64)
        h = lib().orc_compress_graph(self.h, int(stranded), spec, _p(c), C.c_uint64(0 if c is None else len(c)))
        if not h:
            raise _err()
        return Graph(h, self.k)


def compress_kmers(k, stranded, spec, key_hi, key_lo, exts, data=None, seed_order=None):
    key_lo = np.ascontiguousarray(key_lo, dtype=np.uint64)
    key_hi = np.zeros_like(key_lo) if key_hi is None else np.ascontiguousarray(key_hi, dtype=np.uint64)
    exts = np.ascontiguousarray(exts, dtype=np.uint8)
    d = None if data is None else np.ascontiguousarray(data, dtype=np.uint32)
    so = None if seed_order is None else np.ascontiguousarray(seed_order, dtype=np.uint64)
    h = lib().orc_compress_kmers(k, int(stranded), spec, C.c_uint64(len(key_lo)), _p(key_hi), _p(key_lo),
                                 _p(exts), _p(d), _p(so))
    if not h:
        raise _err()
    return Graph(h, k)


def compress_kmers_no_exts(k, stranded, spec, key_hi, key_lo, data=None, seed_order=None):
    """compress_kmers_no_exts (compression.rs:619-659) -> (Graph, the Exts it derived)"""
    key_lo = np.ascontiguousarray(key_lo, dtype=np.uint64)
    key_hi = np.zeros_like(key_lo) if key_hi is None else np.ascontiguousarray(key_hi, dtype=np.uint64)
    d = None if data is None else np.ascontiguousarray(data, dtype=np.uint32)
    so = None if seed_order is None else np.ascontiguousarray(seed_order, dtype=np.uint64)
    ex = np.zeros(max(len(key_lo), 1), np.uint8)
    h = lib().orc_compress_kmers_no_exts(k, int(stranded), spec, C.c_uint64(len(key_lo)), _p(key_hi), _p(key_lo), _p(d), _p(so), _p(ex))
    if not h:
        raise _err()
    return Graph(h, k), ex[:len(key_lo)]


def graph_from_arrays(k, stranded, words, start, length, exts, data=None):
    words = np.ascontiguousarray(words, np.uint64)
    start = np.ascontiguousarray(start, np.uint64)
    length = np.ascontiguousarray(length, np.uint32)
    exts = np.ascontiguousarray(exts, np.uint8)
    d = None if data is None else np.ascontiguousarray(data, np.uint32)
    h = lib().orc_graph_from_arrays(k, int(stranded), C.c_uint64(len(start)), _p(words), _p(start), _p(length),
                                    _p(exts), _p(d))
    return Graph(h, k)


def pack_acgt(ascii_bytes):
    """DnaString::from_acgt_bytes (dna_string.rs:222-250) -> storage words"""
    a = np.frombuffer(bytes(ascii_bytes), np.uint8)
    words = np.zeros((len(a) + 31) // 32, np.uint64)
    if len(a):
        lib().orc_pack_acgt(_p(a), C.c_uint64(len(a)), _p(words))
    return words


def pack_acgt_hashn(ascii_bytes, name):
    a = np.frombuffer(bytes(ascii_bytes), np.uint8)
    nm = np.frombuffer(bytes(name), np.uint8)
    words = np.zeros((len(a) + 31) // 32, np.uint64)
    lib().orc_pack_acgt_hashn(_p(a) if len(a) else None, C.c_uint64(len(a)), _p(nm) if len(nm) else None,
                              C.c_uint64(len(nm)), _p(words) if len(words) else None)
    return words


def siphash(c, d, k0, k1, msg):
    m = np.frombuffer(bytes(msg), np.uint8)
    return int(lib().orc_siphash(c, d, C.c_uint64(k0), C.c_uint64(k1), _p(m) if len(m) else None, C.c_uint64(len(m))))


def unpack_acgt(words, first_base, n):
    w = np.ascontiguousarray(words, np.uint64)
    out = np.zeros(n, np.uint8)
    if n:
        lib().orc_unpack_acgt(_p(w), C.c_uint64(first_base), C.c_uint64(n), _p(out))
    return out.tobytes()


def graph_combine(graphs):
    arr = (C.c_void_p * len(graphs))(*[g.h for g in graphs])
    h = lib().orc_graph_combine(arr, C.c_uint64(len(graphs)))
    if not h:
        raise _err()
    return Graph(h, graphs[0].k)
