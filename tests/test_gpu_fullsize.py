"""GPU, BASELINE.json full sizes -- configs[1]: 10^8 reads x 150 bp, k = 47, non-stranded; config 4's per-GPU share:
1.25 * 10^8 reads at k = 63 (the two-word Kmer: 4-word records, three-array sort form, 20-byte outputs); config 5's share:
7.5 * 10^7 reads at k = 51 -- size-independent properties of filter_kmers that hold for the reference by construction
(src/filter.rs:139-231):

  * the output keys are strictly ascending (the order handed to BoomHashMap2::new, filter.rs:227-230) and canonical
    (key <= rc(key), lib.rs:224-231);
  * with CountFilter(1) every k-mer instance is counted exactly once: sum(count) == N == sum(len - k + 1)
    (no key reaches the u16 saturation point at 30x coverage of a random genome, checked);
  * linearity over a partition of the reads ("checksum of checksums"): the multiset checksum
    sum(count * h(key)) mod 2^64 of the whole input equals the sum of the checksums of its two halves, and
    the number of distinct keys is sub-additive;
  * CountFilterSet(2) and CountFilter(2) select the same key set, and the colour sets are non-empty subsets of
    the C = 4 colours;
  * a 10^5-read prefix of the same stream is bit-exact against the CPU oracle (SURVEY.md section 8d, C3 note).

Everything is computed on the device (torch views over the library's device arrays); nothing reads the
reference tree."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
from pkg import dbg

pytestmark = pytest.mark.gpu

L = 150
_SCALE = float(os.environ.get("DBG_FULLSIZE_SCALE", 1.0))          # < 1 for a quick look on a shared box
CASES = [(47, int(100_000_000 * _SCALE)), (63, int(125_000_000 * _SCALE)), (51, int(75_000_000 * _SCALE))]
CASE_IDS = ["k47-1e8", "k63-1.25e8", "k51-7.5e7"]


class _DevArr:
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def dev_view(ptr, n, typestr="<i8"):
    import torch
    if not ptr or n == 0:
        return torch.zeros(0, dtype=torch.int64, device="cuda")
    return torch.as_tensor(_DevArr(C.cast(ptr, C.c_void_p).value, n, typestr), device="cuda")


@pytest.fixture(scope="module", params=CASES, ids=CASE_IDS)
def env(request):
    """one (k, reads) case: the reads are generated once in HBM; pytest runs all tests of a case before the next case"""
    import importlib
    import torch
    K, N_READS = request.param
    capi = importlib.import_module("rust-debruijn_amd._capi")
    ctx = dbg.Context(0)
    lib = ctx.lib
    dev = torch.device("cuda", 0)
    p = dbg.synth_params(n_reads=N_READS, read_len=L, genome_len=N_READS * L // 30, error_rate=0.001,
                         stranded=False, n_colours=4, first_read=0)
    nw = lib.dbg_synth_words(C.byref(p))
    words = torch.empty(nw, dtype=torch.int64, device=dev)
    start = torch.empty(N_READS, dtype=torch.int64, device=dev)
    length = torch.empty(N_READS, dtype=torch.int32, device=dev)
    colour = torch.empty(N_READS, dtype=torch.uint8, device=dev)
    ctx.check(lib.dbg_synth_reads_dev(ctx.h, C.byref(p), words.data_ptr(), start.data_ptr(), length.data_ptr(), colour.data_ptr()))
    e = dict(capi=capi, ctx=ctx, lib=lib, torch=torch, p=p, nw=nw, words=words, start=start, length=length, colour=colour,
             k=K, n_reads=N_READS)
    yield e
    ctx.close()
    e.clear()
    del words, start, length, colour
    torch.cuda.empty_cache()


only_c2 = pytest.mark.parametrize("env", CASES[:1], ids=CASE_IDS[:1], indirect=True)


def run_filter(e, first, n, summarizer, min_obs, report_all=0):
    capi, ctx, lib = e["capi"], e["ctx"], e["lib"]
    is_set = summarizer == 1
    ss = capi.SeqSet(e["words"].data_ptr(), e["nw"], e["start"][first:].data_ptr(), e["length"][first:].data_ptr(), None,
                     e["colour"][first:].data_ptr() if is_set else None, 1 if is_set else 0, n)
    fp = capi.FilterParams(e["k"], 0, summarizer, min_obs, report_all, 4)
    t = capi.KmerTable()
    ctx.check(lib.dbg_filter_kmers_dev(ctx.h, C.byref(ss), C.byref(fp), C.byref(t)))
    return t


def mix64(torch, hi, lo):
    """64-bit multiplicative checksum term of a key; int64 arithmetic wraps mod 2^64 on the device."""
    a = lo * -7046029254386353131 + hi * -4417276706812531889          # 0x9E3779B97F4A7C15, 0xC2B2AE3D27D4EB4F
    a = a ^ ((a >> 29) & 0x7FFFFFFFF)
    return a * -4658895280553007687                                     # 0xBF58476D1CE4E5B9


def table_stats(e, t):
    torch = e["torch"]
    hi, lo = dev_view(t.key_hi, t.n), dev_view(t.key_lo, t.n)
    cnt = dev_view(t.count, t.n, "<u2").to(torch.int64) if t.count else None
    out = dict(n=int(t.n), n_inst=int(t.n_kmer_instances))
    # strictly ascending as unsigned (hi, lo): flip the sign bit of lo for an unsigned compare
    if t.n > 1:
        ulo = lo ^ (-9223372036854775808)
        ok = (hi[1:] > hi[:-1]) | ((hi[1:] == hi[:-1]) & (ulo[1:] > ulo[:-1]))
        out["ascending"] = bool(ok.all().item())
        del ulo, ok
    else:
        out["ascending"] = True
    if cnt is not None:
        out["sum_count"] = int(cnt.sum().item())
        out["max_count"] = int(cnt.max().item())
        out["checksum"] = int((mix64(torch, hi, lo) * cnt).sum().item())
    out["keysum"] = int(mix64(torch, hi, lo).sum().item())
    return out


def host_sample(e, t, n_sample, seed):
    """random sample of table rows copied to the host: (hi, lo) as Python ints"""
    torch = e["torch"]
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    idx = torch.randint(0, int(t.n), (n_sample,), device="cuda", generator=g)
    hi = dev_view(t.key_hi, t.n)[idx].cpu().numpy().astype(np.uint64)
    lo = dev_view(t.key_lo, t.n)[idx].cpu().numpy().astype(np.uint64)
    return hi, lo


def test_fullsize_count_properties(env):
    e = env
    lib, ctx = e["lib"], e["ctx"]
    K, N_READS = e["k"], e["n_reads"]
    n_expected = N_READS * (L - K + 1)
    half = N_READS // 2

    whole = run_filter(e, 0, N_READS, 0, 1)
    sw = table_stats(e, whole)
    hi, lo = host_sample(e, whole, 200_000, 1)
    lib.dbg_free_table(ctx.h, C.byref(whole))
    assert sw["n_inst"] == n_expected
    assert sw["ascending"]
    assert sw["max_count"] < 65535                       # no saturation, so counts are exactly additive
    assert sw["sum_count"] == n_expected                 # every instance counted once
    # canonical keys (lib.rs:224-231) and k-mer width (kmer.rs:429-437) on a random sample
    for h, l in zip(hi[:20000].tolist(), lo[:20000].tolist()):
        v = (h << 64) | l
        assert v >> (2 * K) == 0
        assert v <= O.kmer_rc(K, v)

    a = run_filter(e, 0, half, 0, 1)
    sa = table_stats(e, a)
    lib.dbg_free_table(ctx.h, C.byref(a))
    b = run_filter(e, half, N_READS - half, 0, 1)
    sb = table_stats(e, b)
    lib.dbg_free_table(ctx.h, C.byref(b))
    assert sa["ascending"] and sb["ascending"]
    assert sa["sum_count"] + sb["sum_count"] == n_expected
    M = (1 << 64) - 1
    assert (sa["checksum"] + sb["checksum"]) & M == sw["checksum"] & M      # linearity over a partition of the reads
    assert max(sa["n"], sb["n"]) <= sw["n"] <= sa["n"] + sb["n"]


def test_fullsize_filter_set_vs_count(env):
    """configs[1] proper: CountFilterSet<u8>(2) -- same valid key set as CountFilter(2); colour sets are sorted,
    deduplicated, non-empty subsets of {0..3} (filter.rs:90-99)."""
    e = env
    torch, lib, ctx = e["torch"], e["lib"], e["ctx"]
    K, N_READS = e["k"], e["n_reads"]
    tc = run_filter(e, 0, N_READS, 0, 2)
    sc = table_stats(e, tc)
    cmin = int(dev_view(tc.count, tc.n, "<u2").to(torch.int64).min().item())
    lib.dbg_free_table(ctx.h, C.byref(tc))
    ts = run_filter(e, 0, N_READS, 1, 2)
    ss = table_stats(e, ts)
    off = dev_view(ts.set_off, ts.n + 1)
    val = dev_view(ts.set_val, ts.n_set_val, "<u4").to(torch.int64) if ts.n_set_val else None
    sizes = off[1:] - off[:-1]
    assert int(off[0].item()) == 0 and int(off[-1].item()) == int(ts.n_set_val)
    assert int(sizes.min().item()) >= 1 and int(sizes.max().item()) <= 4
    assert int(val.min().item()) >= 0 and int(val.max().item()) <= 3
    # inside one set the labels are strictly ascending: a non-ascending step may only happen at a set boundary
    is_boundary = torch.zeros(int(ts.n_set_val) + 1, dtype=torch.bool, device="cuda")
    is_boundary[off] = True
    step = 1 << 27                       # in slices: nonzero() over > 2^30 elements asked for an absurd allocation on this torch build
    for a in range(1, int(ts.n_set_val), step):
        b = min(a + step, int(ts.n_set_val))
        nonasc = (val[a:b] <= val[a - 1:b - 1]).nonzero().flatten() + a
        assert bool(is_boundary[nonasc].all().item())
    del is_boundary
    lib.dbg_free_table(ctx.h, C.byref(ts))
    assert cmin >= 2
    assert sc["ascending"] and ss["ascending"]
    assert sc["n"] == ss["n"] and sc["keysum"] == ss["keysum"]
    assert sc["n_inst"] == ss["n_inst"] == N_READS * (L - K + 1)


def test_fullsize_prefix_bit_exact(env):
    """A prefix of the full-size stream (sparse coverage: almost every k-mer is a singleton) against the oracle."""
    e = env
    capi, lib, ctx = e["capi"], e["lib"], e["ctx"]
    K, N_READS = e["k"], e["n_reads"]
    m = min(100_000, N_READS)
    t = run_filter(e, 0, m, 1, 1)
    h = capi.KmerTable()
    ctx.check(lib.dbg_table_to_host(ctx.h, C.byref(t), C.byref(h)))
    lib.dbg_free_table(ctx.h, C.byref(t))
    n = h.n
    as_np = lambda p, ct, cnt: np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(max(cnt, 1),))[:cnt].copy()
    g_hi, g_lo = as_np(h.key_hi, C.c_uint64, n), as_np(h.key_lo, C.c_uint64, n)
    g_ex = as_np(h.exts, C.c_uint8, n)
    g_off, g_val = as_np(h.set_off, C.c_uint64, n + 1), as_np(h.set_val, C.c_uint32, h.n_set_val)
    lib.dbg_free_table(ctx.h, C.byref(h))
    # the same reads from the bit-identical host generator
    hs = dbg.synth_reads_host(n_reads=m, read_len=L, genome_len=N_READS * L // 30, error_rate=0.001, stranded=False, n_colours=4)
    want = O.filter_kmers(O.SeqSet(hs.words, hs.start, hs.length, None, hs.data, 1), K, O.COUNT_FILTER_SET, 1, stranded=False)
    assert n == want.n
    assert np.array_equal(g_hi, want.key_hi) and np.array_equal(g_lo, want.key_lo) and np.array_equal(g_ex, want.exts)
    assert np.array_equal(g_off, want.set_off) and np.array_equal(g_val, want.set_val)


@only_c2
def test_fullsize_report_all_kmers(env):
    """report_all_kmers at full size (filter.rs:208-212): the all-list is strictly ascending, contains the valid table, and
    with CountFilter(1) -- every distinct k-mer valid -- the two lists are identical."""
    e = env
    torch, lib, ctx = e["torch"], e["lib"], e["ctx"]
    K, N_READS = e["k"], e["n_reads"]
    torch.cuda.empty_cache()             # the earlier tests' temporaries sit in torch's caching allocator; this run wants ~150 GB
    t = run_filter(e, 0, N_READS, 0, 1, report_all=1)
    assert t.n == t.n_all > 0
    same = bool((dev_view(t.key_lo, t.n) == dev_view(t.all_lo, t.n_all)).all().item()) and \
        bool((dev_view(t.key_hi, t.n) == dev_view(t.all_hi, t.n_all)).all().item())
    n1 = int(t.n)
    lib.dbg_free_table(ctx.h, C.byref(t))
    assert same
    t2 = run_filter(e, 0, N_READS, 0, 2, report_all=1)
    hi, lo = dev_view(t2.all_hi, t2.n_all), dev_view(t2.all_lo, t2.n_all)
    ulo = lo ^ (-9223372036854775808)
    asc = bool(((hi[1:] > hi[:-1]) | ((hi[1:] == hi[:-1]) & (ulo[1:] > ulo[:-1]))).all().item())
    n_all2, n_valid2 = int(t2.n_all), int(t2.n)
    lib.dbg_free_table(ctx.h, C.byref(t2))
    assert asc and n_all2 == n1 and n_valid2 < n_all2                 # the distinct set does not depend on min_obs


def _lookup(torch, hi_t, lo_t, qhi, qlo):
    """positions of the (hi, lo) queries in the ascending device table, -1 where absent (hi holds 2k-64 = 30 bits at k = 47)"""
    q_hi = torch.tensor(np.array(qhi, dtype=np.int64), device="cuda")
    q_lo = torch.tensor(np.array(qlo, dtype=np.uint64).view(np.int64), device="cuda")
    a = torch.searchsorted(hi_t, q_hi, right=False)
    b = torch.searchsorted(hi_t, q_hi, right=True)
    out = torch.full_like(a, -1)
    width = int((b - a).max().item()) if len(qhi) else 0
    for j in range(width):                                 # equal-hi runs are a handful of keys long
        pos = torch.clamp(a + j, max=hi_t.numel() - 1)
        hit = (a + j < b) & (lo_t[pos] == q_lo)
        out = torch.where(hit, pos, out)
    return out.cpu().numpy()


@pytest.mark.parametrize("censored", [False, True], ids=["as-filtered", "censored"])
@pytest.mark.parametrize("env", [CASES[0], CASES[1]], ids=[CASE_IDS[0], CASE_IDS[1]], indirect=True)
def test_fullsize_compress_invariants(env, censored):
    """BASELINE config 3 at full size: CountFilter(2) table [-> remove_censored_exts, the pipeline real callers run,
    filter.rs:233-306 / test.rs:236-254] -> compress_kmers_with_hash on the device (index left in HBM).
    Size-independent properties of the reference's result (compression.rs:355-583; the tests of test.rs:236-349 check the
    same on small graphs): the nodes partition the valid k-mers (the k-mer counts add up, sampled nodes consist of table
    k-mers only and no k-mer of two different sampled nodes coincides), a node's data is the saturating sum of its
    k-mers' counts, its Exts are the outward Exts of its end k-mers, inside a node every k-mer has exactly the one
    neighbour the path takes, and no sampled node could have been extended (BaseGraph::is_compressed, graph.rs:296-334,
    at the k-mer level: an end with exactly one extension leads to a k-mer that does not have exactly one extension back,
    unless the path bites its own tail).  The censored table is the long-unitig regime: 10^5 unitigs of kilobases."""
    torch, capi, ctx, lib = env["torch"], env["capi"], env["ctx"], env["lib"]
    K, N_READS = env["k"], env["n_reads"]
    torch.cuda.empty_cache()
    t = run_filter(env, 0, N_READS, 0, 2)
    if censored:
        ctx.check(lib.dbg_remove_censored_exts(ctx.h, K, 0, C.byref(t), 0))
    g = capi.Graph()
    ctx.check(lib.dbg_compress_kmers_with_hash_dev(ctx.h, K, 0, 0, t.n, t.key_hi, t.key_lo, t.exts, None, t.count, C.byref(g)))
    n_nodes = int(g.n_nodes)
    assert n_nodes > 0
    ctx.trim()                           # the ctx's pool hands its free blocks back: the checks below allocate through torch
    def host(ptr, ctype, n):
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(int(n),))
    length = host(g.length, C.c_uint32, n_nodes)
    start = host(g.start, C.c_uint64, n_nodes)
    gexts = host(g.exts, C.c_uint8, n_nodes)
    gdata = host(g.data, C.c_uint32, n_nodes)
    words = host(g.seq_words, C.c_uint64, g.n_seq_words)
    assert int(length.min()) >= K
    assert int((length.astype(np.int64) - (K - 1)).sum()) == int(t.n)                 # every valid k-mer sits in exactly one node
    assert np.array_equal(start[1:], np.cumsum(length[:-1], dtype=np.uint64) + start[0]) and int(start[0]) == 0
    if censored:                                                                       # few, long unitigs
        assert n_nodes * 20 < int(t.n) // (150 - K) and int(length.max()) > 20 * 150
    hi_t, lo_t = dev_view(t.key_hi, t.n), dev_view(t.key_lo, t.n)
    ex_t, cn_t = dev_view(t.exts, t.n, "|u1"), dev_view(t.count, t.n, "<i2")      # int16 view: uint16 cannot be indexed on the device
    rng = np.random.default_rng(3)
    # a sample of ordinary nodes plus the longest ones (fewer of them when they are kilobases long)
    pick = np.unique(np.concatenate([rng.integers(0, n_nodes, 80 if censored else 300), np.argsort(length)[-(6 if censored else 20):]]))
    mask = (1 << (2 * K)) - 1
    top = 2 * (K - 1)
    def rc_of(v):
        r = 0
        for _ in range(K):
            r = (r << 2) | (3 - (v & 3)); v >>= 2
        return r
    def orient(e, f):                                                              # Exts::rc = byte bit-reversal
        return int(f"{int(e):08b}"[::-1], 2) if f else int(e)
    seen = set()
    n_end_checks = 0
    for u in pick:
        s0, ln = int(start[u]), int(length[u])
        bases = [int((int(words[(s0 + i) >> 5]) >> (62 - 2 * ((s0 + i) & 31))) & 3) for i in range(ln)]
        qhi, qlo, flips, mine = [], [], [], set()
        v = rcv = 0
        for i, b in enumerate(bases):
            v = ((v << 2) | b) & mask
            rcv = (rcv >> 2) | ((3 - b) << top)
            if i >= K - 1:
                c = min(v, rcv)
                flips.append(rcv < v)
                assert c not in seen                                               # no k-mer in two places
                seen.add(c); mine.add(c)
                qhi.append(c >> 64); qlo.append(c & 0xFFFFFFFFFFFFFFFF)
        pos = _lookup(torch, hi_t, lo_t, qhi, qlo)
        assert (pos >= 0).all()                                                    # only valid k-mers
        idx = torch.tensor(pos, device="cuda")
        ex = ex_t[idx].cpu().numpy()
        cn = cn_t[idx].cpu().numpy().view(np.uint16).astype(np.int64)
        assert int(gdata[u]) == min(int(cn.sum()), 65535)                          # SimpleCompress(saturating_add)
        oe = [orient(e, f) for e, f in zip(ex, flips)]
        assert (int(gexts[u]) & 0x0f) == (oe[0] & 0x0f) and (int(gexts[u]) & 0xf0) == (oe[-1] & 0xf0)
        for i in range(len(oe) - 1):                                               # the path follows the unique extensions
            assert (oe[i] >> 4) == 1 << bases[i + K] and (oe[i + 1] & 0x0f) == 1 << bases[i]
        # the node could not have been extended at either end (is_compressed at the k-mer level)
        first = last = 0
        for b in bases[:K]:
            first = (first << 2) | b
        for b in bases[-K:]:
            last = (last << 2) | b
        for end_kmer, e_out, right in ((last, oe[-1] >> 4, True), (first, oe[0] & 0x0f, False)):
            if bin(e_out).count("1") != 1:
                continue
            b = e_out.bit_length() - 1
            y = ((end_kmer << 2) | b) & mask if right else (end_kmer >> 2) | (b << top)
            yr = rc_of(y)
            yc, yf = min(y, yr), yr < y
            p1 = _lookup(torch, hi_t, lo_t, [yc >> 64], [yc & 0xFFFFFFFFFFFFFFFF])
            if p1[0] < 0:
                assert not censored                                                # censoring leaves extensions to valid k-mers only
                continue
            ye = orient(int(ex_t[int(p1[0])].item()), yf)
            back = (ye & 0x0f) if right else (ye >> 4)                              # y's extensions back towards this node
            n_end_checks += 1
            assert bin(back).count("1") != 1 or yc in mine, (int(u), right)        # ... or the path bites its own tail
    assert n_end_checks > 0
    lib.dbg_free_graph(ctx.h, C.byref(g))
    lib.dbg_free_table(ctx.h, C.byref(t))


def _sharded_cases():
    """(case, kind, world): k = 47 and 63 with two virtual ranks and both summarizers; k = 51 (config 5's key width and summarizer)
    with two; k = 47 with EIGHT owners -- the target node's shape: 3 bits finer ownership, eight source segments per bin"""
    out = []
    for ci, kind, world in ((0, 0, 2), (0, 1, 2), (1, 0, 2), (1, 1, 2), (2, 1, 2), (0, 0, 8)):
        out.append(pytest.param(CASES[ci], kind, world, id="%s-kind%d-w%d" % (CASE_IDS[ci], kind, world)))
    return out


@pytest.mark.parametrize("env,kind,world", _sharded_cases(), indirect=["env"])
def test_fullsize_sharded_virtual_ranks(env, kind, world):
    """The sharded entry points at full size: `world` virtual ranks hold an equal share of the reads each, scan them into the bins of
    one global plan (dbg_shard_scan / _scatter), each owner counts its share of the bins from `world` bin-ordered source segments
    (dbg_shard_count_begin / _bins / _finish in three ranges, as the pipelined exchange drives it) -- what a rank sees after the
    all-to-all.  Per owner: strictly ascending keys; over both owners: every instance counted once, and the order-independent
    digests of (key, Exts, count | label list) add up to the digest of the single-call table over the same reads (every k-mer
    lives on exactly one owner, msp.rs:279-324)."""
    import importlib
    e = env
    torch, capi, ctx, lib = e["torch"], e["capi"], e["ctx"], e["lib"]
    K, N_READS = e["k"], e["n_reads"]
    D = importlib.import_module("rust-debruijn_amd.distributed")
    dev = torch.device("cuda", 0)
    eng = D.HipEngine(ctx, dev)
    torch.cuda.empty_cache()
    ctx.trim()                           # the earlier tests' scratch sits in the ctx pool; torch needs room for the record tensors
    min_obs = 1 if kind == 0 else 2
    whole = run_filter(e, 0, N_READS, kind, min_obs)
    ctx.trim()
    want = dict(table_stats(e, whole), digest=D.table_digest(whole, dev))
    lib.dbg_free_table(ctx.h, C.byref(whole))
    share = N_READS // world
    shards = []
    for r in range(world):
        first, n = r * share, (share if r + 1 < world else N_READS - share * (world - 1))
        shards.append(capi.SeqSet(e["words"].data_ptr(), e["nw"], e["start"][first:].data_ptr(), e["length"][first:].data_ptr(), None,
                                  e["colour"][first:].data_ptr() if kind == 1 else None, 1 if kind == 1 else 0, n))
    total = sum(eng.count_instances(s, K) for s in shards)
    assert total == want["n_inst"]
    plan = eng.plan(K, False, kind, min_obs, total)
    rw, nb = plan.rec_words, plan.n_bins
    bounds = D.owner_bounds(nb, world, plan.bin_group)
    scanned = []
    for s in shards:
        bin_off, n = eng.scan(s, plan)
        scanned.append((bin_off, eng.scatter(plan, bin_off, n)))
    got = dict(n=0, sum_count=0, digest=0)
    for owner in range(world):
        lo, hi = bounds[owner], bounds[owner + 1]
        cb = D.chunk_bounds(hi - lo, 3, plan.bin_group)
        eng.count_begin(plan, total // world)
        for c in range(3):
            l2, h2 = lo + cb[c], lo + cb[c + 1]
            parts, seg, base = [], torch.zeros(world, h2 - l2 + 1, dtype=torch.int64, device=dev), 0
            for s_, (bin_off, recs) in enumerate(scanned):
                a, b = int(bin_off[l2]), int(bin_off[h2])
                parts.append(recs[a * rw:b * rw])
                seg[s_] = bin_off[l2:h2 + 1] - bin_off[l2] + base
                base += b - a
            rc = torch.cat(parts)
            eng.sync()
            eng.count_bins(plan, rc, seg, world, h2 - l2, 0)
            del rc, parts
        tab = eng.count_finish(plan)
        ctx.trim()
        st = table_stats(e, tab)
        assert st["ascending"] and st["n"] > 0
        got["n"] += st["n"]
        got["sum_count"] += st.get("sum_count", 0)
        got["digest"] = (got["digest"] + D.table_digest(tab, dev)) & ((1 << 64) - 1)
        eng.free_table(tab)
    assert got["n"] == want["n"]
    if kind == 0:
        assert got["sum_count"] == want["sum_count"] == total
    assert got["digest"] == want["digest"]


@only_c2
def test_fullsize_wide_colour_sets(env):
    """The WIDE colour layout (25..64 colours: two mask words per table entry, the sort carries the record's position, Exts and
    colours gathered afterwards) at full size: 64 colours, label = read index mod 64.  Same valid key set as CountFilter(2);
    every label set sorted, de-duplicated, non-empty, within 0..63; a 10^5-read prefix bit-exact against the oracle."""
    e = env
    torch, capi, lib, ctx = e["torch"], e["capi"], e["lib"], e["ctx"]
    K, N_READS = e["k"], e["n_reads"]
    torch.cuda.empty_cache()
    ctx.trim()
    colour64 = (torch.arange(N_READS, device="cuda", dtype=torch.int64) % 64).to(torch.uint8)
    torch.cuda.synchronize()                                         # (the library runs on its own stream)

    def run(first, n, summarizer, min_obs):
        ss = capi.SeqSet(e["words"].data_ptr(), e["nw"], e["start"][first:].data_ptr(), e["length"][first:].data_ptr(), None,
                         colour64[first:].data_ptr() if summarizer else None, 1 if summarizer else 0, n)
        fp = capi.FilterParams(K, 0, summarizer, min_obs, 0, 4)
        t = capi.KmerTable()
        ctx.check(lib.dbg_filter_kmers_dev(ctx.h, C.byref(ss), C.byref(fp), C.byref(t)))
        return t
    tc = run(0, N_READS, 0, 2)
    sc = table_stats(e, tc)
    lib.dbg_free_table(ctx.h, C.byref(tc))
    ts = run(0, N_READS, 1, 2)
    ss_ = table_stats(e, ts)
    ctx.trim()
    off = dev_view(ts.set_off, ts.n + 1)
    nsv = int(ts.n_set_val)
    sizes = off[1:] - off[:-1]
    assert int(off[0].item()) == 0 and int(off[-1].item()) == nsv
    assert int(sizes.min().item()) >= 1 and int(sizes.max().item()) <= 64
    is_boundary = torch.zeros(nsv + 1, dtype=torch.bool, device="cuda")
    is_boundary[off] = True
    vmin, vmax, step = 1 << 30, -1, 1 << 27
    val_all = dev_view(ts.set_val, nsv, "<u4")
    for a in range(0, nsv, step):                                   # in slices (see test_fullsize_filter_set_vs_count)
        b = min(a + step, nsv)
        v = val_all[a:b].to(torch.int64)
        vmin, vmax = min(vmin, int(v.min().item())), max(vmax, int(v.max().item()))
        if a == 0:
            nonasc = (v[1:] <= v[:-1]).nonzero().flatten() + 1
        else:
            prev = val_all[a - 1:b - 1].to(torch.int64)
            nonasc = (v <= prev).nonzero().flatten() + a
        assert bool(is_boundary[nonasc].all().item())
    del is_boundary, val_all
    lib.dbg_free_table(ctx.h, C.byref(ts))
    assert vmin == 0 and vmax == 63                                  # the wide layout was exercised
    assert sc["ascending"] and ss_["ascending"] and sc["n"] == ss_["n"] and sc["keysum"] == ss_["keysum"]
    # prefix of the stream, bit-exact
    m = min(100_000, N_READS)
    t = run(0, m, 1, 1)
    h = capi.KmerTable()
    ctx.check(lib.dbg_table_to_host(ctx.h, C.byref(t), C.byref(h)))
    lib.dbg_free_table(ctx.h, C.byref(t))
    n = h.n
    as_np = lambda p, ct, cnt: np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(max(cnt, 1),))[:cnt].copy()
    g_hi, g_lo, g_ex = as_np(h.key_hi, C.c_uint64, n), as_np(h.key_lo, C.c_uint64, n), as_np(h.exts, C.c_uint8, n)
    g_off, g_val = as_np(h.set_off, C.c_uint64, n + 1), as_np(h.set_val, C.c_uint32, h.n_set_val)
    lib.dbg_free_table(ctx.h, C.byref(h))
    hs = dbg.synth_reads_host(n_reads=m, read_len=L, genome_len=N_READS * L // 30, error_rate=0.001, stranded=False, n_colours=64)
    want = O.filter_kmers(O.SeqSet(hs.words, hs.start, hs.length, None, hs.data, 1), K, O.COUNT_FILTER_SET, 1, stranded=False)
    assert n == want.n and np.array_equal(g_hi, want.key_hi) and np.array_equal(g_lo, want.key_lo) and np.array_equal(g_ex, want.exts)
    assert np.array_equal(g_off, want.set_off) and np.array_equal(g_val, want.set_val)


@pytest.mark.parametrize("route", ["lists", "groups"])
def test_fullsize_label_groups(env, route):
    """More than 64 distinct labels at full size, by both routes -- label lists (fast_labellists.hpp, the default since round 5: one
    pass, every instance's label appended to its k-mer's segment, segments sorted + de-duplicated) and label groups
    (fast_manylabels.hpp: one CountFilter run for the valid k-mers, one 64-colour run
    per label group joined into it): 192 labels, label = i mod 64 + 64 * ((i div 64) mod 3) for read i.  Same valid key set as
    CountFilter(2); every label list strictly ascending within 0..191; against the 64-colour run with label = i mod 64 (the
    projection of these labels): every k-mer has at least as many labels and at most three times as many; a 10^5-read prefix
    bit-exact against the oracle."""
    e = env
    torch, capi, lib, ctx = e["torch"], e["capi"], e["lib"], e["ctx"]
    K, N_READS = e["k"], e["n_reads"]
    torch.cuda.empty_cache()
    ctx.trim()
    idx = torch.arange(N_READS, device="cuda", dtype=torch.int64)
    lab192 = (idx % 64 + 64 * ((idx // 64) % 3)).to(torch.uint8)
    lab64 = (idx % 64).to(torch.uint8)
    del idx
    torch.cuda.synchronize()                                         # the library runs on its own stream: the labels must be complete

    def run(n, labels, summarizer, min_obs):
        ss = capi.SeqSet(e["words"].data_ptr(), e["nw"], e["start"].data_ptr(), e["length"].data_ptr(), None,
                         labels.data_ptr() if summarizer else None, 1 if summarizer else 0, n)
        fp = capi.FilterParams(K, 0, summarizer, min_obs, 0, 4)
        t = capi.KmerTable()
        ctx.check(lib.dbg_filter_kmers_dev(ctx.h, C.byref(ss), C.byref(fp), C.byref(t)))
        return t
    t64 = run(N_READS, lab64, 1, 2)
    s64 = table_stats(e, t64)
    off = dev_view(t64.set_off, t64.n + 1)
    sizes64 = (off[1:] - off[:-1]).clone()
    torch.cuda.synchronize()                                         # (torch's copy must be done before the library frees its arrays)
    lib.dbg_free_table(ctx.h, C.byref(t64))
    ctx.trim()
    route_opt = dict(DBG_LABEL_LISTS="0" if route == "groups" else "1")
    with ctx.options(**route_opt):
        tm = run(N_READS, lab192, 1, 2)
    assert tm.n_passes == (4 if route == "groups" else 1)            # CountFilter + three label groups / one pass of label lists
    sm = table_stats(e, tm)
    assert sm["ascending"] and s64["ascending"] and sm["n"] == s64["n"] and sm["keysum"] == s64["keysum"]
    off = dev_view(tm.set_off, tm.n + 1)
    nsv = int(tm.n_set_val)
    sizes = off[1:] - off[:-1]
    assert int(off[0].item()) == 0 and int(off[-1].item()) == nsv
    assert bool((sizes >= sizes64).all().item()) and bool((sizes <= 3 * sizes64).all().item())
    assert int(sizes.max().item()) <= 192
    is_boundary = torch.zeros(nsv + 1, dtype=torch.bool, device="cuda")
    is_boundary[off] = True
    vmin, vmax, step = 1 << 30, -1, 1 << 27
    val_all = dev_view(tm.set_val, nsv, "<u4")
    for a in range(0, nsv, step):
        b = min(a + step, nsv)
        v = val_all[a:b].to(torch.int64)
        vmin, vmax = min(vmin, int(v.min().item())), max(vmax, int(v.max().item()))
        if a == 0:
            nonasc = (v[1:] <= v[:-1]).nonzero().flatten() + 1
        else:
            prev = val_all[a - 1:b - 1].to(torch.int64)
            nonasc = (v <= prev).nonzero().flatten() + a
        assert bool(is_boundary[nonasc].all().item())
    del is_boundary, val_all, sizes, sizes64
    lib.dbg_free_table(ctx.h, C.byref(tm))
    assert vmin == 0 and vmax == 191
    ctx.trim()
    # prefix of the stream, bit-exact
    m = min(100_000, N_READS)
    with ctx.options(**route_opt):
        t = run(m, lab192, 1, 1)
    h = capi.KmerTable()
    ctx.check(lib.dbg_table_to_host(ctx.h, C.byref(t), C.byref(h)))
    lib.dbg_free_table(ctx.h, C.byref(t))
    n = h.n
    as_np = lambda p, ct, cnt: np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(max(cnt, 1),))[:cnt].copy()
    g_hi, g_lo, g_ex = as_np(h.key_hi, C.c_uint64, n), as_np(h.key_lo, C.c_uint64, n), as_np(h.exts, C.c_uint8, n)
    g_off, g_val = as_np(h.set_off, C.c_uint64, n + 1), as_np(h.set_val, C.c_uint32, h.n_set_val)
    lib.dbg_free_table(ctx.h, C.byref(h))
    hs = dbg.synth_reads_host(n_reads=m, read_len=L, genome_len=N_READS * L // 30, error_rate=0.001, stranded=False, n_colours=4)
    i = np.arange(m, dtype=np.int64)
    want = O.filter_kmers(O.SeqSet(hs.words, hs.start, hs.length, None, (i % 64 + 64 * ((i // 64) % 3)).astype(np.uint8), 1), K,
                          O.COUNT_FILTER_SET, 1, stranded=False)
    assert n == want.n and np.array_equal(g_hi, want.key_hi) and np.array_equal(g_lo, want.key_lo) and np.array_equal(g_ex, want.exts)
    assert np.array_equal(g_off, want.set_off) and np.array_equal(g_val, want.set_val)


@only_c2
def test_fullsize_host_boundary_equals_device_call(env):
    """dbg_filter_kmers (round 5: the caller's packed words are uploaded in chunks of reads while the scan already runs over the chunks
    that have arrived -- dbg_ctx::read_gates; plain and compact host tables) against dbg_filter_kmers_dev over the same reads:
    keys, Exts, counts / label lists equal element for element.  Also with the reads' start offsets NOT monotone (two halves
    swapped: the whole-array upload), and with DBG_HOST_STAGING=whole (no chunking)."""
    e = env
    capi, lib, ctx, torch = e["capi"], e["lib"], e["ctx"], e["torch"]
    K, N = e["k"], min(e["n_reads"], 30_000_000)                     # 1.1 GB of packed words: chunked (>= 256 MB), three calls in seconds
    hw = e["words"].cpu().numpy()
    hst, hl, hc = e["start"][:N].cpu().numpy(), e["length"][:N].cpu().numpy(), e["colour"][:N].cpu().numpy()
    for summarizer, compact, swap, mode in ((0, 0, False, None), (1, 3, False, None), (0, 0, True, None), (1, 0, False, "whole")):
        want = run_filter(e, 0, N, summarizer, 2)
        st, ln, co = hst, hl, hc
        if swap:                                                     # the same reads, second half first: a permutation of the input
            h = N // 2
            st, ln, co = np.concatenate([hst[h:], hst[:h]]), np.concatenate([hl[h:], hl[:h]]), np.concatenate([hc[h:], hc[:h]])
        hs = capi.SeqSet(hw.ctypes.data, e["nw"], st.ctypes.data, ln.ctypes.data, None, co.ctypes.data if summarizer else None, 1 if summarizer else 0, N)
        fp = capi.FilterParams(K, 0, summarizer, 2, 0, 4, compact)
        got = capi.KmerTable()
        old = ctx.set_option("DBG_HOST_STAGING", mode)
        try:
            ctx.check(lib.dbg_filter_kmers(ctx.h, C.byref(hs), C.byref(fp), C.byref(got)))
        finally:
            ctx.set_option("DBG_HOST_STAGING", old)
        n = int(got.n)
        assert n == int(want.n) and n > 0
        host = lambda ptr, cnt, ct: torch.from_numpy(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(cnt,))).cuda()
        assert torch.equal(host(got.key_lo, n, C.c_int64), dev_view(want.key_lo, n))
        assert torch.equal(host(got.key_hi, n, C.c_int64), dev_view(want.key_hi, n))
        assert torch.equal(host(got.exts, n, C.c_uint8), dev_view(want.exts, n, "|u1"))
        if summarizer:
            nv = int(got.n_set_val)
            assert nv == int(want.n_set_val)
            ow, vw = (C.c_int32, C.c_uint8) if compact else (C.c_int64, C.c_int32)
            assert (got.set_off_width, got.set_val_width) == ((4, 1) if compact else (0, 0))
            assert torch.equal(host(got.set_off, n + 1, ow).to(torch.int64), dev_view(want.set_off, n + 1))
            assert torch.equal(host(got.set_val, nv, vw).to(torch.int64), dev_view(want.set_val, nv, "<i4").to(torch.int64))
        else:
            assert torch.equal(host(got.count, n, C.c_int16), dev_view(want.count, n, "<i2"))
        lib.dbg_free_table(ctx.h, C.byref(got))
        lib.dbg_free_table(ctx.h, C.byref(want))
