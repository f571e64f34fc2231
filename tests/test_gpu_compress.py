"""GPU parity: dbg_compress_kmers_with_hash (device link builder + host greedy walk) and
dbg_remove_censored_exts vs the CPU oracle's restatement of src/compression.rs:355-594 and
src/filter.rs:238-306.  With the same seed order the BaseGraph must be literally identical
(packed words, start, length, exts, data); the reference's own invariants (test.rs:386-413,
:248-254) are re-checked on the GPU result."""
import numpy as np
import pytest

import oracle_lib as O
import refgen as R
from graph_canon import canonical_nodes, graph_kmer_set, graphs_equal, node_bases
from pkg import dbg

pytestmark = pytest.mark.gpu

SPECS = [(dbg.SimpleCompress("saturating_add"), O.SPEC_SAT_ADD), (dbg.SimpleCompress("add_mod_65535"), O.SPEC_ADD_MOD),
         (dbg.SimpleCompress("max"), O.SPEC_MAX), (dbg.SimpleCompress("wrapping_add"), O.SPEC_WRAP_ADD)]


@pytest.fixture(scope="module")
def ctx():
    c = dbg.Context(0)
    yield c
    c.close()


@pytest.fixture(autouse=True, params=["device", "host"])
def compress_mode(request, ctx):
    """Every test runs twice: DBG_COMPRESS=device (pointer-jumping unitig construction on the GPU; fails loudly
    instead of falling back) and DBG_COMPRESS=host (device links + literal greedy walk on the host)."""
    import os
    old = ctx.set_option("DBG_COMPRESS", request.param)
    yield request.param
    ctx.set_option("DBG_COMPRESS", old)


@pytest.fixture(autouse=True, params=[None, "3", "2", "50"], ids=["", "segments3", "segments2", "segments50"])
def forced_segments(request, ctx, compress_mode):
    """Every test of the device route runs three times: as is, and with the segment route of unitig.hip forced onto every graph
    (DBG_SEGMENTS_FORCE; every 3rd / 2nd / 50th k-mer cuts its right link, the pieces are compressed and joined again by compress_graph's
    device route; a cycle is cut at its seed k-mer -- named by the second level when it holds a cut, found by a marking walk when it
    holds none, as with every 50th -- and the route runs once more): the random contigs, palindromes, hairpins, degenerate inputs,
    ScmapCompress data and seed orders of the tests below all cross the cut-and-join."""
    if request.param is None:
        yield None
        return
    if compress_mode != "device":
        pytest.skip("device route")
    o1, o2 = ctx.set_option("DBG_SEGMENTS_FORCE", "1"), ctx.set_option("DBG_SEGMENTS", request.param)
    yield request.param
    ctx.set_option("DBG_SEGMENTS_FORCE", o1); ctx.set_option("DBG_SEGMENTS", o2)


def gpu_table(ctx, contigs, k, min_obs, stranded, dup=1):
    seqs = []
    for c in contigs:
        seqs.extend([(c, 0, None)] * dup)
    t, _ = dbg.filter_kmers(seqs, dbg.CountFilter(min_obs), stranded, False, 4, k=k, ctx=ctx)
    return t


def compare(ctx, t, k, stranded, spec_pair, seed_order=None, data=None):
    spec, ospec = spec_pair
    d = t.count if data is None else data
    got = dbg.compress_kmers_with_hash(stranded, spec, t, k=k, seed_order=seed_order, data=d, ctx=ctx)
    want = O.compress_kmers(k, stranded, ospec, t.key_hi, t.key_lo, t.exts, d, seed_order)
    wa = want.arrays()
    assert graphs_equal(got.arrays(), wa), "BaseGraph differs from the oracle"
    return got, want


@pytest.mark.parametrize("k,stranded", [(32, False), (31, False), (31, True), (47, False), (63, False), (64, False), (16, False)])
def test_compress_parity_random_contigs(ctx, k, stranded):
    rng = np.random.default_rng(k + 7 * stranded)
    for trial in range(3):
        contigs = R.random_contigs(rng) if trial else R.simple_random_contigs(rng)
        t = gpu_table(ctx, contigs, k, 1, stranded)
        got, want = compare(ctx, t, k, stranded, SPECS[trial % len(SPECS)])
        # reference invariants on the GPU result (test.rs:386-413, :248-254)
        ga = got.arrays()
        truth = set(t.keys())
        assert graph_kmer_set(ga, k, stranded) == truth
        assert sum(int(x) - k + 1 for x in ga["length"]) == len(truth)
        og = O.graph_from_arrays(k, stranded, ga["words"], ga["start"], ga["length"], ga["exts"], ga["data"])
        assert og.is_compressed(SPECS[trial % len(SPECS)][1]) is None


def test_compress_seed_order_policy_a(ctx):
    """A caller-supplied seed order (the MPHF slot order in a real deployment) is honoured literally."""
    rng = np.random.default_rng(123)
    contigs = R.random_contigs(rng)
    t = gpu_table(ctx, contigs, 31, 1, False)
    base = None
    for _ in range(4):
        perm = rng.permutation(len(t)).astype(np.uint64)
        got, _ = compare(ctx, t, 31, False, SPECS[0], seed_order=perm)
        canon = canonical_nodes(got.arrays(), 31, False)
        assert base is None or canon == base          # policy B: canonical form is seed-order independent
        base = canon


def test_compress_unsorted_index(ctx):
    """index arrays may arrive in any key order (a BoomHashMap2 is in slot order)."""
    rng = np.random.default_rng(5)
    t = gpu_table(ctx, R.random_contigs(rng), 31, 1, False)
    perm = rng.permutation(len(t))
    import copy
    u = copy.copy(t)
    u.key_hi, u.key_lo, u.exts, u.count = t.key_hi[perm], t.key_lo[perm], t.exts[perm], t.count[perm]
    got = dbg.compress_kmers_with_hash(False, SPECS[0][0], u, k=31, ctx=ctx)
    want = O.compress_kmers(31, False, SPECS[0][1], u.key_hi, u.key_lo, u.exts, u.count, None)
    assert graphs_equal(got.arrays(), want.arrays())


def test_compress_degenerate_and_palindromes(ctx):
    seq = R.from_ascii(R.DEGEN)                                      # test.rs:170-193
    t = gpu_table(ctx, [seq, seq], 31, 2, False)
    compare(ctx, t, 31, False, SPECS[0])
    rng = np.random.default_rng(17)                                  # even k: palindromic k-mers are terminal
    half = R.random_dna(rng, 16)
    pal = np.concatenate([R.random_dna(rng, 40), half, R.revcomp_bytes(half), R.random_dna(rng, 40)])
    t = gpu_table(ctx, [pal], 32, 1, False)
    got, _ = compare(ctx, t, 32, False, SPECS[0])
    assert any(int(l) == 32 for l in got.arrays()["length"])         # the palindrome sits alone in its node


def test_compress_cycle(ctx):
    """An isolated cycle is cut at the first-visited k-mer and keeps its hanging exts."""
    rng = np.random.default_rng(3)
    k = 21
    c = R.random_dna(rng, 60)
    circ = np.concatenate([c, c[:k]])                                # covers every k-mer of the circular string once
    for stranded in (True, False):
        t = gpu_table(ctx, [circ], k, 1, stranded)
        got, _ = compare(ctx, t, k, stranded, SPECS[0])
        ga = got.arrays()
        assert len(ga["start"]) == 1 and int(ga["length"][0]) == 60 + k - 1


def test_compress_hanging_exts_after_filter(ctx):
    """k-mers below min_obs are dropped but exts towards them stay (filter.rs never prunes); compression
    then ends the unitig there (compression.rs:386, :410-415)."""
    rng = np.random.default_rng(29)
    g = R.random_dna(rng, 400)
    err = g[100:250].copy()
    err[75] = (err[75] + 1) % 4
    reads = [g[i:i + 150] for i in range(0, 250, 10)] * 2 + [err]
    k = 31
    t, allk = dbg.filter_kmers([(r, 0, None) for r in reads], dbg.CountFilter(2), False, True, 4, k=k, ctx=ctx)
    compare(ctx, t, k, False, SPECS[0])
    # remove_censored_exts_sharded / remove_censored_exts (filter.rs:238-306)
    ah = np.array([v >> 64 for v in allk], np.uint64)
    al = np.array([v & O.M64 for v in allk], np.uint64)
    want_s = O.remove_censored_exts(k, False, t.key_hi, t.key_lo, t.exts, ah, al, sharded=True)
    want_n = O.remove_censored_exts(k, False, t.key_hi, t.key_lo, t.exts)
    import copy
    ts = dbg.remove_censored_exts_sharded(False, copy.copy(t), allk, ctx=ctx)
    tn = dbg.remove_censored_exts(False, copy.copy(t), ctx=ctx)
    assert np.array_equal(ts.exts, want_s) and np.array_equal(tn.exts, want_n)
    assert not np.array_equal(want_n, t.exts)                        # something was actually censored
    got, _ = compare(ctx, tn, k, False, SPECS[0])
    og = O.graph_from_arrays(k, False, **{x: got.arrays()[x] for x in ("words", "start", "length", "exts", "data")})
    assert og.is_compressed(SPECS[0][1]) is None


def test_scmap_compress(ctx):
    rng = np.random.default_rng(8)
    a = R.random_dna(rng, 200)
    k = 31
    t = gpu_table(ctx, [a], k, 1, True)
    order = {v: i for i, v in enumerate(t.keys())}
    col = np.zeros(len(t), np.uint32)
    for j, v in enumerate(R.kmers_of(a, k)):
        col[order[v]] = j // 50
    got, _ = compare(ctx, t, k, True, (dbg.ScmapCompress(), O.SPEC_SCMAP_EQ), data=col)
    assert len(got) == 4


def test_c1_shape_end_to_end(ctx):
    """BASELINE config 1 shape (10k x 150 bp, k=31, stranded, CountFilter(1)) through filter + compress."""
    hs = dbg.synth_reads_host(n_reads=10000, read_len=150, error_rate=0.001, stranded=True, n_colours=0)
    t, _ = dbg.filter_kmers(hs, dbg.CountFilter(1), True, False, 4, k=31, ctx=ctx)
    compare(ctx, t, 31, True, SPECS[0])


def test_device_resident_index(ctx, compress_mode):
    """dbg_filter_kmers_dev -> dbg_compress_kmers_with_hash_dev (index never leaves HBM) equals the host-boundary calls."""
    if compress_mode != "device":
        pytest.skip("device-resident entry point")
    rng = np.random.default_rng(77)
    for k, stranded in [(31, False), (47, False), (47, True), (64, False)]:
        contigs = R.random_contigs(rng)
        seqs = [(c, 0, None) for c in contigs] * 2
        got, n = dbg.filter_and_compress_dev(seqs, dbg.CountFilter(2), stranded, SPECS[0][0], k, ctx=ctx)
        t, _ = dbg.filter_kmers(seqs, dbg.CountFilter(2), stranded, False, 4, k=k, ctx=ctx)
        assert n == len(t)
        want = O.compress_kmers(k, stranded, SPECS[0][1], t.key_hi, t.key_lo, t.exts, t.count, None)
        assert graphs_equal(got.arrays(), want.arrays())


@pytest.mark.parametrize("k", [47, 63])
def test_device_resident_index_refuses_unsorted_keys(ctx, k):
    """dbg_compress_kmers_with_hash_dev needs strictly ascending keys (its neighbour lookups are searches in that order): two rows
    swapped, or one repeated, and the call fails -- checked while the keys are packed (k <= 60) or in a pass of its own (k > 60)."""
    import ctypes as C
    import torch
    capi = __import__("importlib").import_module("rust-debruijn_amd._capi")
    rng = np.random.default_rng(k)
    seqs = [(c, 0, None) for c in R.random_contigs(rng)]
    t, _ = dbg.filter_kmers(seqs, dbg.CountFilter(1), False, False, 4, k=k, ctx=ctx)
    n = len(t)
    assert n > 100
    for case in ("ok", "swapped", "repeated"):
        hi, lo = t.key_hi.copy(), t.key_lo.copy()
        if case == "swapped":
            hi[[40, 41]], lo[[40, 41]] = hi[[41, 40]], lo[[41, 40]]
        if case == "repeated":
            hi[n - 1], lo[n - 1] = hi[n - 2], lo[n - 2]
        d = [torch.from_numpy(a.view(np.int64)).cuda() for a in (hi, lo)] + [torch.from_numpy(t.exts.copy()).cuda()]
        torch.cuda.synchronize()
        g = capi.Graph()
        rc = ctx.lib.dbg_compress_kmers_with_hash_dev(ctx.h, k, 0, 0, n, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), None, None, C.byref(g))
        if case == "ok":
            assert rc == 0 and g.n_nodes > 0
            ctx.lib.dbg_free_graph(ctx.h, C.byref(g))
        else:
            assert rc != 0 and b"ascending" in ctx.lib.dbg_last_error(ctx.h)


@pytest.mark.parametrize("k,stranded,spec_i", [(47, False, 0), (31, True, 2), (63, False, 1)])
def test_compress_long_chains(ctx, compress_mode, k, stranded, spec_i):
    """Error-free reads of a random genome: a handful of nodes, each tens of thousands of k-mers long.  The chain route
    gives up on chains longer than 2^14 k-mers and the doubling route builds them; every wavefront of its emit kernel
    then feeds one or two nodes (data folded per wavefront, not per k-mer)."""
    if compress_mode != "device":
        pytest.skip("device route")
    rng = np.random.default_rng(900 + k)
    genome = R.random_dna(rng, 60000)
    contigs = [genome[a:a + 3000] for a in range(0, 60000 - 3000 + 1, 1000)]      # overlapping pieces: every k-mer 3 times
    if not stranded:
        contigs = [c if i % 2 else (3 - c)[::-1].copy() for i, c in enumerate(contigs)]
    # (round 6) an isolated cycle longer than the walkers' cap next to the long chains: no walker reaches it, the doubling finds it
    cyc = R.random_dna(rng, 20000)
    contigs.append(np.concatenate([cyc, cyc[:k - 1]]))
    contigs.append(np.concatenate([cyc[10000:], cyc[:10000 + k - 1]]))
    # ... and a short one, which the segment route's sampled cuts (every 128th k-mer) are likely to miss: found by the marking walk
    tiny = R.random_dna(rng, 70)
    contigs.append(np.concatenate([tiny, tiny, tiny[:k - 1]]))
    t = gpu_table(ctx, contigs, k, 2, stranded)
    ctx.enable_timing(True)
    got, want = compare(ctx, t, k, stranded, SPECS[spec_i])
    assert "unitig_segments_joined" in [x["name"] for x in ctx.timings()]      # no cycle sends the call to the table route any more
    ctx.enable_timing(False)
    assert max(int(x) for x in got.arrays()["length"]) > (1 << 14) + k
    with ctx.options(DBG_SEGMENTS="16"):                             # every cycle holds a cut: cut again at its seed, joined on the second run
        ctx.enable_timing(True)
        compare(ctx, t, k, stranded, SPECS[spec_i])
        assert "unitig_segments_joined" in [x["name"] for x in ctx.timings()]
        ctx.enable_timing(False)
    # the doubling over the states the walkers left (default since round 6) against the doubling over every state
    with ctx.options(DBG_UNITIG_FULL_DOUBLING="1"):
        compare(ctx, t, k, stranded, SPECS[spec_i])
    order = rng.permutation(len(t)).astype(np.uint64)
    compare(ctx, t, k, stranded, SPECS[spec_i], seed_order=order)


@pytest.mark.parametrize("k,stranded,spec_i,every", [(47, False, 0, None), (31, True, 2, "7"), (63, False, 1, "2"), (32, False, 3, "16"), (64, True, 0, "64"),
                                                    (21, False, 1, "33")])
def test_compress_segment_route(ctx, compress_mode, k, stranded, spec_i, every):
    """Long open chains (round 6: the segment route of unitig.hip): every n-th k-mer cuts the link on its right side, the pieces go through
    the chain route, compress_graph's device route joins them.  Two haplotypes that differ every few kilobases (branches: real chain ends
    with more than one extension), palindromic k-mers for even k, data folds of every spec, an explicit seed order -- the BaseGraph must
    be the oracle's, and the route must have run to its end (no fall-back to the table route)."""
    if compress_mode != "device":
        pytest.skip("device route")
    rng = np.random.default_rng(4200 + k)
    genome = R.random_dna(rng, 70000)
    if k % 2 == 0:                                                   # palindromes: x + rc(x)
        for at in (9000, 33000, 51000):
            x = genome[at:at + k // 2]
            genome[at + k // 2:at + k] = (3 - x)[::-1]
    hap = genome.copy()
    for at in range(2500, 70000, 6000):
        hap[at] = (hap[at] + 1 + rng.integers(0, 3)) % 4             # a SNP: a bubble, four chain ends with a branch behind them
    contigs = []
    for g in (genome, hap):
        contigs += [g[a:a + 3000] for a in range(0, 70000 - 3000 + 1, 1000)]
    if not stranded:
        contigs = [c if i % 2 else (3 - c)[::-1].copy() for i, c in enumerate(contigs)]
    t = gpu_table(ctx, contigs, k, 2, stranded)
    opts = {"DBG_SEGMENTS": every} if every else {}
    with ctx.options(**opts):
        ctx.enable_timing(True)
        got, want = compare(ctx, t, k, stranded, SPECS[spec_i])
        names = [x["name"] for x in ctx.timings()]
        assert "unitig_segments_joined" in names, names
        assert max(int(x) for x in got.arrays()["length"]) > 3000
        order = rng.permutation(len(t)).astype(np.uint64)
        ctx.enable_timing(True)                                      # (clears the list)
        compare(ctx, t, k, stranded, SPECS[spec_i], seed_order=order)
        assert "unitig_segments_joined" in [x["name"] for x in ctx.timings()]
        data = rng.integers(0, 65536, len(t)).astype(np.uint32)     # folds that saturate / wrap
        ctx.enable_timing(True)
        compare(ctx, t, k, stranded, SPECS[spec_i], data=data)
        assert "unitig_segments_joined" in [x["name"] for x in ctx.timings()]
        ctx.enable_timing(False)
    with ctx.options(DBG_SEGMENTS="0"):                              # the table route on the same input
        ctx.enable_timing(True)
        compare(ctx, t, k, stranded, SPECS[spec_i])
        assert "unitig_segments_joined" not in [x["name"] for x in ctx.timings()]
        ctx.enable_timing(False)


def test_compress_segment_route_few_long_chains(ctx, compress_mode):
    """An ordinary graph (thousands of short chains) with two chains beyond the chain route's cap: the chain route gives up on them,
    the segment route takes the whole graph (before round 6: the table route)."""
    if compress_mode != "device":
        pytest.skip("device route")
    k, stranded = 31, False
    rng = np.random.default_rng(77)
    contigs = [R.random_dna(rng, 40 + int(rng.integers(0, 60))) for _ in range(4000)]
    for _ in range(2):
        g = R.random_dna(rng, 30000)
        contigs += [g[a:a + 4000] for a in range(0, 30000 - 4000 + 1, 2000)]
    t = gpu_table(ctx, contigs, k, 1, stranded)
    ctx.enable_timing(True)
    got, want = compare(ctx, t, k, stranded, SPECS[0])
    names = [x["name"] for x in ctx.timings()]
    ctx.enable_timing(False)
    assert "unitig_segments_joined" in names, names
    assert max(int(x) for x in got.arrays()["length"]) > (1 << 14) + k and len(got.arrays()["length"]) > 3000


def test_compress_one_chain_few_ends(ctx, compress_mode):
    """An error-free genome in one piece: two chain ends in front of 4e5 states.  The table route leaves the end walks out (they would
    cover 2 x 2^15 states at a dependent read each) and doubles over everything; the segment route cuts, joins, and meets the same
    situation among its pieces when the genome is larger."""
    if compress_mode != "device":
        pytest.skip("device route")
    k = 31
    rng = np.random.default_rng(5150)
    g = R.random_dna(rng, 200000)
    contigs = [g[a:a + 5000] for a in range(0, 200000 - 5000 + 1, 1000)]
    t = gpu_table(ctx, contigs, k, 2, False)
    got, want = compare(ctx, t, k, False, SPECS[0])
    assert len(got.arrays()["length"]) <= 3 and max(int(x) for x in got.arrays()["length"]) > 150000
    with ctx.options(DBG_SEGMENTS="0"):
        ctx.enable_timing(True)
        compare(ctx, t, k, False, SPECS[0])
        names = [x["name"] for x in ctx.timings()]
        ctx.enable_timing(False)
    assert "unitig_pointer_jump" in names and "unitig_segments_joined" not in names


def test_compress_non_mutual_links(ctx, compress_mode):
    """Inconsistent Exts: A's only right extension leads to B and B has a single left extension, but to a k-mer that is
    not A.  The reference walks such input in visiting order (compression.rs:450-541); the device routes refuse it
    (the link from A is not answered by B's facing link -- found while the chains are walked) and the literal host walk
    reproduces the reference."""
    import os
    if compress_mode != "device":
        pytest.skip("sets the mode itself")
    rng = np.random.default_rng(31)
    k = 31
    contigs = R.simple_random_contigs(rng)
    t = gpu_table(ctx, contigs, k, 1, True)
    ex = t.exts.copy()
    changed = 0
    for i in range(len(ex)):
        left = int(ex[i]) & 0x0f
        if bin(left).count("1") == 1 and changed < 5 and i % 7 == 3:
            ex[i] = (int(ex[i]) & 0xf0) | (((left << 1) | (left >> 3)) & 0x0f)      # the one left extension now names another base
            changed += 1
    assert changed
    t.exts[:] = ex
    want = O.compress_kmers(k, True, O.SPEC_SAT_ADD, t.key_hi, t.key_lo, t.exts, t.count, None)
    ctx.set_option("DBG_COMPRESS", "auto")
    got = dbg.compress_kmers_with_hash(True, SPECS[0][0], t, k=k, data=t.count, ctx=ctx)
    assert graphs_equal(got.arrays(), want.arrays())
    ctx.set_option("DBG_COMPRESS", "device")
    with pytest.raises(dbg.DbgError):
        dbg.compress_kmers_with_hash(True, SPECS[0][0], t, k=k, data=t.count, ctx=ctx)


@pytest.mark.parametrize("env", [{}, {"DBG_UNITIG_NO_CHAINS": "1"}, {"DBG_UNITIG_NO_WALK": "1"},
                                 {"DBG_NO_KEY_RECORDS": "1", "DBG_NO_NODE_RECORDS": "1"}])
@pytest.mark.parametrize("k,stranded", [(47, False), (31, True), (63, False)])
def test_compress_routes(ctx, compress_mode, env, k, stranded):
    """The device construction has three routes (chain walks on packed records or on the separate arrays; end walk into a
    per-state table; doubling) -- all must give the reference's BaseGraph, cycles and seed orders included."""
    import os
    if compress_mode != "device":
        pytest.skip("device routes")
    rng = np.random.default_rng(5 * k + stranded)
    contigs = R.random_contigs(rng)
    cyc = R.random_dna(rng, 3 * k)
    contigs.append(np.concatenate([cyc, cyc[:k - 1]]))                       # an isolated cycle among the chains
    t = gpu_table(ctx, contigs, k, 1, stranded)
    with ctx.options(**env):
        compare(ctx, t, k, stranded, SPECS[0])
        order = rng.permutation(len(t)).astype(np.uint64)
        compare(ctx, t, k, stranded, SPECS[2], seed_order=order)


@pytest.mark.parametrize("k,stranded", [(31, False), (32, True), (47, False), (63, False), (64, True)])
def test_compress_kmers_no_exts_parity(ctx, k, stranded):
    """compress_kmers_no_exts (compression.rs:619-659): Exts from eight neighbour probes into the k-mer set (canonicalised with
    min_rc whether stranded or not), then compress_kmers -- literal BaseGraph and Exts equality with the oracle, keys in a
    shuffled order, random seed order; a duplicate key is the reference's assertion."""
    rng = np.random.default_rng(300 + k)
    t = gpu_table(ctx, R.random_contigs(rng), k, 1, False)
    perm = rng.permutation(len(t))
    hi, lo, cnt = t.key_hi[perm], t.key_lo[perm], t.count[perm].astype(np.uint32)
    for seed_order in (None, rng.permutation(len(t)).astype(np.uint64)):
        want, wex = O.compress_kmers_no_exts(k, stranded, O.SPEC_SAT_ADD, hi, lo, cnt, seed_order)
        assert np.array_equal(dbg.kmer_set_exts(k, hi, lo, ctx=ctx), wex)
        got = dbg.compress_kmers_no_exts(stranded, dbg.SimpleCompress("saturating_add"), k, hi, lo, cnt, seed_order=seed_order, ctx=ctx)
        assert graphs_equal(got.arrays(), want.arrays())
    # the ascending table itself (no host sort inside the call) gives the same Exts
    assert np.array_equal(dbg.kmer_set_exts(k, t.key_hi, t.key_lo, ctx=ctx)[perm], wex)
    with pytest.raises(dbg.DbgError):
        dbg.compress_kmers_no_exts(stranded, dbg.SimpleCompress("saturating_add"), k, np.concatenate([hi, hi[:1]]), np.concatenate([lo, lo[:1]]), ctx=ctx)
