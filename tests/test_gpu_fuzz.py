"""GPU: seeded differential fuzzing of filter_kmers -> compress_kmers_with_hash against the CPU oracle.  Every case draws k, strandedness,
summarizer, min_kmer_obs, report_all_kmers, the label alphabet (narrow / wide colour layout / generic path), read lengths (ragged,
shorter than k, longer than the lane-per-read scanner takes), coverage and error rate at random, runs the default dispatch (dense
path for k <= 15, super-k-mer fast path for 16 <= k <= 64, generic path for large alphabets) and demands bit-exact tables
(src/filter.rs:139-231) and a literal BaseGraph (src/compression.rs:355-594).  The seeds are fixed: a failure is reproducible."""
import numpy as np
import pytest

import oracle_lib as O
import refgen as R
from graph_canon import graphs_equal
from pkg import dbg
from test_gpu_filter import assert_tables_equal, to_host_seqs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = dbg.Context(0)
    yield c
    c.close()


def draw_case(rng):
    k = int(rng.choice([int(rng.integers(4, 16)), int(rng.integers(16, 33)), int(rng.integers(33, 65))]))
    stranded = bool(rng.integers(0, 2))
    is_set = bool(rng.integers(0, 2))
    genome_len = int(rng.choice([300, 2000, 20000]))
    genome = R.random_dna(rng, genome_len)
    if rng.random() < 0.3:                                   # a low-complexity stretch: heavy minimizers, repeats inside a window
        a = int(rng.integers(0, genome_len - 120))
        unit = R.random_dna(rng, int(rng.integers(1, 5)))
        genome[a:a + 120] = np.tile(unit, 120)[:120]
    n_reads = int(rng.choice([30, 300, 1500]))
    long_reads = rng.random() < 0.15
    err = float(rng.choice([0.0, 0.002, 0.02]))
    seqs = []
    for _ in range(n_reads):
        mode = rng.random()
        ln = int(rng.integers(0, k + 3)) if mode < 0.1 else (int(rng.integers(1100, 1600)) if long_reads and mode < 0.3 else int(rng.integers(k, 260)))
        ln = min(ln, genome_len)
        st = int(rng.integers(0, genome_len - ln + 1))
        s = genome[st:st + ln].copy()
        m = rng.random(ln) < err
        s[m] = (s[m] + rng.integers(1, 4, size=int(m.sum()))) % 4
        if not stranded and rng.random() < 0.5:
            s = R.revcomp_bytes(s)
        seqs.append(s.astype(np.uint8))
    exts = rng.integers(0, 256, size=n_reads).astype(np.uint8) if rng.random() < 0.3 else None
    data, width = None, 0
    if is_set:
        width = int(rng.choice([1, 2, 4]))
        alphabet = {0: np.arange(int(rng.integers(1, 24))),                                  # narrow layout
                    1: np.arange(int(rng.integers(25, 64))),                                 # wide layout
                    2: np.sort(rng.choice(np.arange(200 if width == 1 else 60000), size=int(rng.integers(2, 60)), replace=False)),  # sparse alphabet
                    3: np.arange(int(rng.integers(70, 120)))}[int(rng.integers(0, 4))]      # generic path
        data = alphabet[rng.integers(0, len(alphabet), size=n_reads)]
    min_obs = int(rng.choice([1, 1, 2, 3]))
    report_all = bool(rng.integers(0, 2))
    return dict(k=k, stranded=stranded, is_set=is_set, seqs=seqs, exts=exts, data=data, width=width, min_obs=min_obs, report_all=report_all)


import os
N_SEEDS = int(os.environ.get("DBG_FUZZ_SEEDS", 160))            # DBG_FUZZ_SEEDS=5000 for a longer hunt


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_fuzz_filter_and_compress(ctx, seed):
    rng = np.random.default_rng(9000 + seed)
    c = draw_case(rng)
    k, stranded, is_set = c["k"], c["stranded"], c["is_set"]
    if is_set:
        ss = O.SeqSet.from_byte_seqs(c["seqs"], exts=c["exts"], data=c["data"], sizeof_d1=c["width"])
    else:
        ss = O.SeqSet.from_byte_seqs(c["seqs"], exts=c["exts"])
    kind = O.COUNT_FILTER_SET if is_set else O.COUNT_FILTER
    want = O.filter_kmers(ss, k, kind, c["min_obs"], stranded=stranded, report_all=c["report_all"])
    summ = (dbg.CountFilterSet if is_set else dbg.CountFilter)(c["min_obs"])
    got, _ = dbg.filter_kmers(to_host_seqs(ss, c["width"]), summ, stranded, c["report_all"], 4, k=k, ctx=ctx)
    assert_tables_equal(got, want, is_set)
    if len(got) == 0 or c["exts"] is not None:               # (random seq_exts point at k-mers that do not exist: the reference panics there)
        return
    # the index the reads produced -> unitigs, literal equality; data = count, or a stand-in for the label list (its length)
    d = got.count.astype(np.uint32) if not is_set else np.diff(got.set_off).astype(np.uint32)
    spec, ospec = (dbg.ScmapCompress(), O.SPEC_SCMAP_EQ) if is_set else (dbg.SimpleCompress("saturating_add"), O.SPEC_SAT_ADD)
    try:
        og = O.compress_kmers(k, stranded, ospec, got.key_hi, got.key_lo, got.exts, d)
    except RuntimeError:
        with pytest.raises(dbg.DbgError):                    # inconsistent Exts (k-mers filtered away by min_obs): the same panic, as an error
            dbg.compress_kmers_with_hash(stranded, spec, got, k=k, data=d, ctx=ctx)
        return
    g = dbg.compress_kmers_with_hash(stranded, spec, got, k=k, data=d, ctx=ctx)
    assert graphs_equal(g.arrays(), og.arrays())
