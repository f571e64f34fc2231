"""GPU: the N > 1 path of bench.py end to end on a one-GPU box -- torch.distributed.run with 2, 3 and 8 ranks that share
cuda:0 (gloo backend; the dbg_transport is rust-debruijn_amd/transport.py::TorchTransport, payload staged through the host).
Every rank calls the C entry points dbg_shard_filter_kmers_dev / dbg_shard_compress_dev: real kernels, the library's own
ownership, layout, exchange rounds and chunked count; the per-rank tables must add up to the single-GPU table over the same
reads (every k-mer lives on exactly one rank)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_multirank_one_gpu():
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "check_multirank.sh"), "300000"], cwd=ROOT, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and "multirank ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_collective_route_world1_nccl():
    """The RCCL calls of the N > 1 path (all_to_all_single with split sizes, asynchronous, pipelined in rounds) on the one GPU
    of the test box: a 1-rank nccl process group, the exchange route forced.  Same table digest as the plain call."""
    import json
    def run(extra):
        r = subprocess.run(["python", "bench.py", "--reads", "400000", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                            "--compress-reads", "0", "--digest"] + extra, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    a, b = run([]), run(["--force-exchange", "--backend", "nccl"])
    assert a["table_digest"] == b["table_digest"] and a["config"]["valid_kmers_all_ranks"] == b["config"]["valid_kmers_all_ranks"]
    assert b["config"]["path"].startswith("fast")


def test_collective_route_full_size_nccl():
    """The same route at BASELINE configs[1]'s full per-GPU size (10^8 reads): slabs + the compacted copy + the receive buffers
    of the pipelined rounds must fit next to one another, and no single message may reach the sizes at which RCCL transfers were
    seen to arrive incomplete -- what the first contact with a real 8-GPU node would otherwise find out."""
    import json
    r = subprocess.run(["python", "bench.py", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-host-boundary", "--compress-reads", "0",
                        "--digest", "--force-exchange", "--backend", "nccl"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["kmer_instances_per_step"] == 10_400_000_000
    assert d["config"]["valid_kmers_all_ranks"] == 501537896            # the single-call count of this stream (BENCH_r02)


@pytest.mark.parametrize("k,reads,summarizer", [(63, 125_000_000, "count"), (51, 75_000_000, "set")])
def test_collective_route_config4_config5_shares_full_size_nccl(k, reads, summarizer):
    """One GPU's share of BASELINE configs 4 and 5 (10^9 reads / 8 at k = 63: 4-word records, 20-byte outputs; 6*10^8 / 8 at k = 51 with
    label sets) through the library's RCCL table on a one-rank communicator with the exchange route forced: layout, compaction,
    ncclSend / ncclRecv groups and event-ordered rounds at the sizes the 8-GPU runs will have.  Same digest as the plain call."""
    import json
    def run(extra):
        r = subprocess.run(["python", "bench.py", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-host-boundary", "--compress-reads", "0",
                            "--digest", "--k", str(k), "--reads", str(reads), "--summarizer", summarizer] + extra, cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    a, b = run([]), run(["--force-exchange", "--backend", "nccl"])
    assert a["config"]["kmer_instances_per_step"] == reads * (150 - k + 1)
    assert a["table_digest"] == b["table_digest"] and a["config"]["valid_kmers_all_ranks"] == b["config"]["valid_kmers_all_ranks"]
    assert b["exchange"]["transport"].startswith("rccl") and b["exchange"]["rounds"] >= 4


def test_rehearsal_tool_small():
    """tools/rehearse_shard.py (thread-ranks over the in-process transport: both rank-spanning C calls against the single-GPU calls,
    table digests and an order- and strand-independent graph digest) at a size the suite can afford; the full-size runs are in
    profiles/r05_second_stage.txt"""
    r = subprocess.run(["python", os.path.join(ROOT, "tools", "rehearse_shard.py"), "--ranks", "3", "--reads-per-rank", "40000"], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "rehearsal ok" in r.stdout and r.stdout.count("EQUAL") == 3, r.stdout[-3000:] + r.stderr[-2000:]


def test_strong_scaled_baseline_shapes_two_ranks_one_gpu():
    """bench.py --config c4 / c5: BASELINE shapes that name a TOTAL size, split over the ranks (strong scaling).  Two ranks share
    cuda:0 (gloo); a scaled-down total; the tables add up to the single-call table of the same reads and k."""
    import json
    def run(extra):
        r = subprocess.run(["python", "bench.py", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-host-boundary", "--compress-reads", "0",
                            "--digest"] + extra, cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    for cfg, k in (("c4", 63), ("c5", 51)):
        a = run(["--reads", "400000", "--k", str(k)])
        b = run(["--config", cfg, "--reads", "400000", "--gpus", "2", "--backend", "gloo", "--one-device"])
        assert b["scaling"] == "strong" and b["n_gpus"] == 2 and ("k=%d" % k) in b["config"]["workload"]
        assert a["table_digest"] == b["table_digest"] and a["config"]["valid_kmers_all_ranks"] == b["config"]["valid_kmers_all_ranks"]
        assert b["balance"]["records_owned"]["min"] > 0


def test_two_gpus_nccl():
    """Real multi-GPU run (skips on a one-GPU box): 2 ranks, nccl = RCCL over xGMI; per-rank tables add up to the single-GPU
    table, and the rank-spanning compress stage gives the single-process result."""
    import json
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    def run(extra, reads):
        r = subprocess.run(["python", "bench.py", "--reads", str(reads), "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                            "--compress-reads", "0", "--digest"] + extra, cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    a, b = run([], 600000), run(["--gpus", "2"], 300000)
    assert b["n_gpus"] == 2 and b["ranks_seen"] == 2 and b["backend"] == "nccl"
    assert a["table_digest"] == b["table_digest"] and a["config"]["valid_kmers_all_ranks"] == b["config"]["valid_kmers_all_ranks"]
    r = subprocess.run(["python", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29733", os.path.join(ROOT, "tools", "check_sharded_compress.py"), "--backend", "nccl"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "sharded compress ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_sharded_compress_two_ranks_one_gpu():
    """distributed.sharded_compress with the product engine: 2 ranks share cuda:0 (gloo), per-rank compress of the owned bins,
    graphs gathered on rank 0, combine + compress_graph; rank 0 checks the result against the oracle's same flow."""
    r = subprocess.run(["python", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29731", os.path.join(ROOT, "tools", "check_sharded_compress.py"), "--backend", "gloo", "--one-device"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "sharded compress ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("world,reduce,segments", [(3, "gather", False), (3, "tree", False), (2, "tree", False), (4, "tree", False),
                                                   (3, "gather", True), (2, "tree", True)])
def test_sharded_compress_more_shapes_one_gpu(world, reduce, segments):
    """dbg_shard_compress_dev with 3 ranks (an odd tree: one rank sits a level out) and the tree merge: gather is compared node for
    node with the oracle's combine + compress_graph, the tree in canonical form (same unitigs, other order / strand).
    segments: every rank's per-shard unitig construction takes the segment route (unitig.hip: forced, every 3rd k-mer cut) with the
    shard graph staying in HBM (dbg_ctx::graph_sink set by the caller)."""
    env = dict(os.environ, DBG_SEGMENTS_FORCE="1", DBG_SEGMENTS="3") if segments else None
    r = subprocess.run(["python", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                        "--master-port", str(29741 + world + (10 if segments else 0)), os.path.join(ROOT, "tools", "check_sharded_compress.py"), "--backend", "gloo",
                        "--one-device", "--reduce", reduce, "--reads", "21000"], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "sharded compress ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_ownership_follows_the_record_histogram_on_low_complexity_reads():
    """8 ranks on the one GPU, 5 % poly-A / di- / tri-nucleotide repeat reads (tools/check_balance.py): with ownership cut from the
    all-reduced record histogram (dbg_shard_params.balance, dbg_shard_owner_bounds) every rank owns the same number of records to
    within 5 %, equal bin ranges do not; both give the single call's table."""
    import json
    r = subprocess.run(["python", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", "29761", os.path.join(ROOT, "tools", "check_balance.py"), "--backend", "gloo", "--one-device",
                        "--reads", "100000"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["ok"] and d["world"] == 8
    assert d["balanced"]["records_owned_max_over_mean"] <= 1.05 < d["equal_bins"]["records_owned_max_over_mean"]


def test_config4_and_config5_flow_1e6_reads_two_ranks():
    """BASELINE configs 4 and 5 as flows, at 10^6 reads over two ranks (one GPU, gloo): dbg_shard_filter_kmers_dev ->
    dbg_shard_compress_dev (k = 47 counts / saturating_add then max; k = 51 CountFilterSet -> label-list classes -> ScmapCompress)
    -> combine -> compress_graph.  The union of the ranks' tables is the oracle's filter_kmers over all reads, and the final graph
    is the oracle's, node for node (src/test.rs:433-470)."""
    r = subprocess.run(["python", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29771", os.path.join(ROOT, "tools", "check_sharded_compress.py"), "--backend", "gloo", "--one-device",
                        "--reads", "1000000", "--check-table"], cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "sharded compress ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
