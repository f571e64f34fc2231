"""GPU: the N > 1 path of bench.py end to end on a one-GPU box -- torch.distributed.run with 2 and 3 ranks that share
cuda:0 (gloo backend, payload staged through the host; rust-debruijn_amd/distributed.py::_all_to_all).  Real kernels, the
real pipelined exchange + chunked count; the per-rank valid k-mer counts must add up to the single-GPU count over the same
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
