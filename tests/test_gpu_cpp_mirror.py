"""Runs the C++ host-mirror test program (tests/cpp/test_mirror.cpp over include/debruijn_mi355x.hpp):
the reference's reassemble_contigs / reassemble_sharded checks written against the C++ interface."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "_build", "test_mirror")


def build_cpp_mirror():
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    src = os.path.join(ROOT, "tests", "cpp", "test_mirror.cpp")
    deps = [src, os.path.join(ROOT, "include", "debruijn_mi355x.hpp"), os.path.join(ROOT, "include", "dbg_mi355x.h")]
    if os.path.exists(BIN) and all(os.path.getmtime(d) <= os.path.getmtime(BIN) for d in deps):
        return BIN
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), src, "-o", BIN,
                           "-L" + os.path.join(ROOT, "rust-debruijn_amd"), "-ldbg_mi355x",
                           "-Wl,-rpath,$ORIGIN/../../../rust-debruijn_amd", "-Wl,-rpath,/opt/rocm/lib"])
    return BIN


BIN_THREADS = os.path.join(ROOT, "tests", "cpp", "_build", "test_shard_threads")


def build_shard_threads():
    """tests/cpp/test_shard_threads.cpp: the rank-spanning entry points from a compiled host, ranks = threads, the library's
    in-process transport (plain C ABI: compiled as C++ with g++, linked against the library only)"""
    os.makedirs(os.path.dirname(BIN_THREADS), exist_ok=True)
    src = os.path.join(ROOT, "tests", "cpp", "test_shard_threads.cpp")
    deps = [src, os.path.join(ROOT, "include", "dbg_mi355x.h")]
    if os.path.exists(BIN_THREADS) and all(os.path.getmtime(d) <= os.path.getmtime(BIN_THREADS) for d in deps):
        return BIN_THREADS
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-I" + os.path.join(ROOT, "include"), src, "-o", BIN_THREADS,
                           "-L" + os.path.join(ROOT, "rust-debruijn_amd"), "-ldbg_mi355x",
                           "-Wl,-rpath,$ORIGIN/../../../rust-debruijn_amd", "-Wl,-rpath,/opt/rocm/lib"])
    return BIN_THREADS


def test_shard_threads_compiles():
    build_shard_threads()
    assert os.path.exists(BIN_THREADS)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_shard_threads_runs(world):
    """W ranks as threads of one process on the one GPU: per-rank tables merge to the single call's table row for row, the gathered
    graph equals the call-by-call composition array for array, the tree gives the same node and base counts"""
    build_shard_threads()
    r = subprocess.run([BIN_THREADS, str(world)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "shard threads ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.gpu
def test_shard_threads_one_rank_per_device():
    """multi-GPU node only: every thread-rank drives its own device, the in-process transport copies between devices"""
    import torch
    nd = torch.cuda.device_count()
    if nd < 2:
        pytest.skip("needs two GPUs")
    build_shard_threads()
    r = subprocess.run([BIN_THREADS, str(nd), str(nd)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "shard threads ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_cpp_mirror_compiles():
    """CPU check: the C++ mirror header and its test program compile and link against the C ABI library."""
    build_cpp_mirror()
    assert os.path.exists(BIN)


@pytest.mark.gpu
def test_cpp_mirror_runs():
    build_cpp_mirror()
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cpp mirror ok" in r.stdout
