"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and exports every
symbol include/dbg_mi355x.h declares; the product fails loudly (no CPU fallback) without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from pkg import dbg, capi, ROOT


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    hdr = open(os.path.join(ROOT, "include", "dbg_mi355x.h")).read()
    declared = set(re.findall(r"\b(dbg_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(dbg.DbgError):
        dbg.Context(0)


def test_synth_host_is_deterministic_and_packed():
    a = dbg.synth_reads_host(n_reads=50, read_len=150, error_rate=0.01, n_colours=4)
    b = dbg.synth_reads_host(n_reads=50, read_len=150, error_rate=0.01, n_colours=4)
    assert np.array_equal(a.words, b.words)
    assert list(a.start[:3]) == [0, 150, 300] and set(a.length) == {150}
    assert list(a.data[:6]) == [0, 1, 2, 3, 0, 1]
    # first_read shards the same stream
    c = dbg.synth_reads_host(n_reads=10, read_len=150, error_rate=0.01, n_colours=4, first_read=20,
                             genome_len=50 * 150 // 30)
    for i in range(10):
        assert np.array_equal(dbg.unpack_bases(c.words, int(c.start[i]), 150),
                              dbg.unpack_bases(a.words, int(a.start[20 + i]), 150))
    # coverage: reads come from a 250-base genome, both strands
    g = 50 * 150 // 30
    assert g == 250


def test_exts_mirror():
    e = dbg.Exts(0x01)
    assert e.rc().val == 0x80 and repr(dbg.Exts(0x21)) == "A|C"
