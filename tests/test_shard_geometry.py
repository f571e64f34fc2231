"""CPU: the host arithmetic of the rank-spanning flow that lives in the library (dbg_shard_owner_bounds, dbg_shard_round_cuts,
include/dbg_mi355x.h) -- ownership of the global bin space and the cuts of the exchange rounds.  No GPU: these are the functions
every rank evaluates identically from all-reduced values, so their properties are what keeps the ranks in step."""
import ctypes as C

import numpy as np
import pytest

from pkg import capi, D


def bounds(n_bins, world, group=1, hist=None):
    return D.owner_bounds(n_bins, world, group, hist)


def test_equal_split_matches_the_earlier_python_rule():
    for n_bins, world, grp in ((40, 8, 1), (5, 2, 1), (3, 2, 1), (1300000, 8, 1), (96, 3, 4), (7, 8, 1)):
        want = [(r * (n_bins // grp) // world) * grp for r in range(world + 1)]
        assert bounds(n_bins, world, grp) == want


def test_balanced_split_is_monotone_covers_everything_and_balances():
    rng = np.random.default_rng(7)
    for world in (2, 3, 8):
        for trial in range(20):
            n = int(rng.integers(world, 5000))
            h = rng.integers(0, 1000, n).astype(np.uint64)
            if trial % 3 == 0:                                   # a few very heavy bins (low-complexity minimizers)
                h[rng.integers(0, n, 3)] += np.uint64(200000)
            b = bounds(n, world, 1, h)
            assert b[0] == 0 and b[-1] == n and all(b[i] <= b[i + 1] for i in range(world))
            assert all(b[i] < b[i + 1] for i in range(world))    # n >= world: nobody is left without a bin
            own = np.array([h[b[r]:b[r + 1]].sum() for r in range(world)], dtype=np.float64)
            mean, heavy = own.mean(), float(h.max())
            # a contiguous cut cannot do better than one bin's weight around the target
            assert own.max() <= mean + heavy + 1 and own.min() >= mean - heavy * world - 1


def test_balanced_split_moves_boundaries_away_from_a_hot_range():
    n, world = 8000, 8
    h = np.full(n, 100, np.uint64)
    h[:1000] = 800                                               # the first eighth of the bins holds half the records
    eq, bal = bounds(n, world), bounds(n, world, 1, h)
    own = lambda b: np.array([h[b[r]:b[r + 1]].sum() for r in range(world)], dtype=np.float64)
    assert own(eq).max() / own(eq).mean() > 3.0
    assert own(bal).max() / own(bal).mean() < 1.01
    assert bal[1] < eq[1]


def test_groups_and_degenerate_shapes():
    h = np.array([5, 0, 0, 0, 0, 0, 0, 9], np.uint64)
    b = bounds(32, 4, 4, h)                                      # 8 groups of 4 bins
    assert all(x % 4 == 0 for x in b) and b[0] == 0 and b[-1] == 32 and all(b[i] < b[i + 1] for i in range(4))
    assert bounds(3, 8, 1, np.array([1, 1, 1], np.uint64))[-1] == 3          # fewer bins than ranks: still a partition
    z = bounds(64, 4, 1, np.zeros(64, np.uint64))                # no records at all
    assert z[0] == 0 and z[-1] == 64 and all(z[i] <= z[i + 1] for i in range(4))
    lib = capi.load()
    out = (C.c_uint32 * 3)()
    assert lib.dbg_shard_owner_bounds(None, 10, 4, 2, out) != 0  # n_bins not a multiple of the group


@pytest.mark.parametrize("world,n_bins,rounds", [(2, 5, 4), (2, 3, 3), (8, 1300000, 8), (3, 9, 64), (1, 7, 4)])
def test_round_cuts(world, n_bins, rounds):
    b, nr, cuts = D.exchange_geometry(n_bins, world, 1, rounds, force=True)
    smallest = min(b[d + 1] - b[d] for d in range(world))
    assert nr == max(1, min(rounds, smallest))                   # every rank arrives at the same number of rounds
    for d in range(world):
        c = cuts[d]
        assert len(c) == nr + 1 and c[0] == 0 and c[-1] == b[d + 1] - b[d] and all(c[i] <= c[i + 1] for i in range(nr))
