"""Density of sampling schemes for the internal partition of the fast path (DESIGN.md section 7, round-4 review item 2).

A super-k-mer record = a maximal run of consecutive k-mers (windows of W = k - p + 1 p-mers) that select the same p-mer
OCCURRENCE, so records per read = 1 + (L - k) x density, density = selected positions per window slide.  The partition must be a
function of the k-mer alone (every instance of a k-mer in the same bin), i.e. a LOCAL scheme on the k-mer's own W p-mers.  For
forward local schemes the density is bounded below by ceil((W + p) / W) / (W + p)  (Kille, Groot Koerkamp et al. 2024) --
2 / (k + 1) whenever W >= (k + 1) / 2 -- against 2 / (W + 1) for random minimizers.  This script measures, on a random sequence:
  random        min of a random hash over the window (what the scan does)
  open-closed   OC minimizer (Groot Koerkamp & Pibiri 2024): prefer p-mers that are open syncmers w.r.t. their t-mers (smallest
                t-mer in the middle), then closed syncmers (smallest t-mer at an end), then the rest; ties by hash
  mod-mini      mod-minimizer with t = p mod W ... (needs p >= W to pay; here p < W, shown for completeness)
and prints records per 150-base read.  Pure numpy; a few seconds."""
import sys
import numpy as np


def sliding_argmin(v, w):
    """leftmost argmin of every window of w values (vectorised over window offsets: fine for w ~ 33)"""
    n = len(v) - w + 1
    best = v[:n].copy()
    arg = np.zeros(n, np.int64)
    for j in range(1, w):
        x = v[j:j + n]
        lt = x < best
        best = np.where(lt, x, best)
        arg = np.where(lt, j, arg)
    return arg + np.arange(n)


def density(pos):
    return float(np.count_nonzero(np.diff(pos)) + 1) / len(pos)


def main():
    k, p, L = (int(sys.argv[1]) if len(sys.argv) > 1 else 47), 15, 150
    W = k - p + 1
    rng = np.random.default_rng(1)
    n = 3_000_000
    seq = rng.integers(0, 4, n).astype(np.uint64)
    def mers(m):
        v = np.zeros(n - m + 1, np.uint64)
        for i in range(m):
            v = (v << np.uint64(2)) | seq[i:n - m + 1 + i]
        return v
    def h(x, salt):
        x = (x ^ np.uint64(salt)) * np.uint64(0x9E3779B97F4A7C15)
        x ^= x >> np.uint64(29)
        x = x * np.uint64(0xD6E8FEB86659FD93)
        return x ^ (x >> np.uint64(32))
    pm = h(mers(p), 7)
    out = {}
    out["random minimizer"] = density(sliding_argmin(pm, W))
    # open-closed: t-mers inside each p-mer
    t = 5
    tm = h(mers(t), 11)
    inner = p - t + 1
    tpos = sliding_argmin(tm, inner) - np.arange(len(tm) - inner + 1)      # offset of the smallest t-mer in each p-mer
    cls = np.full(len(tpos), 2, np.uint64)
    cls[(tpos == 0) | (tpos == inner - 1)] = 1                              # closed syncmer
    cls[tpos == (inner - 1) // 2] = 0                                       # open syncmer (middle offset)
    key = (cls << np.uint64(60)) | (pm[:len(cls)] >> np.uint64(4))
    out["open-closed minimizer (t = 5)"] = density(sliding_argmin(key, W))
    bound = -(-(W + p) // W) / (W + p)
    print("k = %d, p = %d, W = %d, reads of %d bases: records per read = 1 + %d x density" % (k, p, W, L, L - k))
    for name, d in out.items():
        print("  %-32s density %.4f  (x (W+1) = %.2f)   records / read %.2f" % (name, d, d * (W + 1), 1 + (L - k) * d))
    print("  %-32s density %.4f  (x (W+1) = %.2f)   records / read %.2f" % ("lower bound, forward schemes", bound, bound * (W + 1), 1 + (L - k) * bound))


if __name__ == "__main__":
    main()
