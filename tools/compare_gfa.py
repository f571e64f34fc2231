#!/usr/bin/env python3
"""Compare two GFA files written by DebruijnGraph::write_gfa (src/graph.rs:537-616) -- one by the `debruijn` crate, one by this
library (dbg_graph_write_gfa) -- up to the freedoms the crate itself does not pin: the order of the
nodes (= the MPHF slot order of boomphf, Cargo.toml:27-30), the strand each unitig is written in (non-stranded graphs) and the
k-mer at which an isolated cycle is cut.  Everything else must agree: the set of unitig sequences and the set of links between
unitig ends.

    python tools/compare_gfa.py crate.gfa ours.gfa [--stranded] [--k K]

exit status 0 = same graph, 1 = different (differences are listed), 2 = malformed input.  Pure Python, no GPU: this is the
route to a literal cross-implementation check on a machine that has the Rust toolchain."""
import argparse
import collections
import sys

COMP = str.maketrans("ACGTacgt", "TGCAtgca")


def rc(s):
    return s.translate(COMP)[::-1]


def parse(path):
    """-> (segments {id: sequence}, links [(a, a_orient, b, b_orient, overlap)])"""
    segs, links = {}, []
    with open(path) as f:
        for ln, line in enumerate(f, 1):
            t = line.rstrip("\n").split("\t")
            if not t or t[0] in ("H", ""):
                continue
            if t[0] == "S":
                if len(t) < 3 or t[1] in segs:
                    raise ValueError("%s:%d: bad or repeated S line" % (path, ln))
                segs[t[1]] = t[2].upper()
            elif t[0] == "L":
                if len(t) < 6 or t[2] not in "+-" or t[4] not in "+-":
                    raise ValueError("%s:%d: bad L line" % (path, ln))
                links.append((t[1], t[2], t[3], t[4], t[5]))
    for a, _, b, _, _ in links:
        if a not in segs or b not in segs:
            raise ValueError("%s: link to a segment that is not in the file" % path)
    return segs, links


def canonical(path, stranded, k):
    """-> (multiset of node keys, multiset of link keys).  A node key is its sequence in the orientation that compares smaller
    (non-stranded), or ('cycle', sorted canonical k-mers) for an isolated cycle; a link key is the unordered pair of the two
    (node key, end) it joins, ends named in the node's canonical orientation."""
    segs, links = parse(path)
    if k is None:
        ovl = {l[4] for l in links}
        if len(ovl) > 1:
            raise ValueError("%s: links with different overlaps %s" % (path, sorted(ovl)))
        k = int(next(iter(ovl))[:-1]) + 1 if ovl else None
    # ends: a '+' source leaves through its right end, a '-' source through its left end; a '+' target is entered at its left
    # end, a '-' target at its right end (graph.rs:561-590: l_edges are written with the source as '-')
    touching = collections.defaultdict(list)
    for a, ao, b, bo, _ in links:
        touching[a].append((b, ao, bo))
        touching[b].append((a, bo, ao))
    key, flip = {}, {}
    for sid, s in segs.items():
        is_cycle = False
        if k is not None and len(s) >= k and touching[sid] and all(o == sid for o, _, _ in touching[sid]):
            # linked to nothing but itself, right end into left end in the same orientation: an isolated cycle -- where it is cut
            # (and, non-stranded, in which direction it is read) is the implementation's choice
            own = [(ao, bo) for a, ao, b, bo, _ in links if a == sid and b == sid]
            closes = s[:k - 1] == s[len(s) - (k - 1):] if k > 1 else True
            is_cycle = closes and all(ao == bo for ao, bo in own)
        if is_cycle:
            kms = [s[i:i + k] for i in range(len(s) - k + 1)]
            if not stranded:
                kms = [min(x, rc(x)) for x in kms]
            key[sid], flip[sid] = ("cycle", tuple(sorted(kms))), False
        elif stranded:
            key[sid], flip[sid] = s, False
        else:
            r = rc(s)
            key[sid], flip[sid] = (r, True) if r < s else (s, False)
    nodes = collections.Counter(key.values())
    lk = collections.Counter()
    for a, ao, b, bo, _ in links:
        if isinstance(key[a], tuple):                     # the self link of an isolated cycle carries no information beyond the node
            continue
        ea = "R" if (ao == "+") != flip[a] else "L"
        eb = "L" if (bo == "+") != flip[b] else "R"
        if not stranded and key[a] == rc(key[a]):         # a palindromic unitig has indistinguishable ends
            ea = "*"
        if not stranded and key[b] == rc(key[b]):
            eb = "*"
        lk[tuple(sorted([(key[a], ea), (key[b], eb)]))] += 1
    return nodes, lk, k


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("a")
    ap.add_argument("b")
    ap.add_argument("--stranded", action="store_true", help="the graphs are stranded: sequences are compared as written")
    ap.add_argument("--k", type=int, default=None, help="k-mer length (default: overlap of the L lines + 1)")
    ap.add_argument("--show", type=int, default=10, help="differences listed per category")
    args = ap.parse_args(argv)
    try:
        na, la, ka = canonical(args.a, args.stranded, args.k)
        nb, lb, kb = canonical(args.b, args.stranded, args.k)
    except (ValueError, OSError) as e:
        print("error:", e, file=sys.stderr)
        return 2
    bad = 0
    if ka is not None and kb is not None and ka != kb:
        print("k differs: %d vs %d" % (ka, kb))
        bad = 1
    for what, x, y in (("unitigs", na, nb), ("links", la, lb)):
        only_a, only_b = x - y, y - x
        print("%s: %d vs %d, %d only in %s, %d only in %s" % (what, sum(x.values()), sum(y.values()), sum(only_a.values()), args.a,
                                                               sum(only_b.values()), args.b))
        for side, d in ((args.a, only_a), (args.b, only_b)):
            for i, (kk, c) in enumerate(sorted(d.items(), key=repr)):
                if i >= args.show:
                    break
                print("  only in %s (x%d): %s" % (side, c, (repr(kk)[:200])))
        bad |= bool(only_a or only_b)
    print("DIFFERENT" if bad else "SAME GRAPH")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
