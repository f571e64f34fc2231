"""Canonical form of a BaseGraph for comparisons that must not depend on the (unpinned) MPHF
seed order: node permutation, reverse-complement orientation (non-stranded) and the cut point of
isolated cycles are quotiented out ("policy B")."""
import numpy as np
from oracle_lib import unpack_bases
from refgen import kmers_of, canon, exts_rc_py, kmer_rc_int


def node_bases(g, i):
    return unpack_bases(g["words"], int(g["start"][i]), int(g["length"][i]))


def _is_cycle(bases, exts, k, stranded):
    """True when the node's right end feeds its own left end in the same orientation."""
    r = exts >> 4
    l = exts & 0xF
    if bin(r).count("1") != 1 or bin(l).count("1") != 1:
        return False
    ks = kmers_of(bases, k)
    rb = r.bit_length() - 1
    mask = (1 << (2 * k)) - 1
    nxt = ((ks[-1] << 2) | rb) & mask
    lb = l.bit_length() - 1
    prv = (ks[0] >> 2) | (lb << (2 * (k - 1)))
    return nxt == ks[0] and prv == ks[-1]


def canonical_nodes(g, k, stranded):
    """-> sorted list of node keys.  Linear node: (seq tuple oriented to its lexicographic minimum
    when non-stranded, exts in that orientation, data).  Cycle node: ('cycle', sorted canonical
    k-mer tuple, data)."""
    out = []
    n = len(g["start"])
    for i in range(n):
        b = [int(x) for x in node_bases(g, i)]
        e = int(g["exts"][i])
        d = int(g["data"][i])
        if _is_cycle(b, e, k, stranded):
            ks = kmers_of(b, k)
            cs = tuple(sorted(ks if stranded else [canon(k, v) for v in ks]))
            out.append(("cycle", cs, d))
            continue
        fwd = tuple(b)
        if stranded:
            out.append((fwd, e, d))
        else:
            rc = tuple(3 - x for x in reversed(b))
            if rc < fwd:
                out.append((rc, exts_rc_py(e), d))
            elif rc == fwd:
                out.append((fwd, min(e, exts_rc_py(e)), d))
            else:
                out.append((fwd, e, d))
    out.sort(key=repr)
    return out


def graph_kmer_set(g, k, stranded):
    s = set()
    for i in range(len(g["start"])):
        for v in kmers_of(node_bases(g, i), k):
            s.add(v if stranded else canon(k, v))
    return s


def graphs_equal(a, b):
    """Literal BaseGraph equality: same packed words, start, length, exts, data (policy A)."""
    if a["n_bases"] != b["n_bases"]:
        return False
    nw = (a["n_bases"] + 31) // 32
    return (np.array_equal(a["words"][:nw], b["words"][:nw]) and np.array_equal(a["start"], b["start"])
            and np.array_equal(a["length"], b["length"]) and np.array_equal(a["exts"], b["exts"])
            and np.array_equal(a["data"], b["data"]))
