"""CPU checker engine for rust-debruijn_amd.distributed (TEST INFRASTRUCTURE ONLY): the same stage
interface as HipEngine, implemented with the oracle, so that the torch.distributed orchestration
(bin ownership, split sizes, the two all-to-alls, the segment table) runs on CPU with gloo."""
import numpy as np
import torch

import oracle_lib as O

RW = 4            # 3 base words + 1 meta word, like the product's k=47 record
P = 8


class Plan:
    def __init__(self, k, stranded, kind, min_obs, total, n_bins=None):
        self.k, self.stranded, self.summarizer, self.min_kmer_obs = k, stranded, kind, min_obs
        self.n_bins = n_bins or max(4, total // 400)
        self.rec_words = RW


class OracleEngine:
    device = torch.device("cpu")

    def __init__(self, n_bins=None):
        self.n_bins = n_bins                      # force a (tiny) bin count: ranks then own different numbers of bins

    def count_instances(self, ss, k):
        return int(sum(max(0, int(l) - k + 1) for l in ss.length))

    def max_label(self, ss):
        return int(ss.data.max()) if getattr(ss, "data", None) is not None and len(ss.data) else 0

    def label_presence(self, ss):
        pres = np.zeros(65537, np.uint8)
        if getattr(ss, "data", None) is not None:
            pres[np.minimum(np.asarray(ss.data, dtype=np.int64), 65536)] = 1
        return pres

    def plan(self, k, stranded, kind, min_obs, total, max_label=0, merge_dups=False, labels=None):
        self.labels = labels                      # (the oracle counts labels as they are: the list is only recorded)
        return Plan(k, stranded, kind, min_obs, total, self.n_bins)

    def scan(self, ss, plan):
        rows, bins = [], []
        for i in range(ss.n):
            seq = ss.bases(i)
            if len(seq) < plan.k:
                continue
            d = int(ss.data[i]) if ss.data is not None else 0
            bu, ex, st, ln = O.msp_sequence(seq, plan.k, P, None, rc=not plan.stranded)
            for b, e, s, l in zip(bu, ex, st, ln):
                s, l = int(s), int(l)
                e = int(e)
                if s == 0:
                    e |= int(ss.exts[i]) & 0x0F
                if s + l == len(seq):
                    e |= int(ss.exts[i]) & 0xF0
                w = list(O.pack_bases(seq[s:s + l])) + [0, 0, 0]
                rows.append([int(w[0]), int(w[1]), int(w[2]), l | (e << 8) | (d << 16)])
                bins.append(((int(b) * 2654435761) % (1 << 32)) * plan.n_bins >> 32)
        self._rows = np.array(rows, dtype=np.uint64).reshape(-1, RW)
        self._bins = np.array(bins, dtype=np.int64)
        hist = np.bincount(self._bins, minlength=plan.n_bins)
        bin_off = np.concatenate([[0], np.cumsum(hist)]).astype(np.int64)
        return torch.from_numpy(bin_off), len(rows)

    def scatter(self, plan, bin_off, n_recs):
        """records of bin b go to [bin_off[b], bin_off[b] + count_b): any layout the caller asks for"""
        off = bin_off.numpy()
        order = np.argsort(self._bins, kind="stable")
        sb = self._bins[order]
        first = np.searchsorted(sb, sb, side="left")                 # position of each record's bin in the sorted run
        dest = off[sb] + (np.arange(len(sb)) - first)
        out = np.zeros((len(sb), RW), dtype=np.uint64)
        out[dest] = self._rows[order]
        return torch.from_numpy(out.astype(np.int64).reshape(-1))

    # chunked form used by the pipelined exchange
    def sync(self):
        pass

    def count_begin(self, plan, hint):
        self._acc = ([], [], [])

    def count_bins(self, plan, recs, seg_off, n_src, nb_chunk, units=0):
        r = recs.numpy().astype(np.uint64).reshape(-1, RW)[: int(seg_off.max())] if recs.numel() >= RW else np.zeros((0, RW), np.uint64)
        seg = seg_off.numpy()
        assert seg.shape == (n_src, nb_chunk + 1)
        used = np.zeros(len(r), dtype=bool)
        for s in range(n_src):
            for b in range(nb_chunk):
                for j in range(int(seg[s, b]), int(seg[s, b + 1])):
                    assert not used[j]
                    used[j] = True
                    meta = int(r[j, 3])
                    l = meta & 0xFF
                    self._acc[0].append(O.unpack_bases(r[j, :3], 0, l))
                    self._acc[1].append((meta >> 8) & 0xFF)
                    self._acc[2].append(meta >> 16)
        assert used.all()                       # every received record belongs to exactly one bin segment of the chunk

    def count_finish(self, plan):
        seqs, exts, data = self._acc
        ss = O.SeqSet.from_byte_seqs(seqs, exts=exts, data=data if plan.summarizer == O.COUNT_FILTER_SET else None,
                                     sizeof_d1=1)
        return O.filter_kmers(ss, plan.k, plan.summarizer, plan.min_kmer_obs, stranded=plan.stranded)

    def count(self, plan, recs, seg_off, n_src, nb_local, hint):
        r = recs.numpy().astype(np.uint64).reshape(-1, RW)
        seg = seg_off.numpy()
        assert seg.shape == (n_src, nb_local + 1)
        used = np.zeros(len(r), dtype=bool)
        seqs, exts, data = [], [], []
        for s in range(n_src):
            for b in range(nb_local):
                for j in range(int(seg[s, b]), int(seg[s, b + 1])):
                    assert not used[j]
                    used[j] = True
                    meta = int(r[j, 3])
                    l = meta & 0xFF
                    seqs.append(O.unpack_bases(r[j, :3], 0, l))
                    exts.append((meta >> 8) & 0xFF)
                    data.append(meta >> 16)
        assert used.all()                       # every received record belongs to exactly one owned bin segment
        ss = O.SeqSet.from_byte_seqs(seqs, exts=exts, data=data if plan.summarizer == O.COUNT_FILTER_SET else None,
                                     sizeof_d1=1)
        return O.filter_kmers(ss, plan.k, plan.summarizer, plan.min_kmer_obs, stranded=plan.stranded)

    # ---- rank-spanning compress stage (distributed.sharded_compress) with the oracle as the engine ----
    @staticmethod
    def _to_base_graph(og, k, stranded, classes=None):
        from pkg import dbg
        a = og.arrays()
        g = dbg.BaseGraph(k, dbg.PackedDnaStringSet(a["words"][:a["n_words"]], a["start"], a["length"], a["n_bases"]), a["exts"], a["data"],
                          stranded)
        g.classes = classes
        return g

    @staticmethod
    def _to_oracle_graph(g):
        a = g.arrays()
        return O.graph_from_arrays(g.k, g.stranded, np.concatenate([a["words"], np.zeros(2, np.uint64)]), a["start"], a["length"],
                                   a["exts"], a["data"])

    def compress_table(self, tab, k, stranded, spec):
        classes, data = None, tab.count
        if spec.kind == O.SPEC_SCMAP_EQ:                       # label lists -> rank-local class ids (any injective numbering)
            sets = [tuple(int(x) for x in tab.set_val[int(tab.set_off[i]):int(tab.set_off[i + 1])]) for i in range(tab.n)]
            classes = sorted(set(sets), reverse=True)          # deliberately NOT the product's order: ids are rank-local
            pos = {t: i for i, t in enumerate(classes)}
            data = np.array([pos[t] for t in sets], dtype=np.uint32)
        og = O.compress_kmers(k, stranded, spec.kind, tab.key_hi, tab.key_lo, tab.exts, data)
        return self._to_base_graph(og, k, stranded, classes)

    def combine(self, graphs):
        og = O.graph_combine([self._to_oracle_graph(g) for g in graphs])
        return self._to_base_graph(og, graphs[0].k, graphs[0].stranded)

    def compress_graph(self, stranded, spec, graph):
        og = self._to_oracle_graph(graph).finish().compress_graph(stranded, spec.kind)
        return self._to_base_graph(og, graph.k, stranded)
