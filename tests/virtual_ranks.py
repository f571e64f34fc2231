"""Several virtual ranks on one GPU (TEST INFRASTRUCTURE): every rank's reads are scanned into the bins of one global plan
by the real kernels, the all-to-all is replaced by tensor slicing, and each virtual owner counts its bins from `world`
bin-ordered segments -- exactly what a rank sees after the RCCL exchange."""
import importlib

import numpy as np
import torch

from pkg import dbg

D = importlib.import_module("rust-debruijn_amd.distributed")


def dev_seqset(engine, hs, with_data):
    dev = engine.device
    words = torch.from_numpy(hs.words.view(np.int64)).to(dev)
    start = torch.from_numpy(hs.start.view(np.int64)).to(dev)
    length = torch.from_numpy(hs.length.view(np.int32)).to(dev)
    data = torch.from_numpy(hs.data.view({1: np.uint8, 2: np.int16, 4: np.int32}[hs.data.dtype.itemsize])).to(dev) if with_data else None
    return engine.seqset(words, start, length, data, hs.data.dtype.itemsize if with_data else 0), (words, start, length, data)


def owner_tables(eng, host_shards, k, stranded, kind, min_obs, merge_dups=True, with_records=False):
    """host_shards: one HostSeqs per virtual rank -> (list of device tables, one per owner; total k-mer instances[; records
    every rank sends, with_records]).  merge_dups: the senders merge their identical records (dbg_shard_plan.merge_dups).
    The caller releases the tables with eng.free_table."""
    world = len(host_shards)
    shards, keep = [], []
    for hs in host_shards:
        ss, kp = dev_seqset(eng, hs, kind == 1)
        shards.append(ss)
        keep.append(kp)
    total = sum(eng.count_instances(s, k) for s in shards)
    max_label = max(eng.max_label(s) for s in shards) if kind == 1 else 0          # (all-reduce MAX in the real flow)
    labels = None
    if max_label >= 64:                                                             # sparse alphabet: union over the ranks
        pres = np.maximum.reduce([eng.label_presence(s) for s in shards])
        assert not pres[65536]
        labels = np.nonzero(pres[:65536])[0].tolist()
    plan = eng.plan(k, stranded, kind, min_obs, total, max_label, merge_dups=merge_dups, labels=labels)
    rw, nb = plan.rec_words, plan.n_bins
    bounds = D.owner_bounds(nb, world, plan.bin_group)
    scanned, n_recs = [], []
    for s in shards:
        bin_off, n = eng.scan(s, plan)
        n_recs.append(n)
        scanned.append((bin_off, eng.scatter(plan, bin_off, n)))
    tabs = []
    for owner in range(world):
        lo, hi = bounds[owner], bounds[owner + 1]
        nb_local = hi - lo
        slabs, hists = [], []
        for bin_off, recs in scanned:
            a, b = int(bin_off[lo]), int(bin_off[hi])
            slabs.append(recs[a * rw:b * rw])
            hists.append(bin_off[lo + 1:hi + 1] - bin_off[lo:hi])
        recv = torch.cat(slabs) if sum(len(s) for s in slabs) else torch.zeros(1, dtype=torch.int64, device=eng.device)
        seg_off = torch.zeros(world, nb_local + 1, dtype=torch.int64, device=eng.device)
        base = 0
        for s in range(world):
            seg_off[s, 1:] = torch.cumsum(hists[s], 0)
            seg_off[s] += base
            base += len(slabs[s]) // rw
        tabs.append(eng.count(plan, recv, seg_off, world, nb_local, total))
    return (tabs, total, n_recs) if with_records else (tabs, total)
