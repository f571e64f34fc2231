#!/usr/bin/env python3
"""bench.py -- k-mer extract -> MSP shard -> count/filter throughput on MI355X.

One "step" = one pass of the hot path (dbg_filter_kmers_dev, device-resident in and out) over one
batch of synthetic reads already in HBM (SURVEY.md section 8d stream: splitmix64 genome at 30x,
150-bp reads, substitution rate 0.001, random strand).  Metric: k-mer instances per second
(BASELINE.json: "Gkmer/s extracted+counted (k=47, 150 bp synthetic reads)").

    python bench.py --gpus N --steps K --warmup W [--reads R] [--k 47]

N > 1 is launched by torch.distributed.run, one rank per GPU; see DESIGN.md (multi-GPU).
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E peak (MI355X_MICROARCH.md)
# Random device-scope atomic adds (returning or not, any scope, any counter layout) complete at 2.7e10 per second on this device:
# tools/micro/atomic_stride.hip, atomic_scope.hip, atomic_noret.hip -> profiles/r04_micro_benchmarks.txt.  The scan reserves one slab
# slot per super-k-mer record with such an atomic: that rate, not HBM, is its roof.
ATOMIC_PEAK_PER_S = 2.7e10


def alg_bytes_per_kmer(k, read_len, is_set, u_over_n):
    """SURVEY.md section 8(d) two-touch model: B_in + 2R + (U/N) B_out."""
    key = 8 if k <= 32 else 16
    b_in = (read_len / 4.0) / (read_len - k + 1)
    r = key + 1 + (1 if is_set else 0)
    b_out = key + 1 + 2
    return b_in + 2 * r + u_over_n * b_out


# the files that hold the kernels of the timed path and of the shapes reported next to it (host-side files -- api, transports, graph,
# the rank-spanning flows -- do not change what the counters measured); tools/pmc_traffic.py hashes the same list
KERNEL_SOURCES = ("dbg_device.hpp", "dbg_msp_device.hpp", "fast_manylabels.hpp", "fastpath.hip", "radix.hip", "scan.hip",
                  "fast_labellists.hpp", "densepath.hip")


def kernel_source_sha16():
    import hashlib
    h = hashlib.sha256()
    cdir = os.path.join(ROOT, "rust-debruijn_amd", "csrc")
    for f_ in KERNEL_SOURCES:
        h.update(f_.encode()); h.update(open(os.path.join(cdir, f_), "rb").read())
    return h.hexdigest()[:16]


def pmc_kernel_names(k):
    """timing name of the library -> kernel name(s) in a rocprofv3 counter file"""
    return {"sk_scan": "sk_scan_lane_kernel", "sk_scan_long": "sk_scan_kernel", "bin_count": "bin_count_kernel",
            # (round 3: look-back passes -- one up-front histogram kernel, the scatter kernel does its own offsets; the classic
            #  kernels remain as DBG_ONESWEEP=0 and as the fall-back)
            "radix_hist": (("radix16_global_hist_kernel", "radix16_hist_kernel") if 2 * k <= 96
                           else ("radix_global_hist_kernel", "radix_hist_kernel")),
            "radix_scatter": (("radix16_onesweep_kernel", "radix16_scatter_kernel") if 2 * k <= 96 else ("radix_scatter_kernel",)),
            "span_sort": (("span_sort16_groups_kernel", "span_sort16_kernel") if 2 * k <= 96
                          else ("span_sort_groups_kernel", "span_sort_kernel")), "set_csr": ("csr_apply_kernel", "ll_csr_kernel"),
            "sk_scatter": "sk_scatter_kernel", "slab_compact": "slab_compact_kernel", "bin_labels": "bin_labels_kernel",
            "list_meta": "ll_meta_kernel"}


def shape_kernel_rows(shape, k, kernel_ms, n_inst, n_recs, src_sha):
    """other_shapes: per-kernel rows with the counter evidence of profiles/r*_pmc_shape_<shape>.json (tools/r06_pmc_shapes.sh: measured HBM bytes, VALU busy, the binding
    resource), hash-guarded like the main rows"""
    import glob
    tf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_shape_%s.json" % shape)))
    tj = json.load(open(tf[-1])) if tf else {}
    stale = bool(tj) and tj.get("_kernel_source_sha16") != src_sha
    names = pmc_kernel_names(k)
    rows = []
    for nm, ms in sorted(kernel_ms.items(), key=lambda kv: -kv[1]):
        pn = names.get(nm, nm)
        ent = next((tj[c] for c in (pn if isinstance(pn, tuple) else (pn,)) if c in tj), None)
        r = {"kernel": nm, "ms_per_step": ms, "measured_hbm_frac": None, "bound_resource": None, "bound_frac": None}
        if nm in ("sk_scan", "sk_scan_long") and n_recs and ms > 0:
            r["bound_resource"] = "global slot atomics (%.1e/s device rate)" % ATOMIC_PEAK_PER_S
            r["bound_frac"] = round(n_recs / (ms * 1e-3) / ATOMIC_PEAK_PER_S, 4)
        if ent and ms > 0:
            tb = ent["bytes_per_instance"] * n_inst
            r["traffic_bytes_per_step"] = round(tb, 0)
            r["measured_hbm_frac"] = round(tb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            if r["bound_resource"] is None:
                if ent.get("valu_busy") is not None and r["measured_hbm_frac"] < 0.25:
                    r["valu_busy"], r["lds_active"], r["wave_wait_share"] = ent.get("valu_busy"), ent.get("lds_active"), ent.get("wave_wait_share")
                    r["bound_resource"], r["bound_frac"] = "VALU issue next to dependent LDS round trips / barriers", ent.get("valu_busy")
                else:
                    r["bound_resource"], r["bound_frac"] = "HBM bandwidth", r["measured_hbm_frac"]
        rows.append(r)
    return {"kernels": rows, "traffic_source": os.path.basename(tf[-1]) if tf else None, "traffic_stale": stale if tf else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=int(os.environ.get("DBG_BENCH_READS", 0)))
    ap.add_argument("--config", default="c2", choices=["c2", "c4", "c5"],
                    help="c2 (default): BASELINE configs[1], 10^8 reads PER GPU, k = 47 -- weak scaling.  c4 / c5: the BASELINE shapes that "
                         "name a TOTAL size, strong-scaled over the ranks: c4 = 10^9 reads, k = 63; c5 = 6*10^8 reads (3 Gbp x 30), k = 51, "
                         "CountFilterSet (the index ScmapCompress works on).  --reads then means total reads")
    ap.add_argument("--k", type=int, default=47)
    ap.add_argument("--summarizer", default="set", choices=["set", "count"])
    ap.add_argument("--min-obs", type=int, default=2)
    ap.add_argument("--error-rate", type=float, default=0.001, help="substitution error rate of the synthetic reads (SURVEY 8d: 0.001, also 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pmc-json", default=None,
                    help="HBM traffic per kernel from a counter pass of this same tree (tools/pmc_round.sh -> tools/pmc_traffic.py); default: "
                         "the newest profiles/r*_pmc_traffic*.json.  A file measured on other kernel sources is flagged traffic_stale")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --one-device: several ranks on ONE GPU, payload staged through the host (a functional check of the "
                         "N > 1 orchestration on a single-GPU box; its timings mean nothing)")
    ap.add_argument("--one-device", action="store_true", help="every rank uses cuda:0")
    ap.add_argument("--force-sharded", action="store_true",
                    help="time the multi-GPU pipeline (scan -> compaction -> exchange -> count) even on one GPU")
    ap.add_argument("--force-exchange", action="store_true",
                    help="with one rank: initialise a 1-rank process group and run the pipelined all-to-all route anyway (exercises "
                         "the RCCL calls of the N > 1 path on a one-GPU box; implies --force-sharded)")
    ap.add_argument("--digest", action="store_true",
                    help="report an order-independent digest of the result table(s): per-rank digests of a sharded run add up to the "
                         "single-GPU digest over the same reads (tools/check_multirank.sh)")
    ap.add_argument("--no-host-boundary", action="store_true",
                    help="skip the host-pointer measurement (dbg_filter_kmers: host arrays in, host table out -- the reference's own boundary)")
    ap.add_argument("--no-other-shapes", action="store_true", help="skip the secondary shapes (k = 31, 63, 24; CountFilter) timed on the same reads")
    ap.add_argument("--compress-reads", type=int, default=-1,
                    help="reads of the stream used for the secondary unitigs/s measurement (0 = skip, -1 = all: BASELINE config 3)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started by hand as `python bench.py --gpus N`: become the launcher the driver would have used, one rank per GPU
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)

    import numpy as np
    import torch
    dbg = importlib.import_module("rust-debruijn_amd")
    capi = importlib.import_module("rust-debruijn_amd._capi")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.one_device:
        local_rank = 0
    elif world > torch.cuda.device_count():
        sys.exit("bench.py: %d ranks but only %d GPU(s) visible (--one-device --backend gloo shares cuda:0 for a functional check)"
                 % (world, torch.cuda.device_count()))
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    elif args.force_exchange:
        import socket
        import torch.distributed as dist
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        kw = dict(device_id=torch.device("cuda", local_rank)) if args.backend == "nccl" else {}
        dist.init_process_group(args.backend, init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, **kw)
        args.force_sharded = True
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ctx = dbg.Context(local_rank)
    lib = ctx.lib

    k, L = args.k, 150
    strong = args.config != "c2"
    if strong:                                          # a BASELINE shape with a total size: split over the ranks
        k = {"c4": 63, "c5": 51}[args.config] if args.k == 47 else args.k
        total_reads = args.reads or {"c4": 1_000_000_000, "c5": 600_000_000}[args.config]
        reads_per_gpu = total_reads // world
    else:
        reads_per_gpu = args.reads or 100_000_000      # BASELINE configs[1]: 100M x 150 bp, k=47, CountFilterSet
    is_set = args.summarizer == "set"
    n_reads_total = reads_per_gpu * world               # weak scaling: fixed reads per GPU (strong: the total split evenly)
    genome_len = n_reads_total * L // 30

    # ---- synthetic input, generated directly in HBM ----
    p = dbg.synth_params(n_reads=reads_per_gpu, read_len=L, genome_len=genome_len, error_rate=args.error_rate,
                         stranded=False, n_colours=4, first_read=rank * reads_per_gpu)
    nw = lib.dbg_synth_words(C.byref(p))
    words = torch.empty(nw, dtype=torch.int64, device=dev)
    start = torch.empty(reads_per_gpu, dtype=torch.int64, device=dev)
    length = torch.empty(reads_per_gpu, dtype=torch.int32, device=dev)
    colour = torch.empty(reads_per_gpu, dtype=torch.uint8, device=dev)
    ctx.check(lib.dbg_synth_reads_dev(ctx.h, C.byref(p), words.data_ptr(), start.data_ptr(), length.data_ptr(),
                                      colour.data_ptr()))
    ss = capi.SeqSet(words.data_ptr(), nw, start.data_ptr(), length.data_ptr(), None,
                     colour.data_ptr() if is_set else None, 1 if is_set else 0, reads_per_gpu)
    fp = capi.FilterParams(k, 0, 1 if is_set else 0, args.min_obs, 0, 4)

    D = importlib.import_module("rust-debruijn_amd.distributed")
    engine = D.HipEngine(ctx, dev)
    xstats = {}

    def step():
        if world == 1 and not args.force_sharded:
            # the drop-in entry point: filter_kmers, device-resident in and out
            t = capi.KmerTable()
            ctx.check(lib.dbg_filter_kmers_dev(ctx.h, C.byref(ss), C.byref(fp), C.byref(t)))
            res = (t.n, t.n_kmer_instances, D.table_digest(t, dev) if args.digest else 0)
            lib.dbg_free_table(ctx.h, C.byref(t))
            return res
        # N > 1: scan local reads -> all-to-all of minimizer-bin slabs (RCCL over xGMI) -> count owned bins
        st = {}
        tab, total, n_local, n_recs = D.sharded_filter_kmers(engine, ss, k, False, 1 if is_set else 0, args.min_obs, stats=st,
                                                                 force_exchange=args.force_exchange)
        for kk, v in st.items():
            if isinstance(v, (str, bool)) or v is None:
                xstats[kk] = v                                   # (transport name, merge / balance flags: not summed over steps)
            elif isinstance(v, list):
                old = xstats.get(kk, [0.0] * len(v))
                xstats[kk] = [a_ + b_ for a_, b_ in zip(old, v)] if len(old) == len(v) else list(v)
            else:
                xstats[kk] = xstats.get(kk, 0) + v
        res = (tab.n, n_local, D.table_digest(tab, dev) if args.digest else 0)
        engine.free_table(tab)
        return res

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    xstats.clear()
    ctx.enable_timing(True)
    barrier()
    t0 = time.perf_counter()
    n_valid = n_inst = digest = 0
    ktimes = {}
    kseries = {}                                        # per-step HIP-event ms of each kernel: a bimodal scan shows here, not in a mean
    for _ in range(args.steps):
        n_valid, n_inst, digest = step()
        for kt in ctx.timings():
            a = ktimes.setdefault(kt["name"], dict(ms=0.0, launches=0, units=0))
            a["ms"] += kt["ms"]; a["launches"] += kt["launches"]; a["units"] += kt["units"]
            kseries.setdefault(kt["name"], []).append(round(kt["ms"], 3))
    barrier()
    dt = time.perf_counter() - t0
    ctx.enable_timing(False)
    if world > 1:
        import torch.distributed as dist
        rdev = dev if args.backend == "nccl" else torch.device("cpu")
        tt = torch.tensor([dt], dtype=torch.float64, device=rdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        cnt = torch.tensor([n_inst, n_valid, 1, int(xstats.get("exchange_bytes_sent", 0))], dtype=torch.int64, device=rdev)
        dist.all_reduce(cnt)
        n_inst_total = int(cnt[0].item())
        n_valid_total = int(cnt[1].item())
        ranks_seen = int(cnt[2].item())
        xbytes_total = int(cnt[3].item())
        xe = torch.tensor([xstats.get("exchange_exposed_ms", 0.0)], dtype=torch.float64, device=rdev)
        dist.all_reduce(xe, op=dist.ReduceOp.MAX)
        exposed_ms = float(xe.item())
        # per-rank balance: what a 1 -> 8 curve needs to be diagnosed from the record alone
        kms = lambda *names: sum(ktimes.get(n_, {}).get("ms", 0.0) for n_ in names) / max(args.steps, 1)
        mine = torch.tensor([xstats.get("records_owned", 0) / max(args.steps, 1), kms("bin_count"), kms("sk_scan", "sk_scan_long"),
                             kms("slab_compact", "sk_scatter"), kms("radix_hist", "radix_scatter", "span_sort", "set_csr"),
                             xstats.get("exchange_exposed_ms", 0.0) / max(args.steps, 1), float(n_valid)], dtype=torch.float64, device=rdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = torch.stack(allr).cpu()
        bal_names = ["records_owned", "bin_count_ms", "sk_scan_ms", "compact_ms", "sort_ms", "exchange_exposed_ms", "valid_kmers"]
        balance = {nm: {"max": round(float(allr[:, i].max()), 3), "mean": round(float(allr[:, i].mean()), 3),
                        "min": round(float(allr[:, i].min()), 3)} for i, nm in enumerate(bal_names)}
        rounds = xstats.get("exchange_exposed_ms_by_round", [])
        if rounds:
            rr = torch.tensor(rounds, dtype=torch.float64, device=rdev) / max(args.steps, 1)
            dist.all_reduce(rr, op=dist.ReduceOp.MAX)
            balance["exchange_exposed_ms_by_round_max_rank"] = [round(float(x), 3) for x in rr.cpu()]
        if args.digest:                                    # sum of the per-rank digests mod 2^64 (two 32-bit halves: no overflow)
            dg = torch.tensor([digest & 0xFFFFFFFF, digest >> 32], dtype=torch.int64, device=rdev)
            dist.all_reduce(dg)
            digest = (int(dg[0].item()) + (int(dg[1].item()) << 32)) & ((1 << 64) - 1)
    else:
        n_inst_total = n_inst
        n_valid_total = n_valid
        ranks_seen, xbytes_total, exposed_ms = 1, 0, 0.0
        balance = None
    ms_per_step = dt / args.steps * 1e3
    value = n_inst_total * args.steps / dt / 1e9
    # how the scan's slab is backed and how fast it takes random record writes (after the timed region: the slab sits in the pool)
    slab = None
    try:
        cst = ctx.stats()
        pr = ctx.probe_slab()
        slab = {"backing": cst["slab_backing_name"], "gb": round(cst["slab_bytes"] / 2**30, 2), "note": cst["slab_note"] or None,
                # slab tournament: first-scan ms of the candidate blocks the first calls tried; the fastest one stayed (it needs DBG_SLAB_TRIALS = 4
                # calls: --warmup >= 4 keeps the trials out of the timed steps)
                "tournament_scan_ms": cst["slab_trial_ms"], "tournament_settled": cst["slab_candidates_pooled"] <= 1 and cst["slab_trials_done"] >= 4,
                "probe_ms": round(pr[0], 3) if pr else None, "probe_writes": pr[1] if pr else None,
                "probe_note": "2^27 record-sized random writes into the pooled slab, best of 3 (26 GB, 24-byte records: 5.4 ms in a well-placed block .. 7.4 in a badly placed one)",
                "alloc": {kk: (round(v, 4) if isinstance(v, float) else v) for kk, v in cst.items()
                          if kk.startswith(("n_", "s_", "pooled"))}}
    except Exception as e:                               # (never let a diagnostic take the bench line down)
        slab = {"error": str(e)}

    out = None
    if rank == 0:
        u_over_n = n_valid / max(n_inst, 1)
        b_alg = alg_bytes_per_kmer(k, L, is_set, u_over_n)
        # dominant kernel = largest share of measured HIP-event time
        n_recs = ktimes.pop("sk_records", {}).get("units", 0) / max(args.steps, 1)
        fast = "bin_count" in ktimes
        dom = max(ktimes.items(), key=lambda kv: kv[1]["ms"]) if ktimes else None
        roof = None
        if dom:
            key = 8 if k <= 32 else 16
            rbytes = key + 4                                   # record = key + 4-byte payload in this build
            pint = 15 if k >= 30 else (14 if k >= 26 else 13)         # fast_internal_p (fastpath.hip)
            rec_b = 8 * max(2, (2 * (2 * k - pint) + 20 + 63) // 64)   # super-k-mer record bytes (bases + 20 meta bits)
            b_in = (L / 4.0) / (L - k + 1)
            sk_b = n_recs * rec_b / max(n_inst, 1)             # super-k-mer bytes per k-mer instance
            r_alg = key + 1 + (1 if is_set else 0)             # SURVEY 8(d) record R
            # algorithmic bytes per unit of each kernel (unit = k-mer instance for the scan and the
            # counting kernel, valid k-mer for the order-restoring stages)
            sort_rec = 16 if 2 * k <= 96 else rbytes
            alg = {"extract_kmers": b_in + rbytes, "radix_scatter": 2 * sort_rec, "radix_hist": 4 if 2 * k <= 96 else 8,
                   "reduce_groups": rbytes + u_over_n * (key + 3), "sk_scan": b_in + sk_b, "sk_scan_long": b_in + sk_b,
                   "sk_scatter": 2 * rec_b + 4, "slab_compact": 2 * rec_b, "bin_count": r_alg + u_over_n * (key + 3),
                   "span_sort": sort_rec + 16 + 1 + (4 if is_set else 2), "set_csr": 4 + 8 + 4 * 1.5}
            # HBM traffic per k-mer instance from the PMC passes (tools/pmc.sh: separate --pmc runs of this same bench at
            # 10M reads; FETCH_SIZE doubled per the gfx950 correction): newest profiles/r*_pmc_traffic*.json
            import glob
            tj, tf = {}, sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")))
            if args.pmc_json:
                tf = [args.pmc_json]
            if tf:
                tj = json.load(open(tf[-1]))
            # the counter passes cannot run inside the timed bench (rocprofv3 wraps the process): the traffic file carries the hash
            # of the kernel sources it was measured on, and a file from other sources is reported as stale
            src_sha = kernel_source_sha16()
            traffic_stale = bool(tj) and tj.get("_kernel_source_sha16") != src_sha
            pmc_name = pmc_kernel_names(k)

            def row(name, a):
                per_unit = alg.get(name, 2 * rbytes)
                step_ms = a["ms"] / args.steps
                units_step = a["units"] / args.steps
                ach = per_unit * units_step / (step_ms * 1e-3) / 1e9 if step_ms > 0 else 0.0
                pn = pmc_name.get(name, name)
                ent = next((tj[c] for c in (pn if isinstance(pn, tuple) else (pn,)) if c in tj), None)
                r = {"kernel": name, "ms_per_step": round(step_ms, 3), "launches_per_step": a["launches"] / args.steps,
                     "alg_bytes_per_unit": round(per_unit, 2), "units_per_step": units_step,
                     "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4),
                     "traffic_bytes_per_step": None, "measured_gb_per_s": None, "measured_hbm_frac": None, "bound": "hbm",
                     # the resource that actually binds the kernel and how much of it is used (frac above is the SURVEY 8(d)
                     # contract figure -- algorithmic bytes against the HBM roof -- whatever the kernel is bound by)
                     "bound_resource": None, "bound_frac": None}
                if ent and step_ms > 0:
                    tb = ent["bytes_per_instance"] * n_inst     # the PMC file is normalised per k-mer instance of the profiled run
                    pl = ent.get("dispatches", 0) / max(tj.get("_steps", 0), 1)
                    if tj.get("_steps") and pl > 0:             # ... and per launch, if the profiled size ran another number of passes
                        tb *= (a["launches"] / args.steps) / pl
                    r["traffic_bytes_per_step"] = round(tb, 0)
                    r["measured_gb_per_s"] = round(tb / (step_ms * 1e-3) / 1e9, 1)
                    r["measured_hbm_frac"] = round(tb / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                    # a kernel that moves < 0.2x its algorithmic bytes through HBM is not bounded by HBM: its work happens in
                    # LDS / VALU (the counting kernel keeps the records' k-mers on chip)
                    if tb < 0.2 * per_unit * units_step:
                        r["bound"] = "lds/valu"
                    elif r["measured_hbm_frac"] < 0.25:
                        r["bound"] = "valu/latency"
                if name in ("sk_scan", "sk_scan_long") and n_recs and step_ms > 0:
                    r["bound_resource"] = "global slot atomics (one returning atomicAdd per super-k-mer record; device rate %.1e/s, tools/micro/atomic_stride.hip)" % ATOMIC_PEAK_PER_S
                    r["atomics_per_step"] = n_recs
                    r["bound_frac"] = round(n_recs / (step_ms * 1e-3) / ATOMIC_PEAK_PER_S, 4)
                    r["bound"] = "atomics"
                elif ent and ent.get("valu_busy") is not None and r["bound"] != "hbm":
                    # counters of the same PMC passes as the traffic (hash-guarded by traffic_stale): VALUBusy and the share of CU
                    # cycles with the LDS busy; neither saturates in bin_count -- it is bound by the dependent LDS / barrier chain
                    # of a bin at two workgroups per CU -- so the larger of the two is what is reported
                    r["valu_busy"], r["lds_active"] = ent.get("valu_busy"), ent.get("lds_active")
                    r["lds_bank_conflict_share"], r["wave_wait_share"] = ent.get("lds_conflict"), ent.get("wave_wait_share")
                    r["bound_resource"] = "VALU issue (SQ_ACTIVE_INST_VALU*4/1024 SIMDs per cycle) next to a dependent LDS / barrier chain"
                    r["bound_frac"] = ent.get("valu_busy")
                elif r["measured_hbm_frac"] is not None:
                    r["bound_resource"] = "HBM bandwidth (FETCH_SIZE*2 + WRITE_SIZE per second / 8 TB/s)"
                    r["bound_frac"] = r["measured_hbm_frac"]
                return r

            rows = [row(nm, a) for nm, a in sorted(ktimes.items(), key=lambda kv: -kv[1]["ms"])]
            d = rows[0]
            a = dom[1]
            roof = {"bound": d["bound"], "kernel": d["kernel"], "achieved": d["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": d["frac"], "traffic": (round(d["traffic_bytes_per_step"] / d["launches_per_step"], 0)
                                                   if d["traffic_bytes_per_step"] is not None else None),
                    "measured_hbm_frac": d["measured_hbm_frac"],
                    "alg_bytes_per_unit": d["alg_bytes_per_unit"], "units_per_launch": a["units"] / a["launches"],
                    "avg_launch_ms": round(a["ms"] / a["launches"], 4), "launches_per_step": d["launches_per_step"],
                    "whole_path_alg_frac": round(value * b_alg / HBM_PEAK_GBS, 4),
                    "whole_path_alg_bytes_per_unit": round(b_alg, 2),
                    # what THIS design has to move per k-mer instance at the least: the packed reads in, every super-k-mer record
                    # written once and read once, the valid k-mers' table out (SURVEY 8(d)'s two-touch model prices 17-20 B records
                    # per instance; super-k-mers carry ~1.6 B) -- and with the three order-restoring passes it actually runs
                    "design_floor_bytes_per_unit": round(b_in + 2 * sk_b + u_over_n * (key + 3), 2),
                    "design_traffic_bytes_per_unit": round(b_in + 2 * sk_b + u_over_n * (16 + 3 * 2 * sort_rec + sort_rec + key + 3 + (4 + 4 * 1.5 if is_set else 0)), 2),
                    "whole_path_measured_bytes_per_unit": (round(sum(r["traffic_bytes_per_step"] or 0 for r in rows) / max(n_inst, 1), 2)
                                                           if any(r["traffic_bytes_per_step"] for r in rows) else None),
                    "whole_path_measured_hbm_frac": (round(sum(r["traffic_bytes_per_step"] or 0 for r in rows) / (ms_per_step * 1e-3) / 1e9
                                                           / HBM_PEAK_GBS, 4) if any(r["traffic_bytes_per_step"] for r in rows) else None),
                    "note": "achieved/frac: SURVEY 8(d) algorithmic bytes / HIP-event time (the contract figure); measured_*: HBM bytes "
                            "from the FETCH_SIZE/WRITE_SIZE counter passes (%s); bound = lds/valu when the measured traffic is "
                            "below 0.2x the algorithmic bytes" % (os.path.basename(tf[-1]) if tf else "no PMC file"),
                    "traffic_source": os.path.basename(tf[-1]) if tf else None, "traffic_stale": traffic_stale if tf else None,
                    "kernel_source_sha16": src_sha,
                    "kernels": rows,
                    "kernel_ms_per_step": {n: round(v["ms"] / args.steps, 3) for n, v in ktimes.items()},
                    "kernel_ms_min_med_max": {n: [min(v), sorted(v)[len(v) // 2], max(v)] for n, v in kseries.items() if n != "sk_records" and v},
                    "kernel_ms_by_step": {n: v for n, v in kseries.items() if n in ("sk_scan", "bin_count")}}
        other = None
        if world == 1 and not args.force_sharded and not args.no_other_shapes:
            # the shapes the crate is used with day to day, on the same reads (BASELINE configs 4 / 5 per-GPU k, Kmer32's neighbourhood,
            # the plain counter): two timed steps each after one warm-up, per-kernel ms included
            other = {}
            # (k47_set_5000_labels: 5000 distinct u32 labels spread over [0, 2^24), one per read at random -- the label-list route)
            gen = torch.Generator(device=dev); gen.manual_seed(7)
            lab5k = torch.randperm(1 << 24, device=dev, generator=gen)[:5000].to(torch.int32)[
                torch.randint(0, 5000, (reads_per_gpu,), device=dev, generator=gen)].contiguous()
            for nm, k2, set2 in (("k31_set", 31, 1), ("k63_set", 63, 1), ("k47_count", 47, 0), ("k24_set", 24, 1), ("k47_set_5000_labels", 47, 2)):
                if k2 == k and bool(set2) == is_set and set2 != 2:
                    continue
                fp2 = capi.FilterParams(k2, 0, 1 if set2 else 0, args.min_obs, 0, 4)
                ss2 = capi.SeqSet(words.data_ptr(), nw, start.data_ptr(), length.data_ptr(), None,
                                  lab5k.data_ptr() if set2 == 2 else (colour.data_ptr() if set2 else None), 4 if set2 == 2 else (1 if set2 else 0), reads_per_gpu)
                kt2, kt_units, ni2 = {}, {}, 0
                # (four untimed calls: the slab tournament of the shape -- the library tries four slab blocks and keeps the fastest)
                for rep in range(-3, 3):
                    if rep == 1:
                        ctx.enable_timing(True)
                        torch.cuda.synchronize()
                        o0 = time.perf_counter()
                    t = capi.KmerTable()
                    ctx.check(lib.dbg_filter_kmers_dev(ctx.h, C.byref(ss2), C.byref(fp2), C.byref(t)))
                    ni2, nv2 = t.n_kmer_instances, t.n
                    lib.dbg_free_table(ctx.h, C.byref(t))
                    if rep >= 1:
                        for kt in ctx.timings():
                            kt2[kt["name"]] = kt2.get(kt["name"], 0.0) + kt["ms"]
                            kt_units[kt["name"]] = kt_units.get(kt["name"], 0) + kt["units"]
                torch.cuda.synchronize()
                odt = (time.perf_counter() - o0) / 2
                ctx.enable_timing(False)
                nrec2 = kt_units.get("sk_records", 0) / 2
                kt2.pop("sk_records", None)
                other[nm] = {"k": k2, "summarizer": "CountFilterSet" if set2 else "CountFilter", "value": round(ni2 / odt / 1e9, 3), "unit": "Gkmer/s",
                             "ms_per_step": round(odt * 1e3, 3), "valid_kmers": int(nv2), "kernel_ms_per_step": {n_: round(v / 2, 3) for n_, v in kt2.items()}}
                other[nm]["superkmer_records_per_step"] = nrec2
                other[nm]["roofline"] = shape_kernel_rows(nm, k2, other[nm]["kernel_ms_per_step"], ni2, nrec2, kernel_source_sha16())
                if set2 == 2:
                    other[nm]["labels"] = "5000 distinct u32 labels in [0, 2^24), one per read at random"
            del lab5k
            ctx.trim()                                               # (one tournament winner per shape sits in the pool)
        cpu = None
        if not args.no_cpu_baseline and world == 1:                # timed on rank 0 at N = 1 only
            import oracle_lib as O
            n_s = 500000                                       # ~52M k-mer instances: ~8 s of single-thread CPU work (the rate does not depend on the sample size)
            hs = dbg.synth_reads_host(n_reads=n_s, read_len=L, genome_len=n_s * L // 30, error_rate=args.error_rate,
                                      stranded=False, n_colours=4)
            so = O.SeqSet(hs.words, hs.start, hs.length, None, hs.data if is_set else None, 1 if is_set else 0)
            sec, nv = O.time_filter_kmers(so, k, O.COUNT_FILTER_SET if is_set else O.COUNT_FILTER, args.min_obs, False)
            cpu = {"value": round(n_s * (L - k + 1) / sec / 1e9, 5), "unit": "Gkmer/s", "cores": 1, "kind": "port",
                   "sample": "first %d reads of an equally-parameterised stream (%.1fM k-mer instances), "
                             "oracle filter_kmers single thread, %.1f s" % (n_s, n_s * (L - k + 1) / 1e6, sec),
                   "host_cores_available": os.cpu_count()}
            # the same port on every host core, parallelised the way callers parallelise the crate (which has no
            # parallel path of its own): msp_sequence -> shards -> filter_kmers per shard (test.rs:418-504)
            cores = len(os.sched_getaffinity(0))
            try:                                               # container CPU quota (cgroup v2), if any
                q, per = open("/sys/fs/cgroup/cpu.max").read().split()
                if q != "max":
                    cores = max(1, min(cores, -(-int(q) // int(per))))
            except (OSError, ValueError):
                pass
            cpu["host_cores_granted"] = cores
            nt = 2 * cores                                     # two threads per granted core measured best on the GPU box
            if cores > 1:
                n_m = min(reads_per_gpu, max(n_s, 125000 * nt))     # a few seconds of wall time, a few GB of host memory
                pm = dbg.synth_params(n_reads=n_m, read_len=L, genome_len=n_m * L // 30, error_rate=args.error_rate,
                                      stranded=False, n_colours=4)
                nwm = lib.dbg_synth_words(C.byref(pm))
                wm = torch.empty(nwm, dtype=torch.int64, device=dev)
                sm = torch.empty(n_m, dtype=torch.int64, device=dev)
                lm = torch.empty(n_m, dtype=torch.int32, device=dev)
                cm = torch.empty(n_m, dtype=torch.uint8, device=dev)
                ctx.check(lib.dbg_synth_reads_dev(ctx.h, C.byref(pm), wm.data_ptr(), sm.data_ptr(), lm.data_ptr(), cm.data_ptr()))
                som = O.SeqSet(wm.cpu().numpy().view(np.uint64), sm.cpu().numpy().view(np.uint64),
                               lm.cpu().numpy().view(np.uint32), None,
                               cm.cpu().numpy().astype(np.uint32) if is_set else None, 1 if is_set else 0)
                del wm, sm, lm, cm
                msec, mnv = O.time_filter_kmers_sharded_mt(som, k, O.COUNT_FILTER_SET if is_set else O.COUNT_FILTER,
                                                           args.min_obs, False, nt, 16 * nt)
                cpu["all_cores"] = {"value": round(n_m * (L - k + 1) / msec / 1e9, 5), "unit": "Gkmer/s", "cores": cores,
                                    "threads": nt, "kind": "port", "valid_kmers": int(mnv),
                                    "sample": "%d reads of an equally-parameterised stream (%.0fM k-mer instances), oracle "
                                              "msp_sequence(p=8) -> %d shards -> filter_kmers per shard on %d threads, %.1f s"
                                              % (n_m, n_m * (L - k + 1) / 1e6, 16 * nt, nt, msec)}
        hostb = None
        if not args.no_host_boundary and world == 1:
            # The reference's own boundary (filter.rs:139-148): the caller's reads in (pageable) host memory, the table back in
            # host memory.  PCIe-inclusive, so it is never `value`; CountFilter is the north-star figure, CountFilterSet = the
            # bench workload.  First call of each kind warms the ctx's pinned result pool (hipHostMalloc is slow), second is timed.
            hw, hst, hl, hc = words.cpu().numpy(), start.cpu().numpy(), length.cpu().numpy(), colour.cpu().numpy()
            hostb = {}
            # (CountFilterSet_compact: dbg_filter_params.compact_sets -- the CSR crosses PCIe as u32 offsets + u8 labels)
            for kind, setk, compact in (("CountFilter", 0, 0), ("CountFilterSet", 1, 0), ("CountFilterSet_compact", 1, 3)):
                hss = capi.SeqSet(hw.ctypes.data, nw, hst.ctypes.data, hl.ctypes.data, None, hc.ctypes.data if setk else None, 1 if setk else 0,
                                  reads_per_gpu)
                hfp = capi.FilterParams(k, 0, setk, args.min_obs, 0, 4, compact)
                for rep in range(2):
                    ht = capi.KmerTable()
                    torch.cuda.synchronize()
                    h0 = time.perf_counter()
                    ctx.check(lib.dbg_filter_kmers(ctx.h, C.byref(hss), C.byref(hfp), C.byref(ht)))
                    hdt = time.perf_counter() - h0
                    b_in = hw.nbytes + hst.nbytes + hl.nbytes + (hc.nbytes if setk else 0)
                    b_out = ht.n * 17 + ((ht.n + 1) * (ht.set_off_width or 8) + ht.n_set_val * (ht.set_val_width or 4) if setk else ht.n * 2)
                    n_i = ht.n_kmer_instances
                    lib.dbg_free_table(ctx.h, C.byref(ht))
                hostb[kind] = {"value": round(n_i / hdt / 1e9, 3), "unit": "Gkmer/s", "seconds": round(hdt, 4),
                               "gb_host_to_device": round(b_in / 1e9, 2), "gb_device_to_host": round(b_out / 1e9, 2),
                               "pcie_gb_per_s": round((b_in + b_out) / hdt / 1e9, 1)}
            hostb["boundary"] = "dbg_filter_kmers: pageable host arrays in, host table out (pinned result pool of the ctx, warm)"
            # one pinned copy runs at 57 GB/s in either direction and the two directions do not add up (tools/micro/pcie_bw.hip,
            # profiles/r05_pcie_bw.txt): the bytes above over that rate, plus the kernels that cannot run under a copy (everything but
            # the scan), is what this boundary can reach on this link
            hostb["pcie_bound_note"] = "(GB in + GB out) / 57 GB/s + ~0.08 s of kernels that no copy can hide"
            del hw, hst, hl, hc
        comp = None
        if args.compress_reads and world == 1:
            # second half of the metric ("+ unitigs/s compressed"): CountFilter(2) table of a prefix of the same
            # stream -> host (the Rust caller holds a BoomHashMap2 in host memory) -> compress_kmers_with_hash.
            # Timed end to end at the host boundary: H2D of the index, device links + unitig construction, D2H.
            m = reads_per_gpu if args.compress_reads < 0 else min(args.compress_reads, reads_per_gpu)
            ss2 = capi.SeqSet(words.data_ptr(), nw, start.data_ptr(), length.data_ptr(), None, None, 0, m)
            fp2 = capi.FilterParams(k, 0, 0, 2, 0, 4)
            t2 = capi.KmerTable()
            ctx.check(lib.dbg_filter_kmers_dev(ctx.h, C.byref(ss2), C.byref(fp2), C.byref(t2)))
            # (a) index kept in HBM between filter_kmers and compress (dbg_compress_kmers_with_hash_dev); graph to the host
            gd = capi.Graph()
            for rep in range(2):                                   # first call warms the library's device-memory pool
                if rep:
                    lib.dbg_free_graph(ctx.h, C.byref(gd))
                torch.cuda.synchronize()
                c0 = time.perf_counter()
                ctx.check(lib.dbg_compress_kmers_with_hash_dev(ctx.h, k, 0, 0, t2.n, t2.key_hi, t2.key_lo, t2.exts, None, t2.count, C.byref(gd)))
                ddt = time.perf_counter() - c0
            dev_nodes = gd.n_nodes
            lib.dbg_free_graph(ctx.h, C.byref(gd))
            # (b) the reference's boundary: index in host memory
            h2 = capi.KmerTable()
            ctx.check(lib.dbg_table_to_host(ctx.h, C.byref(t2), C.byref(h2)))
            lib.dbg_free_table(ctx.h, C.byref(t2))
            nk = h2.n
            d32 = np.ctypeslib.as_array(C.cast(h2.count, C.POINTER(C.c_uint16)), shape=(max(nk, 1),))[:nk].astype(np.uint32)
            g = capi.Graph()
            c0 = time.perf_counter()
            ctx.check(lib.dbg_compress_kmers_with_hash(ctx.h, k, 0, 0, nk, h2.key_hi, h2.key_lo, h2.exts,
                                                       d32.ctypes.data_as(C.c_void_p), None, C.byref(g)))
            cdt = time.perf_counter() - c0
            comp = {"reads": m, "valid_kmers": nk, "unitigs": g.n_nodes, "seconds": round(cdt, 4),
                    "unitigs_per_s": round(g.n_nodes / cdt, 1), "kmers_per_s": round(nk / cdt, 1),
                    "spec": "SimpleCompress(saturating_add)", "seed_order": "ascending key (policy B)",
                    "boundary": "host arrays in, host BaseGraph out (PCIe included)",
                    "device_resident_index": {"seconds": round(ddt, 4), "unitigs": dev_nodes, "unitigs_per_s": round(dev_nodes / ddt, 1),
                                              "kmers_per_s": round(nk / ddt, 1), "boundary": "index in HBM, host BaseGraph out"}}
            lib.dbg_free_graph(ctx.h, C.byref(g))
            lib.dbg_free_table(ctx.h, C.byref(h2))
            # (c) the pipeline real callers run (filter.rs:233-306, test.rs:236-254): filter_kmers -> remove_censored_exts ->
            # compress_kmers_with_hash.  Extensions towards k-mers that did not pass the filter are dropped first, so the valid
            # k-mers join into few, long unitigs (10^5 of ~4 kb instead of 2*10^7 short ones): another regime for the chain walks.
            # Warm (the bench ctx) and FIRST CALL (a fresh ctx: nothing pooled, nothing pinned), the latter itemised from the
            # ctx's allocation account.
            t3 = capi.KmerTable()
            ctx.check(lib.dbg_filter_kmers_dev(ctx.h, C.byref(ss2), C.byref(fp2), C.byref(t3)))
            ctx.enable_timing(True)
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            ctx.check(lib.dbg_remove_censored_exts(ctx.h, k, 0, C.byref(t3), 0))
            torch.cuda.synchronize()
            cens_dt = time.perf_counter() - c0

            def censored_compress(cx):
                gg = capi.Graph()
                cx.enable_timing(True)
                torch.cuda.synchronize()
                q0 = time.perf_counter()
                cx.check(lib.dbg_compress_kmers_with_hash_dev(cx.h, k, 0, 0, t3.n, t3.key_hi, t3.key_lo, t3.exts, None, t3.count, C.byref(gg)))
                qdt = time.perf_counter() - q0
                kt = {x["name"]: round(x["ms"], 3) for x in cx.timings()}
                cx.enable_timing(False)
                nn = int(gg.n_nodes)
                ln = np.ctypeslib.as_array(C.cast(gg.length, C.POINTER(C.c_uint32)), shape=(max(nn, 1),))[:nn]
                info = {"seconds": round(qdt, 4), "unitigs": nn, "unitigs_per_s": round(nn / qdt, 1), "kmers_per_s": round(t3.n / qdt, 1),
                        "longest_unitig_bases": int(ln.max()) if nn else 0, "mean_unitig_bases": round(float(ln.mean()), 1) if nn else 0.0,
                        "chain_route": ("segments (pieces through the chain walk, joined by compress_graph's device route)" if "unitig_segments_joined" in kt else "chain walk (unitig_chain_scan)" if "unitig_chain_scan" in kt and "unitig_pointer_jump" not in kt and "unitig_walk_ends" not in kt
                                        else ("pointer jumping" if "unitig_pointer_jump" in kt else ("end walks" if "unitig_walk_ends" in kt else "host walk"))),
                        "kernel_ms": kt}
                lib.dbg_free_graph(cx.h, C.byref(gg))
                return info
            censored_compress(ctx)                                   # (pool warm-up of this shape)
            warm = censored_compress(ctx)
            ctx.trim()                                               # the bench ctx's pool holds most of the device by now; a second ctx needs room
            cold_ctx = dbg.Context(local_rank)
            cold = censored_compress(cold_ctx)
            cst = cold_ctx.stats()
            cold["alloc_account"] = {kk: (round(v, 4) if isinstance(v, float) else v) for kk, v in cst.items() if kk.startswith(("n_", "s_", "pooled"))}
            cold_ctx.close()
            comp["censored"] = {"pipeline": "dbg_filter_kmers_dev(CountFilter(2)) -> dbg_remove_censored_exts -> dbg_compress_kmers_with_hash_dev (index in HBM, host BaseGraph out)",
                                "remove_censored_exts_seconds": round(cens_dt, 4), "warm": warm, "first_call_fresh_ctx": cold}
            lib.dbg_free_table(ctx.h, C.byref(t3))
        out = {
            "metric": "Gkmer/s extracted+counted (k=%d, 150 bp synthetic reads)" % k, "value": round(value, 4),
            "unit": "Gkmer/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": ("%s%dx150bp synthetic reads per GPU, k=%d, non-stranded, %s(min=%d), 30x, e=%g"
                                    % ("BASELINE config %s strong-scaled: %d reads in all = " % (args.config[1], n_reads_total) if strong else "",
                                       reads_per_gpu, k, "CountFilterSet<u8>" if is_set else "CountFilter", args.min_obs, args.error_rate)),
                       "kmer_instances_per_step": n_inst_total, "valid_kmers_rank0": n_valid, "valid_kmers_all_ranks": n_valid_total,
                       "path": ("fast (minimizer scan -> super-k-mer slabs per bin -> per-bin LDS hash tables -> order-restoring hybrid sort)" if fast
                                else "generic (extract -> global LSD radix sort -> segmented reduce)"),
                       "superkmer_records_per_step": n_recs,
                       "multi_gpu": ("reads sharded by index; one all-to-all of super-k-mer bin slabs; each rank counts "
                                     "the bins it owns" if world > 1 else "n/a")},
            "table_digest": ("%016x" % digest) if args.digest else None,
            "ranks_seen": ranks_seen, "backend": args.backend if world > 1 else None,
            "exchange": ({"bytes_sent_per_step_all_ranks": xbytes_total // max(args.steps, 1),
                          "exposed_ms_per_step_max_rank": round(exposed_ms / max(args.steps, 1), 3),
                          "rounds": int(xstats.get("exchange_rounds", 0)) // max(args.steps, 1),
                          "sender_merge": bool(xstats.get("merge_dups", 0)),
                          "ownership": "record histogram (all-reduced), greedy contiguous cut" if xstats.get("balanced") else "equal bin ranges",
                          "entry_point": "dbg_shard_filter_kmers_dev (C ABI; rounds ordered by HIP events on a communication stream)",
                          "transport": xstats.get("transport"), "transport_fallback": xstats.get("transport_fallback"),
                          "setup_ms_per_step_rank0": round(xstats.get("setup_ms", 0.0) / max(args.steps, 1), 3)} if (world > 1 or args.force_exchange) else None),
            "balance": balance,
            "slab": slab,
            "roofline": roof, "other_shapes": other, "cpu_baseline": cpu, "cpu_baseline_all_cores": (cpu or {}).get("all_cores"),
            "host_boundary": hostb, "compress": comp,
        }
    D.close_transports()
    ctx.close()
    if world > 1 or args.force_exchange:
        import torch.distributed as dist
        dist.destroy_process_group()
    if out is not None:
        # after the process group is gone: RCCL writes a "Librccl path" line to stdout when it shuts down, and the JSON line
        # is meant to be the last thing rank 0 prints
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
