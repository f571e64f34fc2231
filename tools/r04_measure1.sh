#!/bin/bash
# Round-4 measurement batch 1 (through gpurun): (a) per-kernel times of the device-resident compress at config-3 size, to bisect
# the r01 -> r03 regression against profiles/r01_compress_config3_kernel_stats.csv; (b) LDS bank conflicts of the counting kernel
# PER PHASE (the DBG_ABL_COUNT builds stop after phase n of a bin: conflicts of phase n = counters(n) - counters(n - 1));
# (c) ownership on a low-complexity stream with 8 ranks on the one GPU; (d) the new multirank / entry-point tests.
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04m1
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# (a)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/compress -- python $R/tools/bench_compress.py 100000000 none > $O/compress.log 2>&1
tail -4 $O/compress.log
# (b)
NR=10000000
for v in 2 3 4 5 full; do
  L=$R/rust-debruijn_amd/_exp/libabl$v.so
  [ $v = full ] && L=$R/rust-debruijn_amd/libdbg_mi355x.so
  DBG_LIB=$L timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $O/lds_$v -- \
    python $R/bench.py --reads $NR --steps 1 --warmup 0 --no-cpu-baseline --compress-reads 0 --no-other-shapes --no-host-boundary > $O/lds_$v.log 2>&1 || tail -n 3 $O/lds_$v.log
done
cd $R
python - <<'PY'
import csv, glob, collections
rows = {}
for v in ["2", "3", "4", "5", "full"]:
    agg = collections.defaultdict(float)
    for f in glob.glob("gpurun_out/r04m1/lds_%s/*/*counter_collection.csv" % v):
        for r in csv.DictReader(open(f)):
            if "bin_count_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
    dur = 0
    for f in glob.glob("gpurun_out/r04m1/lds_%s/*/*kernel_trace.csv" % v):
        for r in csv.DictReader(open(f)):
            if "bin_count_kernel" in r["Kernel_Name"]:
                dur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    rows[v] = (dur, dict(agg))
    print("upto", v, "ms=%.2f" % dur, " ".join("%s=%.4g" % kv for kv in sorted(agg.items())))
names = {"2": "bounds+clear", "3": "staging", "4": "chunk map", "5": "chunk set-up", "full": "roll+probe+emit"}
prev = (0.0, collections.defaultdict(float))
print("\nper phase (difference of consecutive builds):")
for v in ["2", "3", "4", "5", "full"]:
    d, a = rows[v]
    conf = a.get("SQ_LDS_BANK_CONFLICT", 0) - prev[1].get("SQ_LDS_BANK_CONFLICT", 0)
    act = a.get("SQ_LDS_IDX_ACTIVE", 0) - prev[1].get("SQ_LDS_IDX_ACTIVE", 0)
    ins = a.get("SQ_INSTS_LDS", 0) - prev[1].get("SQ_INSTS_LDS", 0)
    print("%-16s ms=%6.2f  lds_insts=%.3g  idx_active=%.3g  bank_conflict=%.3g  conflict/active=%.1f%%" % (names[v], d - prev[0], ins, act, conf, 100 * conf / act if act else 0))
    prev = (d, a)
PY
# (c)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29811 tools/check_balance.py --one-device --backend gloo --reads 250000 > $O/balance8.log 2>&1
grep '^{' $O/balance8.log | tail -1 | cut -c1-1500
# (d)
timeout 1500 python -m pytest tests/test_gpu_shard_entry.py tests/test_gpu_multirank.py -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
