#!/bin/bash
# Instruction counts and duration of the counting kernel per phase: runs the DBG_ABL_COUNT builds (tools/build_variant.py ablN
# fastpath.hip -DDBG_ABL_COUNT=N, N = 2..5, and the regular build) under one counter pass each.  usage: tools/abl_count.sh [reads]
NR=${1:-10000000}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/abl_count
rm -rf $OUT; mkdir -p $OUT
for v in 2 3 4 5 full; do
  L=$GRAFT_REPO_ROOT/rust-debruijn_amd/_exp/libabl$v.so
  [ $v = full ] && L=$GRAFT_REPO_ROOT/rust-debruijn_amd/libdbg_mi355x.so
  DBG_LIB=$L timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/$v -- \
    python $GRAFT_REPO_ROOT/bench.py --reads $NR --steps 1 --warmup 0 --no-cpu-baseline --compress-reads 0 --no-other-shapes --no-host-boundary > $OUT/$v.log 2>&1 || tail -n 3 $OUT/$v.log
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for v in ["2", "3", "4", "5", "full"]:
    agg = collections.defaultdict(float)
    for f in glob.glob("gpurun_out/abl_count/%s/*/*counter_collection.csv" % v):
        for r in csv.DictReader(open(f)):
            if "bin_count_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
    dur = 0
    for f in glob.glob("gpurun_out/abl_count/%s/*/*kernel_trace.csv" % v):
        for r in csv.DictReader(open(f)):
            if "bin_count_kernel" in r["Kernel_Name"]:
                dur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    print(v, "ms=%.2f" % dur, " ".join("%s=%.4g" % kv for kv in sorted(agg.items())))
PY
