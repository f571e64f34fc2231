#!/bin/bash
# Instruction counts and duration of the wave-per-bin counting kernel per phase: runs the WV_ABL builds (tools/build_variant.py wablN
# fastpath.hip -DWV_ABL=N, N = 1..5, and the regular build) under one counter pass each.  usage: tools/abl_wave.sh [reads]
NR=${1:-20000000}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/abl_wave
rm -rf $OUT; mkdir -p $OUT
for v in 1 2 3 4 5 full; do
  L=$GRAFT_REPO_ROOT/rust-debruijn_amd/_exp/libwabl$v.so
  [ $v = full ] && L=$GRAFT_REPO_ROOT/rust-debruijn_amd/libdbg_mi355x.so
  [ -f $L ] || continue
  DBG_LIB=$L timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/$v -- \
    python $GRAFT_REPO_ROOT/bench.py --reads $NR --steps 1 --warmup 0 --no-cpu-baseline --compress-reads 0 --no-host-boundary > $OUT/$v.log 2>&1 || tail -n 3 $OUT/$v.log
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for v in ["1", "2", "3", "4", "5", "full"]:
    agg = collections.defaultdict(float)
    for f in glob.glob("gpurun_out/abl_wave/%s/*/*counter_collection.csv" % v):
        for r in csv.DictReader(open(f)):
            if "bin_count_wave" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
    dur = 0
    for f in glob.glob("gpurun_out/abl_wave/%s/*/*kernel_trace.csv" % v):
        for r in csv.DictReader(open(f)):
            if "bin_count_wave" in r["Kernel_Name"]:
                dur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if agg: print(v, "ms=%.2f" % dur, " ".join("%s=%.4g" % kv for kv in sorted(agg.items())))
PY
