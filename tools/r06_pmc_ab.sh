# Executed instruction counters of bin_count_kernel under experimental builds (rust-debruijn_amd/_exp/lib<name>.so, tools/build_variant.py):
#   LIBS="main hfold" bash tools/r06_pmc_ab.sh     -> SQ_INSTS_VALU, SQ_INSTS_LDS, SQ_ACTIVE_INST_VALU, wave / busy / wait cycles per build (2 steps at 2e7 reads)
cd /tmp && export TMPDIR=/tmp
export DBG_SLAB_TRIALS=1
for lib in ${LIBS:-main}; do
  L=$GRAFT_REPO_ROOT/rust-debruijn_amd/_exp/lib$lib.so; [ $lib = main ] && L=$GRAFT_REPO_ROOT/rust-debruijn_amd/libdbg_mi355x.so
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_ab_$lib; rm -rf $OUT; mkdir -p $OUT
  DBG_LIB=$L timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/bench.py --reads 20000000 --steps 2 --warmup 0 --no-cpu-baseline --compress-reads 0 --no-other-shapes --no-host-boundary > $OUT/log.txt 2>&1
  python3 - $OUT $lib <<'PY'
import csv, glob, sys, collections
root, lib = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(float)
for f in glob.glob(root + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "bin_count_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
print(lib, {k: "%.4g" % v for k, v in sorted(agg.items())})
PY
done
