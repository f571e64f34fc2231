#!/bin/bash
# PMC collection (separate passes; never combined with trace domains other than kernel-trace).
# usage: tools/pmc.sh <tag> <bench args...>
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
run() { # name counters...
  n=$1; shift
  if [ -n "$PMC_SCRIPT" ]; then   # another driver than bench.py (tools/bench_msp.py, tools/bench_compress.py)
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- python $GRAFT_REPO_ROOT/$PMC_SCRIPT "${BENCH_ARGS[@]}" > $OUT/$n.log 2>&1 || tail -5 $OUT/$n.log
  else
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- python $GRAFT_REPO_ROOT/bench.py "${BENCH_ARGS[@]}" > $OUT/$n.log 2>&1 || tail -5 $OUT/$n.log
  fi
}
BENCH_ARGS=("$@")
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY
run sq2 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
