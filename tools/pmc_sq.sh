#!/bin/bash
# SQ counter passes only (two rocprofv3 --pmc runs), summary per kernel.  usage: tools/pmc_sq.sh <tag> <bench args...>
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq1 -- python $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/sq1.log 2>&1 || tail -5 $OUT/sq1.log
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/sq2 -- python $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/sq2.log 2>&1 || tail -5 $OUT/sq2.log
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -- python $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/grbm.log 2>&1 || tail -5 $OUT/grbm.log
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT > $OUT/summary.txt
head -60 $OUT/summary.txt
