#!/bin/bash
# rocprofv3 kernel trace + statistics, then the counter passes, of CountFilterSet with a large label alphabet (label lists, fast_labellists.hpp):
# tools/prof_labels.sh <tag> [reads] [labels]   ->  gpurun_out/prof_<tag>/kernel_stats.csv, gpurun_out/pmc_<tag>/summary.txt
TAG=${1:-labels}; READS=${2:-100000000}; LABELS=${3:-5000}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/tools/bench_labels.py --reads $READS --iters 3 $LABELS > $OUT/bench.log 2>&1
grep "^labels" $OUT/bench.log | cut -c1-400
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -14 $OUT/kernel_stats.csv | cut -c1-200
cd $GRAFT_REPO_ROOT
PMC_SCRIPT=tools/bench_labels.py timeout 1200 bash tools/pmc.sh $TAG --reads 20000000 --iters 1 $LABELS > /dev/null 2>&1
grep -A14 "bin_labels" gpurun_out/pmc_$TAG/summary.txt | head -40
