#!/bin/bash
# rocprofv3 kernel trace + stats of the censored config-3 pipeline (tools/bench_censored.py: filter -> remove_censored_exts -> compress x 3)
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_censored_$TAG
mkdir -p $OUT
timeout 800 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/tools/bench_censored.py > $OUT/bench.log 2>&1
grep "compress call" $OUT/bench.log | cut -c1-200
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -24 $OUT/kernel_stats.csv | cut -c1-200
