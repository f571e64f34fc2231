#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench (100M reads), summary to gpurun_out/prof_<tag>
TAG=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 4 --no-cpu-baseline --compress-reads 0 --no-other-shapes "$@" > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-300
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -30 $OUT/kernel_stats.csv | cut -c1-160
