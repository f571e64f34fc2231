#!/bin/bash
# internal minimizer length p (DBG_FAST_P) at small k under resident workgroups: fewer, longer super-k-mers against fewer distinct minimizers
mkdir -p gpurun_out
run() { python bench.py --k $1 --steps 3 --warmup 4 --no-cpu-baseline --compress-reads 0 --no-other-shapes --no-host-boundary 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('k=$1 p=$2', j['value'], j['ms_per_step'], [(k['kernel'], k['ms_per_step']) for k in j['roofline']['kernels']][:3])
"; }
{
for kp in "16 13" "16 12" "16 11" "16 10" "20 13" "20 12" "20 11" "24 13" "24 12" "24 11" "28 14" "28 13" "28 12" "31 15" "31 14" "31 13" "47 15" "47 14" "47 13"; do
  set -- $kp; DBG_FAST_P=$2 run $1 $2
done
} > gpurun_out/p_sweep.txt 2>&1
cat gpurun_out/p_sweep.txt
