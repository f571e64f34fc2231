#!/bin/bash
# k-mer instances per bin (DBG_FAST_TARGET) under resident workgroups; default at k = 47: 8085
mkdir -p gpurun_out
run() { python bench.py --steps 6 --warmup 4 --no-cpu-baseline --compress-reads 0 --no-other-shapes --no-host-boundary 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$1', j['value'], j['ms_per_step'], [(k['kernel'], k['ms_per_step']) for k in j['roofline']['kernels']][:4])
"; }
{
for t in 5000 6500 8085 10000 12000 16000; do DBG_FAST_TARGET=$t run target$t; done
} > gpurun_out/target_sweep.txt 2>&1
cat gpurun_out/target_sweep.txt
