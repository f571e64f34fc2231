#!/bin/bash
# 256-thread workgroups with 1024-slot tables (four per CU instead of two) under resident workgroups
mkdir -p gpurun_out
run() { python bench.py --steps 6 --warmup 4 --no-cpu-baseline --compress-reads 0 --no-other-shapes --no-host-boundary 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$1', j['value'], j['ms_per_step'], [(k['kernel'], k['ms_per_step']) for k in j['roofline']['kernels']][:4])
"; }
{
run default
for t in 3000 3500 4000 4500 5000 6000; do DBG_FAST_NT=256 DBG_FAST_TABLE=1024 DBG_FAST_TARGET=$t run nt256_t1024_target$t; done
} > gpurun_out/nt256_sweep.txt 2>&1
cat gpurun_out/nt256_sweep.txt
