#!/bin/bash
# quick check of a fast-path change: the default bench (8 steps) twice + the fast-path / filter / fuzz tests
mkdir -p gpurun_out
run() { python bench.py --steps 8 --warmup 4 --no-cpu-baseline --compress-reads 0 --no-other-shapes --no-host-boundary 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$1', j['value'], j['ms_per_step'], [(k['kernel'], k['ms_per_step']) for k in j['roofline']['kernels']][:4])
"; }
{
run a; run b
timeout 1500 python -m pytest tests/test_gpu_fastpath.py tests/test_gpu_filter.py tests/test_gpu_fuzz.py tests/test_gpu_sharded.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
} > gpurun_out/quick.txt 2>&1
cat gpurun_out/quick.txt
