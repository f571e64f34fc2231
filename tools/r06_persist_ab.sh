#!/bin/bash
# Resident workgroups of the counting kernels (bin_count, bin_labels) against one workgroup per bin (DBG_FAST_PERSIST=0)
mkdir -p gpurun_out
run() { python bench.py --steps 8 --warmup 4 --no-cpu-baseline --compress-reads 0 --no-other-shapes --no-host-boundary 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$1', j['value'], j['ms_per_step'], [(k['kernel'], k['ms_per_step']) for k in j['roofline']['kernels']][:3])
"; }
{
DBG_FAST_PERSIST=0 run per-bin
run resident
DBG_FAST_PERSIST=0 run per-bin
run resident
echo "--- label lists (tools/bench_labels.py)"
DBG_FAST_PERSIST=0 python tools/bench_labels.py 100 250 5000 2>&1 | tail -4
python tools/bench_labels.py 100 250 5000 2>&1 | tail -4
echo "--- tests"
timeout 1500 python -m pytest tests/test_gpu_fastpath.py tests/test_gpu_filter.py tests/test_gpu_labellists.py tests/test_gpu_fullwidth_labels.py tests/test_gpu_fuzz.py tests/test_gpu_shard_entry.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -4
} > gpurun_out/persist_ab.txt 2>&1
tail -40 gpurun_out/persist_ab.txt
