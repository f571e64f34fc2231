#!/bin/bash
# Segment route of the unitig construction (long chains in two levels): interval sweep on the censored config-3 pipeline + tests
mkdir -p gpurun_out
{
for e in 128 128; do echo "DBG_SEGMENTS=$e"; DBG_SEGMENTS=$e timeout 500 python tools/bench_censored.py 2>&1 | grep "compress call 2" | cut -c1-700; done
echo "--- tests"
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_compress.py tests/test_gpu_graph.py tests/test_gpu_sharded.py tests/test_gpu_shard_entry.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
} > gpurun_out/segments.txt 2>&1
cat gpurun_out/segments.txt
