#!/bin/bash
B="timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --compress-reads 0"
ex() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])" "$1"; }
timeout 400 python -m pytest tests/test_gpu_fastpath.py tests/test_gpu_filter.py tests/test_gpu_sharded.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -2
$B 2>&1 | ex csr
