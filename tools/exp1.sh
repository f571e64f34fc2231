#!/bin/bash
B="timeout 300 python bench.py --reads 20000000 --steps 2 --warmup 1 --no-cpu-baseline --compress-reads 0"
ex() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step']['bin_count'])" "$1"; }
for t in 9500 8000 7400 7000 6500 11000; do DBG_FAST_TARGET=$t $B 2>&1 | ex tgt$t; done
