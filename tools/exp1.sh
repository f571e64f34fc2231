#!/bin/bash
B="timeout 120 python bench.py --reads 10000000 --steps 1 --warmup 0 --no-cpu-baseline --compress-reads 0"
DBG_DEBUG=1 $B 2>&1 | grep "fastpath" | cut -c1-330
