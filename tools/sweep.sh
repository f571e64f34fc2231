#!/bin/bash
# usage: tools/sweep.sh "ENV1=.. ENV2=.." ... ; prints kernel ms per step at 100M reads
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*\|kernel_ms_per_step.*' | tr '\n' ' '
  echo
done
