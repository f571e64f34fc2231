#!/bin/bash
# A/B of experimental builds on the default bench: tools/ab_libs.sh name1 name2 ...  (rust-debruijn_amd/_exp/lib<name>.so; "main" = the regular build)
for n in "$@"; do
  L=$PWD/rust-debruijn_amd/_exp/lib$n.so; [ $n = main ] && L=$PWD/rust-debruijn_amd/libdbg_mi355x.so
  DBG_LIB=$L python bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --compress-reads 0 --no-other-shapes --no-host-boundary 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$n', j['value'], j['ms_per_step'], [(k['kernel'], k['ms_per_step']) for k in j['roofline']['kernels']][:3])
"
done
