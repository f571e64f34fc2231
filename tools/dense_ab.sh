#!/bin/bash
# dense path A/B at 2*10^7 reads: partitioned LDS counting (k >= 9, DBG_DENSE_PART) / key-range passes (k = 8, DBG_DENSE_RANGES) against the
# device-atomic form.  usage: tools/dense_ab.sh [k ...]
KS=${@:-8 9 11 13 15}
for k in $KS; do for sm in count set; do for m in 1 0; do
EXTRA="--summarizer $sm"
DBG_DENSE_PART=$m DBG_DENSE_RANGES=$m DBG_PATH=dense timeout 300 python bench.py --k $k --reads 20000000 --no-cpu-baseline --no-host-boundary --compress-reads 0 --no-other-shapes $EXTRA 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('k=$k $sm lds/partitioned=$m', d['value'], 'Gkmer/s', d['ms_per_step'], 'ms', d['roofline']['kernel_ms_per_step'], 'valid', d['config']['valid_kmers_rank0'])"
done; done; done
