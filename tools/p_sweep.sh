# Internal minimizer length p of the fast path against k (DBG_FAST_P): default bench shape, one JSON line per (k, p)
# usage: tools/p_sweep.sh r05 "16 18 20 22" "11 12 13 14 15"
R=${1:-r05}; KS=${2:-"16 18 20 22"}; PS=${3:-"11 12 13 14 15"}
OUT=gpurun_out/${R}_p_sweep.json
: > $OUT
for k in $KS; do for p in $PS; do
  if [ $p -gt $((k-3)) ]; then continue; fi
  DBG_FAST_P=$p timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-boundary --compress-reads 0 --no-other-shapes --k $k ${READS:+--reads $READS} 2>&1 | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'k': $k, 'p': $p, 'value': d['value'], 'ms_per_step': d['ms_per_step'],
  'records_per_step': d['config']['superkmer_records_per_step'], 'kernel_ms_per_step': d['roofline']['kernel_ms_per_step']}))" >> $OUT
done; done
cat $OUT
