# The default bench shape (10^8 x 150 bp, CountFilterSet(2)) over k: per-kernel ms and records per read.
# One JSON line per k into gpurun_out/<round>_k_sweep.json  (usage: tools/k_sweep.sh r05 [extra bench args])
R=${1:-r05}; shift
KS=${KS:-"16 18 20 22 24 28 31 32 40 47 56 64"}
OUT=gpurun_out/${R}_k_sweep.json
: > $OUT
for k in $KS; do
  timeout 300 python bench.py --steps 3 --warmup 4 --no-cpu-baseline --no-host-boundary --compress-reads 0 --no-other-shapes --k $k "$@" 2>&1 | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
n = d['config']['kmer_instances_per_step']
reads = 100000000
print(json.dumps({'k': $k, 'value': d['value'], 'ms_per_step': d['ms_per_step'],
  'records_per_read': round(d['config']['superkmer_records_per_step'] / reads, 3),
  'valid_kmers': d['config']['valid_kmers_rank0'], 'kmer_instances': n,
  'kernel_ms_per_step': d['roofline']['kernel_ms_per_step']}))" >> $OUT
done
cat $OUT
