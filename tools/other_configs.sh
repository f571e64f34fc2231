# The other BASELINE shapes on one GPU (per-GPU shares of configs 4 and 5, CountFilter, error-free reads, larger input):
# one JSON line each into gpurun_out/<round>_other_configs.json
R=${1:-r02}
OUT=gpurun_out/${R}_other_configs.json
: > $OUT
run() { timeout 300 python bench.py --steps 3 --warmup 4 --no-cpu-baseline --no-host-boundary --compress-reads 0 --no-other-shapes "$@" 2>&1 | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'args': '$*', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'workload': d['config']['workload'], 'kernel_ms_per_step': d['roofline']['kernel_ms_per_step']}))" >> $OUT; }
run --k 63
run --k 51
run --k 31
run --summarizer count
run --error-rate 0
run --reads 250000000
cat $OUT
