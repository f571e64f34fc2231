# Round-5 measurement set (run through gpurun): default bench, kernel trace, host boundary, k = 31 spread over processes.
R=${1:-r05}
export PYTHONUNBUFFERED=1
timeout 900 python bench.py > gpurun_out/${R}_bench_default.json 2> gpurun_out/${R}_bench_default.err; tail -c 2500 gpurun_out/${R}_bench_default.json | head -c 2400; echo
for i in 1 2 3 4 5 6; do
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-boundary --compress-reads 0 --no-other-shapes --k ${K:-47} 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('process $i k=${K:-47}:', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
done
