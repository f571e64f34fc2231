#!/bin/bash
# End-of-round measurements on the GPU box (through gpurun; every command under its own timeout).  usage: tools/round_measure.sh r03
R=${1:-r03}
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/${R}_gputest.log 2>&1; tail -3 gpurun_out/${R}_gputest.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${R}_bench_default.json 2> gpurun_out/${R}_bench_default.err; tail -c 600 gpurun_out/${R}_bench_default.json
timeout 400 bash tools/prof.sh $R --no-host-boundary > gpurun_out/prof_$R.log 2>&1; tail -3 gpurun_out/prof_$R.log
timeout 900 bash tools/other_configs.sh $R > gpurun_out/${R}_other_configs.log 2>&1; tail -2 gpurun_out/${R}_other_configs.log
timeout 600 python tools/bench_colours.py 4 24 40 64 100 250 > gpurun_out/${R}_colours.txt 2>&1; tail -4 gpurun_out/${R}_colours.txt
: > gpurun_out/${R}_generic_dense.txt
for a in "--k 47 --reads 10000000:generic" "--k 15 --reads 20000000:dense" "--k 11 --reads 20000000:dense" "--k 8 --reads 20000000:dense" "--k 15 --reads 20000000:generic"; do
  args=${a%%:*}; path=${a##*:}
  DBG_PATH=$path timeout 300 python bench.py $args --no-cpu-baseline --no-host-boundary --compress-reads 0 --no-other-shapes 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'args': '$args', 'path': '$path', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'valid': d['config']['valid_kmers_rank0'], 'kernel_ms_per_step': d['roofline']['kernel_ms_per_step']}))" >> gpurun_out/${R}_generic_dense.txt
done
cat gpurun_out/${R}_generic_dense.txt
