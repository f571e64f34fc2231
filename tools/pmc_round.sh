# One round's counter passes (run on the GPU box through gpurun; every command under its own timeout):
#   fast path at 6*10^7 reads (bench.py), the MSP scanner (tools/bench_msp.py), compress at config 3 / 10 (tools/bench_compress.py)
# and the kernel trace of the default bench.  Summaries land in gpurun_out/; the ones to keep are copied to profiles/ by hand.
export PYTHONUNBUFFERED=1 DBG_SLAB_TRIALS=1      # (counter passes run two steps: no slab tournament inside them)
R=${1:-r02}
NR=${2:-60000000}      # reads of the fast-path counter passes: large enough that the sort takes the same form as at 10^8
timeout 900 bash tools/pmc.sh $R --reads $NR --steps 2 --warmup 0 --no-cpu-baseline --compress-reads 0 --no-other-shapes --no-host-boundary > gpurun_out/pmc_$R.log 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_$R gpurun_out/${R}_pmc_traffic_$((NR/1000000))Mreads.json $((NR*104*2)) 2 > gpurun_out/pmc_${R}_traffic.log 2>&1
PMC_SCRIPT=tools/bench_msp.py timeout 600 bash tools/pmc.sh ${R}_msp 10000000 3 > gpurun_out/pmc_${R}_msp.log 2>&1
PMC_SCRIPT=tools/bench_compress.py timeout 900 bash tools/pmc.sh ${R}_compress 10000000 device > gpurun_out/pmc_${R}_compress.log 2>&1
env -u DBG_SLAB_TRIALS timeout 400 bash tools/prof.sh $R --no-host-boundary > gpurun_out/prof_$R.log 2>&1
for f in gpurun_out/pmc_$R.log gpurun_out/pmc_${R}_msp.log gpurun_out/pmc_${R}_compress.log gpurun_out/prof_$R.log; do tail -n 3 $f; done
