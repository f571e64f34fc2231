set -x
export PYTHONUNBUFFERED=1
timeout 900 bash tools/pmc.sh r02 --reads 10000000 --steps 2 --warmup 0 --no-cpu-baseline --compress-reads 0 --no-host-boundary > gpurun_out/pmc_r02.log 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_r02 gpurun_out/r02_pmc_traffic_10Mreads.json 2080000000 > gpurun_out/pmc_r02_traffic.log 2>&1
timeout 400 bash tools/prof.sh r02 > gpurun_out/prof_r02.log 2>&1
tail -5 gpurun_out/pmc_r02.log; tail -30 gpurun_out/pmc_r02_traffic.log; tail -20 gpurun_out/prof_r02.log
