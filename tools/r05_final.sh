# End-of-round measurement set of round 5 (run through gpurun; every command under its own timeout).
# usage: tools/r05_final.sh [tag]     results land in gpurun_out/<tag>_*; the ones to keep are copied to profiles/ by hand
R=${1:-r05}
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/${R}_gputest_tail.txt; cat gpurun_out/${R}_gputest_tail.txt
bash tools/pmc_round.sh $R > gpurun_out/${R}_pmc_round.log 2>&1; tail -3 gpurun_out/${R}_pmc_round.log
cp gpurun_out/${R}_pmc_traffic_60Mreads.json profiles/${R}_pmc_traffic_60Mreads.json      # (on the box: the default bench reads the newest traffic file under profiles/)
timeout 900 python bench.py > gpurun_out/${R}_bench_default.json 2> gpurun_out/${R}_bench_default.err; tail -c 1500 gpurun_out/${R}_bench_default.json | head -c 1400; echo
bash tools/k_sweep.sh $R > gpurun_out/${R}_k_sweep.log 2>&1; tail -13 gpurun_out/${R}_k_sweep.log | cut -c1-200
bash tools/other_configs.sh $R > gpurun_out/${R}_other_configs.log 2>&1; tail -7 gpurun_out/${R}_other_configs.log | cut -c1-260
rm -f gpurun_out/${R}_second_stage.txt
for a in "--ranks 8 --reads-per-rank 12500000 --k 47" "--ranks 8 --reads-per-rank 15625000 --k 63" "--ranks 8 --reads-per-rank 9375000 --k 51 --summarizer set"; do
  timeout 1200 python tools/rehearse_shard.py $a --out gpurun_out/${R}_second_stage.txt 2>&1 | grep -v amdgpu.ids | tail -3
done
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-boundary --compress-reads 0 --no-other-shapes --force-exchange --backend nccl > gpurun_out/${R}_bench_force_exchange.json 2> gpurun_out/${R}_bench_force_exchange.err; tail -c 700 gpurun_out/${R}_bench_force_exchange.json | head -c 600; echo
