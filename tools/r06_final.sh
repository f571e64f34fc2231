# End-of-round measurement set of round 6 (run through gpurun; every command under its own timeout).
# usage: tools/r06_final.sh [tag]     results land in gpurun_out/<tag>_*; the ones to keep are copied to profiles/ by hand
R=${1:-r06}
export PYTHONUNBUFFERED=1
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6 > gpurun_out/${R}_gputest_tail.txt; cat gpurun_out/${R}_gputest_tail.txt
bash tools/pmc_round.sh $R > gpurun_out/${R}_pmc_round.log 2>&1; tail -3 gpurun_out/${R}_pmc_round.log
cp gpurun_out/${R}_pmc_traffic_60Mreads.json profiles/${R}_pmc_traffic_60Mreads.json      # (on the box: the default bench reads the newest traffic file under profiles/)
bash tools/r06_pmc_shapes.sh $R > gpurun_out/${R}_pmc_shapes.log 2>&1; tail -6 gpurun_out/${R}_pmc_shapes.log
cp gpurun_out/${R}_pmc_shape_*.json profiles/ 2>/dev/null
# the driver's command, three fresh processes
for i in 1 2 3; do
  timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${R}_bench_default_$i.json 2> gpurun_out/${R}_bench_default_$i.err
  python3 -c "
import json; j=json.loads(open('gpurun_out/${R}_bench_default_$i.json').read().strip().splitlines()[-1]); r=j['roofline']
print('bench $i', j['value'], j['ms_per_step'], r['kernel_ms_per_step'], 'scan', r['kernel_ms_min_med_max']['sk_scan'], j['slab']['tournament_scan_ms'], 'censored', j['compress']['censored']['warm']['seconds'], j['compress']['censored']['first_call_fresh_ctx']['seconds'])"
done
timeout 600 python3 bench.py --gpus 1 --reads 100000000 --steps 1 --warmup 0 --digest --no-cpu-baseline --no-host-boundary --compress-reads 0 --no-other-shapes 2>/dev/null | grep '^{' | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); json.dump({'c2_1e8_reads_k47_set': [d['config']['valid_kmers_all_ranks'], d['table_digest']]}, open('gpurun_out/${R}_digests.json','w')); print(open('gpurun_out/${R}_digests.json').read())"
bash tools/k_sweep.sh $R > gpurun_out/${R}_k_sweep.log 2>&1; tail -13 gpurun_out/${R}_k_sweep.log | cut -c1-200
bash tools/other_configs.sh $R > gpurun_out/${R}_other_configs.log 2>&1; tail -7 gpurun_out/${R}_other_configs.log | cut -c1-260
rm -f gpurun_out/${R}_second_stage.txt
for a in "--ranks 8 --reads-per-rank 12500000 --k 47" "--ranks 8 --reads-per-rank 15625000 --k 63" "--ranks 8 --reads-per-rank 9375000 --k 51 --summarizer set"; do
  timeout 1200 python tools/rehearse_shard.py $a --out gpurun_out/${R}_second_stage.txt 2>&1 | grep -v amdgpu.ids | tail -2
done
timeout 900 python bench.py --steps 5 --warmup 5 --no-cpu-baseline --no-host-boundary --compress-reads 0 --no-other-shapes --force-exchange --backend nccl > gpurun_out/${R}_bench_force_exchange.json 2> gpurun_out/${R}_bench_force_exchange.err; tail -c 700 gpurun_out/${R}_bench_force_exchange.json | head -c 600; echo
timeout 1500 bash tools/first_contact_8gpu.sh --dry-run --out gpurun_out/${R}_first_contact_dry > gpurun_out/${R}_first_contact_dry.log 2>&1; tail -3 gpurun_out/${R}_first_contact_dry.log
timeout 600 python tools/bench_labels.py 250 5000 2>&1 | grep "^labels" > gpurun_out/${R}_label_lists.txt; cat gpurun_out/${R}_label_lists.txt | cut -c1-300
