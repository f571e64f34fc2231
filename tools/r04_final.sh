#!/bin/bash
# Round-4 end-of-round measurements on the GPU box (through gpurun; every command under its own timeout).
R=r04
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout 400 bash tools/prof.sh $R --no-host-boundary > gpurun_out/prof_$R.log 2>&1; tail -3 gpurun_out/prof_$R.log
timeout 900 bash tools/pmc.sh $R --reads 60000000 --steps 2 --warmup 0 --no-cpu-baseline --compress-reads 0 --no-other-shapes --no-host-boundary > gpurun_out/pmc_$R.log 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_$R gpurun_out/${R}_pmc_traffic_60Mreads.json $((60000000*104*2)) 2 > gpurun_out/pmc_${R}_traffic.log 2>&1; tail -3 gpurun_out/pmc_${R}_traffic.log
# the default bench AFTER the counter passes: it reads the traffic file of this very tree (profiles/ on the box; copy it home afterwards)
cp gpurun_out/${R}_pmc_traffic_60Mreads.json profiles/
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${R}_bench_default.json 2> gpurun_out/${R}_bench_default.err; tail -c 300 gpurun_out/${R}_bench_default.json; echo
PMC_SCRIPT=tools/bench_compress.py timeout 900 bash tools/pmc.sh ${R}_compress 10000000 none > gpurun_out/pmc_${R}_compress.log 2>&1; tail -2 gpurun_out/pmc_${R}_compress.log
timeout 900 bash tools/other_configs.sh $R > gpurun_out/${R}_other_configs.log 2>&1; tail -2 gpurun_out/${R}_other_configs.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${R}_compress -- python $GRAFT_REPO_ROOT/tools/bench_compress.py 100000000 none > $GRAFT_REPO_ROOT/gpurun_out/prof_${R}_compress.log 2>&1
cd $GRAFT_REPO_ROOT; grep "dev-index" gpurun_out/prof_${R}_compress.log | tail -1
# the collective route at full size through the C entry point (one-rank RCCL communicator) and the sharded pipeline's overhead
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-boundary --compress-reads 0 --no-other-shapes --force-exchange --backend nccl > gpurun_out/${R}_bench_force_exchange.json 2> gpurun_out/${R}_bench_force_exchange.err; tail -c 900 gpurun_out/${R}_bench_force_exchange.json | head -c 600; echo
# 8 ranks on the one GPU: ownership on the low-complexity stream
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29811 tools/check_balance.py --one-device --backend gloo --reads 250000 2>/dev/null | grep '^{' | tail -1 > gpurun_out/${R}_balance_8ranks.json; cut -c1-400 gpurun_out/${R}_balance_8ranks.json
# dense path: LDS / partitioned forms against the device-atomic form
timeout 900 bash tools/dense_ab.sh 8 9 10 11 12 13 14 15 > gpurun_out/${R}_generic_dense.txt 2>&1; grep "=1" gpurun_out/${R}_generic_dense.txt | cut -c1-60
