# Counter passes (tools/pmc.sh: separate --pmc runs, kernel trace only) of the secondary shapes the bench line reports under other_shapes,
# so that their rows carry measured HBM bytes / VALU busy / the binding resource like the main rows (round-5 review, item 8).
# usage: tools/r06_pmc_shapes.sh [round tag] [reads]        -> gpurun_out/<tag>_pmc_shape_<shape>.json (copy to profiles/)
export PYTHONUNBUFFERED=1 DBG_SLAB_TRIALS=1
R=${1:-r06}; NR=${2:-60000000}
Q="--reads $NR --steps 2 --warmup 0 --no-cpu-baseline --compress-reads 0 --no-other-shapes --no-host-boundary"
for sh in "k31_set 31 set" "k63_set 63 set" "k24_set 24 set" "k47_count 47 count"; do
  set -- $sh
  timeout 900 bash tools/pmc.sh ${R}_$1 --k $2 --summarizer $3 $Q > gpurun_out/pmc_${R}_$1.log 2>&1
  python tools/pmc_traffic.py gpurun_out/pmc_${R}_$1 gpurun_out/${R}_pmc_shape_$1.json $((NR*(151-$2)*2)) 2 > /dev/null 2>&1
done
# label lists: tools/bench_labels.py runs --iters calls over 5000 labels
PMC_SCRIPT=tools/bench_labels.py timeout 900 bash tools/pmc.sh ${R}_labels --reads $NR --iters 2 5000 > gpurun_out/pmc_${R}_labels.log 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_${R}_labels gpurun_out/${R}_pmc_shape_k47_set_5000_labels.json $((NR*104*2)) 2 > /dev/null 2>&1
ls -la gpurun_out/${R}_pmc_shape_*.json
