# Soak of the round-6 tree: long differential fuzz hunts against the oracle (single call, rank-spanning calls with full-width labels in the draw)
# and mid-size whole-table comparisons.  usage: tools/r06_soak.sh [tag]
R=${1:-r06}
export PYTHONUNBUFFERED=1
DBG_FUZZ_SEEDS=12000 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -x -q -n 4 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3 > gpurun_out/${R}_fuzz_hunt.txt; cat gpurun_out/${R}_fuzz_hunt.txt
DBG_SHARD_FUZZ_SEEDS=3000 timeout 2400 python -m pytest tests/test_gpu_shard_fuzz.py -x -q -n 4 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3 > gpurun_out/${R}_shard_fuzz_hunt.txt; cat gpurun_out/${R}_shard_fuzz_hunt.txt
timeout 1500 python tools/hunt_midsize.py 24 606 2>&1 | tail -3 > gpurun_out/${R}_hunt_midsize.txt; cat gpurun_out/${R}_hunt_midsize.txt
