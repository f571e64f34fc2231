export PYTHONUNBUFFERED=1
DBG_FUZZ_SEEDS=40000 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -x -q -n 4 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -2 > gpurun_out/r06b_fuzz_hunt.txt; cat gpurun_out/r06b_fuzz_hunt.txt
DBG_SHARD_FUZZ_SEEDS=8000 timeout 3000 python -m pytest tests/test_gpu_shard_fuzz.py -x -q -n 4 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -2 > gpurun_out/r06b_shard_fuzz_hunt.txt; cat gpurun_out/r06b_shard_fuzz_hunt.txt
