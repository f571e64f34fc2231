#!/bin/bash
# First-contact kit for a multi-GPU MI355X node (round-5 review, item 7): everything the 1 -> 8 curve and the rank-spanning calls need
# checked the day the hardware appears, in order, each step under its own timeout, results in one SCALE-shaped JSON.
#
#   tools/first_contact_8gpu.sh [--dry-run] [--out gpurun_out/first_contact] [--worlds "2 4 8"]
#
# --dry-run: a one-GPU box -- every multi-rank step runs with all ranks on cuda:0 over gloo (--one-device --backend gloo), at reduced
#            sizes; steps that need real peers are listed as skipped.  It proves the script, not the hardware.
# Steps
#   0  devices, peer-access matrix
#   1  pytest tests/test_gpu_multirank.py::test_two_gpus_nccl                (RCCL world 2: first ever execution with world > 1)
#   2  digest check: bench.py --gpus W --reads 1e8/W --digest, W in worlds   (per-rank tables add up to the committed single-GPU
#                                                                             digest of C2: profiles/r06_digests.json)
#   3  weak scaling: bench.py --gpus 1 / W (C2: 1e8 reads per GPU, 20 steps, 5 warm-up) -> value, balance, exposed exchange per round
#   4  strong shapes: bench.py --gpus 8 --config c4 / c5                      (BASELINE configs 4 / 5 at their 8-GPU totals)
#   5  in-process transport, one thread-rank per DEVICE (peer copies): tests/cpp test_shard_threads, tools/rehearse_shard.py --per-device
#   6  fault injection with real peers: DBG_TEST_PER_DEVICE=1 pytest tests/test_gpu_shard_faults.py tests/test_gpu_shard_fuzz.py
#   7  summary JSON: <out>/first_contact.json  {n_gpus, value, ms_per_step, ranks_seen, balance, exchange, digest_ok} per run
set -u
DRY=0; OUT=gpurun_out/first_contact; WORLDS="2 4 8"
while [ $# -gt 0 ]; do case $1 in --dry-run) DRY=1;; --out) OUT=$2; shift;; --worlds) WORLDS=$2; shift;; esac; shift; done
mkdir -p $OUT
export PYTHONUNBUFFERED=1 MASTER_ADDR=127.0.0.1
NGPU=$(python3 -c "import torch; print(torch.cuda.device_count())")
log() { echo "[first-contact] $*" | tee -a $OUT/log.txt; }
Q="--no-cpu-baseline --no-host-boundary --compress-reads 0 --no-other-shapes"
if [ $DRY = 1 ]; then MODE="--one-device --backend gloo"; DIGEST_TOTAL=4000000; WEAK_READS="--reads 2000000"; STEPS="--steps 2 --warmup 1"; C4="--reads 4000000"; C5="--reads 4000000"
else MODE="--backend nccl"; DIGEST_TOTAL=100000000; WEAK_READS=""; STEPS="--steps 20 --warmup 5"; C4=""; C5=""; fi
: > $OUT/log.txt
log "devices visible: $NGPU (dry run: $DRY)"

# ---- 0: devices and peer access ----
python3 - > $OUT/step0_devices.txt 2>&1 <<PY
import torch
n = torch.cuda.device_count()
print("devices", n)
for i in range(n):
    print(i, torch.cuda.get_device_name(i), "%.0f GB" % (torch.cuda.get_device_properties(i).total_memory / 1e9),
          "peers:", [j for j in range(n) if j != i and torch.cuda.can_device_access_peer(i, j)])
PY
cat $OUT/step0_devices.txt | tee -a $OUT/log.txt

# ---- 1: RCCL with world 2 ----
if [ $DRY = 0 ] && [ $NGPU -ge 2 ]; then
  timeout 900 python -m pytest tests/test_gpu_multirank.py -x -q -k two_gpus_nccl 2>&1 | tail -3 | tee $OUT/step1_two_gpus_nccl.txt
else log "step 1 skipped (needs two GPUs): test_two_gpus_nccl"; echo skipped > $OUT/step1_two_gpus_nccl.txt; fi

# ---- 2: digests ----
run_bench() { # tag, args...
  local tag=$1; shift
  timeout 1800 python bench.py "$@" > $OUT/$tag.json 2> $OUT/$tag.err
  local rc=$?
  [ $rc = 0 ] || log "$tag: bench.py exited with $rc: $(tail -2 $OUT/$tag.err | tr '\n' ' ')"
  return $rc
}
run_bench digest_w1 --gpus 1 --reads $DIGEST_TOTAL --steps 1 --warmup 0 --digest $Q
for W in $WORLDS; do
  if [ $DRY = 0 ] && [ $NGPU -lt $W ]; then log "step 2 world $W skipped: $NGPU GPUs"; continue; fi
  run_bench digest_w$W --gpus $W --reads $((DIGEST_TOTAL / W)) --steps 1 --warmup 0 --digest $MODE $Q
done

# ---- 3: weak scaling (C2 per GPU) ----
run_bench weak_w1 --gpus 1 $WEAK_READS $STEPS $Q
for W in $WORLDS; do
  if [ $DRY = 0 ] && [ $NGPU -lt $W ]; then log "step 3 world $W skipped: $NGPU GPUs"; continue; fi
  run_bench weak_w$W --gpus $W $WEAK_READS $STEPS $MODE $Q
done

# ---- 4: BASELINE configs 4 / 5 at their totals (strong) ----
WMAX=$(echo $WORLDS | awk '{print $NF}')
if [ $DRY = 1 ] || [ $NGPU -ge $WMAX ]; then
  run_bench c4_w$WMAX --gpus $WMAX --config c4 $C4 --steps 3 --warmup 1 --digest $MODE $Q
  run_bench c5_w$WMAX --gpus $WMAX --config c5 $C5 --steps 3 --warmup 1 --digest $MODE $Q
else log "step 4 skipped: $NGPU GPUs"; fi

# ---- 5: in-process transport, one thread-rank per device ----
if [ $DRY = 0 ] && [ $NGPU -ge 2 ]; then
  python -c "import sys; sys.path.insert(0, 'tests'); import test_gpu_cpp_mirror as t; print(t.build_shard_threads())" > /dev/null
  timeout 900 tests/cpp/_build/test_shard_threads $NGPU $NGPU 2>&1 | tail -3 | tee $OUT/step5_shard_threads.txt
  timeout 1800 python tools/rehearse_shard.py --ranks $NGPU --reads-per-rank 12500000 --k 47 --per-device --out $OUT/step5_rehearse.txt 2>&1 | tail -4
else
  python -c "import sys; sys.path.insert(0, 'tests'); import test_gpu_cpp_mirror as t; print(t.build_shard_threads())" > /dev/null
  timeout 900 tests/cpp/_build/test_shard_threads 4 1 2>&1 | tail -2 | tee $OUT/step5_shard_threads.txt
  timeout 900 python tools/rehearse_shard.py --ranks 4 --reads-per-rank 500000 --k 47 --out $OUT/step5_rehearse.txt 2>&1 | tail -3
  log "step 5 ran with every thread-rank on cuda:0 (dry run): peer copies not exercised"
fi

# ---- 6: fault injection with real peers ----
if [ $DRY = 0 ] && [ $NGPU -ge 2 ]; then
  DBG_TEST_PER_DEVICE=1 timeout 1800 python -m pytest tests/test_gpu_shard_faults.py tests/test_gpu_cpp_mirror.py -x -q -k "fault or per_device" 2>&1 | tail -3 | tee $OUT/step6_faults.txt
else
  timeout 1200 python -m pytest tests/test_gpu_shard_faults.py -x -q 2>&1 | tail -2 | tee $OUT/step6_faults.txt
  log "step 6 ran on one device (dry run)"
fi

# ---- 7: summary ----
python3 - $OUT $DRY <<'PY' | tee -a $OUT/log.txt
import glob, json, os, sys
out, dry = sys.argv[1], sys.argv[2] == "1"
def line(f):
    try:
        return json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    except Exception as e:
        return {"error": "%s: %s" % (os.path.basename(f), e)}
runs, dig = {}, {}
for f in sorted(glob.glob(os.path.join(out, "*.json"))):
    tag = os.path.basename(f)[:-5]
    if tag == "first_contact":
        continue
    j = line(f)
    runs[tag] = {k: j.get(k) for k in ("n_gpus", "value", "unit", "ms_per_step", "scaling", "ranks_seen", "backend", "table_digest", "balance", "exchange", "error")}
    runs[tag]["valid_kmers_all_ranks"] = (j.get("config") or {}).get("valid_kmers_all_ranks")
    if tag.startswith("digest_w"):
        dig[tag] = (runs[tag]["valid_kmers_all_ranks"], j.get("table_digest"))
ref = dig.get("digest_w1")
committed = None
try:
    committed = json.load(open("profiles/r06_digests.json")).get("c2_1e8_reads_k47_set")
except Exception:
    pass
summary = {"dry_run": dry, "runs": runs,
           "digest_ok": {t: (v == ref) for t, v in dig.items()},
           "digest_w1_matches_committed": (None if (dry or not committed or not ref) else [ref[0], ref[1]] == committed),
           "weak_scaling": {t: runs[t]["value"] for t in runs if t.startswith("weak_w")}}
w1 = summary["weak_scaling"].get("weak_w1")
if w1:
    summary["speedup_vs_1gpu"] = {t: round(v / w1, 3) for t, v in summary["weak_scaling"].items() if v}
json.dump(summary, open(os.path.join(out, "first_contact.json"), "w"), indent=1)
print(json.dumps({k: summary[k] for k in ("dry_run", "digest_ok", "digest_w1_matches_committed", "weak_scaling", "speedup_vs_1gpu") if k in summary}))
PY
