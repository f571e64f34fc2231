# Round 6, item 1 of the round-5 review: the driver's bench line (python3 bench.py --gpus 1 --steps 20 --warmup 5) against the
# builder's (3 / 1): fresh processes, per-step series of the scan, slab backing + write probe in every line.
# usage: tools/r06_repro.sh [tag] [n]
R=${1:-r06a}; N=${2:-4}
export PYTHONUNBUFFERED=1
Q="--no-cpu-baseline --no-host-boundary --compress-reads 0 --no-other-shapes"
for i in $(seq 1 $N); do
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 $Q > gpurun_out/${R}_s20_$i.json 2> gpurun_out/${R}_s20_$i.err
done
for i in 1 2; do
  timeout 600 python3 bench.py --gpus 1 --steps 3 --warmup 1 $Q > gpurun_out/${R}_s3_$i.json 2> gpurun_out/${R}_s3_$i.err
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${R}_s*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j["roofline"]
        print(f, j["value"], j["ms_per_step"], "scan", r["kernel_ms_min_med_max"].get("sk_scan"), "count", r["kernel_ms_min_med_max"].get("bin_count"),
              "slab", {k: j["slab"].get(k) for k in ("backing", "mb_per_handle", "handles", "probe_ms", "note")})
        print("   scan by step", r["kernel_ms_by_step"]["sk_scan"])
        print("   alloc", j["slab"]["alloc"])
    except Exception as e:
        print(f, "ERR", e)
PY
