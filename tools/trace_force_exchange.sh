cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/trace_fx
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_fx -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-boundary --compress-reads 0 --no-other-shapes --force-exchange --backend nccl > $GRAFT_REPO_ROOT/gpurun_out/trace_fx.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace_fx/*/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:28], r.get("Stream_Id","?"), r.get("Queue_Id","?")) for r in csv.DictReader(open(f))]
rows.sort()
# last step: from the last sk_scan_lane on
idx = max(i for i, r in enumerate(rows) if r[2].startswith("sk_scan_lane"))
t0 = rows[idx][0]
for s, e, n, st, q in rows[idx:]:
    if any(n.startswith(p) for p in ("sk_scan", "slab_compact", "bin_count", "radix16_one", "span_sort", "ovf_base", "sk_scatter", "seg_tables")):
        print("%9.3f %9.3f  %-28s q=%s" % ((s - t0) / 1e6, (e - t0) / 1e6, n, q))
PY
