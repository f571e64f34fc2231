"""Aggregates rocprofv3 --pmc counter_collection CSVs per kernel name (sum over dispatches)."""
import csv, glob, os, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
def newest_per_pass(root):
    """gpurun merges every call's files into the same local directory: keep only the newest run of each pass"""
    best = {}
    for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
        d = os.path.dirname(f)
        if d not in best or os.path.getmtime(f) > os.path.getmtime(best[d]):
            best[d] = f
    return sorted(best.values())


for f in newest_per_pass(root):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
names = sorted({c for k in agg for c in agg[k]})
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0)):
    print(k)
    for c in names:
        if c in agg[k]:
            print("    %-24s %16.0f   (%d dispatches)" % (c, agg[k][c], calls[k][c]))
