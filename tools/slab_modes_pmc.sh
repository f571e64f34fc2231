#!/bin/bash
# The two placement kinds of a 26 GB slab block (the same random 24-byte writes take 5.9 or 7.0-7.5 ms
# depending on the block): address-translation and L2 / fabric counters of tools/micro/slab_probe2.hip, per dispatch, so that a
# fast and a slow block of ONE process can be compared.  usage: tools/slab_modes_pmc.sh <tag>
TAG=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/slabmodes_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $OUT/slab_probe2 tools/micro/slab_probe2.hip || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
grep -o "UTCL[A-Z0-9_]*\|TCC_EA0_[A-Z0-9_]*\|TCC_TAG_STALL[A-Z0-9_]*\|TCP_TCC_[A-Z0-9_]*\|TCC_[A-Z]*STALL[A-Z0-9_]*\|GRBM_GUI_ACTIVE\|TCC_HIT\b\|TCC_MISS\b\|TCC_WRITEBACK\b\|TCC_REQ\b" $OUT/avail.txt | sort -u > $OUT/candidates.txt
echo "candidate counters:"; tr '\n' ' ' < $OUT/candidates.txt; echo
$OUT/slab_probe2 > $OUT/plain.txt 2>&1; cat $OUT/plain.txt
run() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- $OUT/slab_probe2 > $OUT/$n.log 2>&1 || tail -3 $OUT/$n.log; }
run p1 TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST TCP_UTCL1_PERMISSION_MISS
run p2 TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_STALL TCC_EA0_WR_UNCACHED_32B
run p3 TCC_HIT TCC_MISS TCC_WRITEBACK TCC_REQ
run p4 TCC_EA0_WRREQ_IO_CREDIT_STALL TCC_EA0_WRREQ_GMI_CREDIT_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_TOO_MANY_EA_WRREQS_STALL
run p5 TCC_EA0_WRREQ_DRAM TCC_EA0_ATOMIC TCC_TAG_STALL GRBM_GUI_ACTIVE
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections, os
out = "$OUT"
for p in sorted(glob.glob(os.path.join(out, "p*"))):
    if not os.path.isdir(p): continue
    fs = glob.glob(os.path.join(p, "*", "*counter_collection.csv"))
    if not fs:
        print(os.path.basename(p), "no counter file"); continue
    rows = collections.defaultdict(dict)
    for r in csv.DictReader(open(sorted(fs)[-1])):
        rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(rows)
    names = sorted({c for i in ids for c in rows[i]})
    print(os.path.basename(p), names)
    # dispatches come in the program's order: 2 rounds x 5 blocks x 3 repetitions
    for j, i in enumerate(ids):
        if j % 3 == 2:
            print("  round %d block %d: " % (j // 15, (j // 3) % 5) + "  ".join("%s=%.4g" % (c, rows[i].get(c, float('nan'))) for c in names))
PY
