"""Instruction budget of the counting kernel from its ISA listing (round-5 review, item 3).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-gpu-rdc -ffp-contract=off -S --cuda-device-only -DDBG_ASM_MARKS \
          rust-debruijn_amd/csrc/fastpath.hip -o /tmp/fastpath_marks.s
    python tools/isa_budget.py /tmp/fastpath_marks.s [KW NBW SET NT T WIDE WEIGHTED]     (default 2 3 1 512 2048 0 0: the C2 variant)

The compiler annotates every basic block with the loop it belongs to ("in Loop: Header=BBn_m Depth=d"); block layout does not follow
source order, so instructions are attributed to their INNERMOST LOOP, and the loop tree is printed with the static instruction count of
each loop's own blocks (child loops excluded) by issue class: vector ALU (quarter-rate 32-bit multiplies / v_mad_u64_u32 and half-rate
64-bit shifts apart), LDS, global memory, scalar, branches, waits.  The `; ##MARK` comments (fastpath.hip: MARK(...)) that fall into a
loop name it.  Static counts: one trip of each loop; divergent branches of a trip are all counted (they run one after the other)."""
import collections
import re
import sys


def classify(op):
    if op.startswith(("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mul_lo_i32")):
        return "v.mul32"
    if op.startswith(("v_mad_u64_u32", "v_mad_i64_i32")):
        return "v.mad64"
    if op.startswith(("v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64")):
        return "v.sh64"
    if op.startswith("v_cmp"):
        return "v.cmp"
    if op.startswith("v_cndmask"):
        return "v.sel"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "v.lane"
    if op.startswith("v_"):
        return "v.other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_barrier", "s_nop", "s_sleep")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


COLS = ["v.other", "v.cmp", "v.sel", "v.lane", "v.sh64", "v.mul32", "v.mad64", "lds", "vmem", "salu", "branch", "wait"]
WEIGHT = {"v.mul32": 4, "v.mad64": 4, "v.sh64": 2}


def main():
    path = sys.argv[1]
    t = [int(x) for x in sys.argv[2:9]] if len(sys.argv) >= 9 else [2, 3, 1, 512, 2048, 0, 0]
    name = "bin_count_kernelILi%dELi%dELb%dELi%dELi%dELb%dELb%dEE" % tuple(t)
    lines = open(path).read().splitlines()
    beg = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and name in l and l.split()[0].endswith(":"))
    end = next(i for i in range(beg, len(lines)) if lines[i].startswith(".Lfunc_end"))
    own = collections.defaultdict(collections.Counter)       # loop header -> class counts of its own blocks
    parent, depth, marks, ops = {}, {"-": 0}, collections.defaultdict(list), collections.defaultdict(collections.Counter)
    cur = "-"
    i = beg + 1
    while i < end:
        l = lines[i]
        m = re.match(r"^\.(LBB\d+_\d+):\s*(?:;\s*(.*))?$", l)
        if m:
            label, note = m.group(1), m.group(2) or ""
            hdr = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", note)
            par = None
            j = i + 1
            block_notes = [note]
            while j < end and lines[j].lstrip().startswith(";") and "##MARK" not in lines[j]:
                block_notes.append(lines[j]); j += 1
            txt = " ".join(block_notes)
            hdr = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", txt)
            if hdr:
                cur = label
                depth[cur] = int(hdr.group(1))
                pm = re.findall(r"Parent Loop (BB\d+_\d+) Depth=(\d+)", txt)
                parent[cur] = "L" + max(pm, key=lambda x: int(x[1]))[0] if pm else "-"
            else:
                inl = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", txt)
                cur = "L" + inl.group(1) if inl else "-"
            i += 1
            continue
        s = l.strip()
        mk = re.match(r";\s*##MARK\s+(\S+)", s)
        if mk:
            marks[cur].append(mk.group(1))
        elif s and not s.startswith((";", ".", "//")) and not s.endswith(":"):
            op = s.split()[0]
            own[cur][classify(op)] += 1
            if op.startswith("v_"):
                ops[cur][re.sub(r"_e(32|64)$|_dpp$|_sdwa$", "", op)] += 1
        i += 1
    children = collections.defaultdict(list)
    for k, p in parent.items():
        children[p].append(k)
    print("kernel %s: static instruction counts per loop (own blocks; child loops listed below their parent)" % name)
    print("%-34s %s | valu  issue*" % ("loop [marks inside]", " ".join("%7s" % c for c in COLS)))

    def show(k, ind):
        c = own.get(k, collections.Counter())
        valu = sum(v for n, v in c.items() if n.startswith("v."))
        cyc = sum(v * WEIGHT.get(n, 1) for n, v in c.items() if n.startswith("v."))
        tag = ("  " * ind + (k if k != "-" else "(straight-line code)") + (" [" + ",".join(marks[k]) + "]" if marks.get(k) else ""))[:34]
        print("%-34s %s | %4d  %5d" % (tag, " ".join("%7d" % c.get(x, 0) for x in COLS), valu, cyc))
        for ch in sorted(children.get(k, []), key=lambda x: int(x.split("_")[1])):
            show(ch, ind + 1)
    show("-", 0)
    print("* issue slots in units of one full-rate wave64 VALU instruction (4 clocks on a 16-lane SIMD): v_mul_lo/hi_u32 and v_mad_u64_u32 count 4, 64-bit shifts 2")
    if "--ops" in sys.argv:
        for k in sorted(ops, key=lambda x: -sum(ops[x].values()))[:6]:
            print("\nVALU opcodes of %s: %s" % (k, ", ".join("%s x%d" % kv for kv in ops[k].most_common(40))))


if __name__ == "__main__":
    main()
