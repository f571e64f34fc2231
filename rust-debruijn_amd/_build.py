"""Builds libdbg_mi355x.so (hand-written HIP kernels for gfx950 + the C ABI) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU; the built .so travels to the GPU box with the
repository snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libdbg_mi355x.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result", "-ffp-contract=off"] + os.environ.get("DBG_EXTRA_HIPCC_FLAGS", "").split()     # e.g. -DDBG_PHASE_TIMES (measurement builds)


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "dbg_mi355x.h"))
    return hs


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _check_bin_count_budget(remarks):
    """The counting kernel is laid out for two 512-thread workgroups per CU: 160 KB of LDS / 2 and 128 VGPRs.  A variant that
    slips over either limit silently halves its occupancy (it happened to the k >= 56 colour-set variant), so the build checks
    the compiler's resource report."""
    import re
    name = None
    matched = set()
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            continue
        # (the WIDE colour-set variants -- two more mask words per entry -- use 1024-entry tables: ...Li512ELi1024ELb1E)
        if name and "bin_count_kernel" in name and ("Li512ELi2048ELb0E" in name or "Li512ELi1024ELb1E" in name):
            m = re.search(r"LDS Size \[bytes/block\]: (\d+)", line)
            if m and int(m.group(1)) > 81920:
                raise RuntimeError("bin_count variant %s needs %s bytes of LDS: more than half a CU's 160 KB" % (name, m.group(1)))
            m = re.search(r"Occupancy \[waves/SIMD\]: (\d+)", line)
            if m and int(m.group(1)) < 4:
                raise RuntimeError("bin_count variant %s reaches %s waves per SIMD, not the 4 that two workgroups per CU need" % (name, m.group(1)))
            if m:
                matched.add(name)
    if not matched:
        raise RuntimeError("the resource check saw no bin_count_kernel<..,512,2048> variant in hipcc's remarks: the remark format or the "
                           "kernel's template signature changed -- update _check_bin_count_budget")


def _without_remarks(stderr):
    """hipcc's stderr minus the -Rpass-analysis resource report (remark lines and their source-line echo + caret), so that
    warnings for fastpath.hip are still shown"""
    out, skip = [], 0
    for line in stderr.splitlines(True):
        if "remark:" in line:
            skip = 2
            continue
        if skip and (line.lstrip().startswith("|") or "^" in line or "__global__" in line or line[:1] in " \t" or line.strip().split(" ")[0].isdigit()):
            skip -= 1
            continue
        skip = 0
        if "remarks generated" in line or "remark generated" in line:
            continue
        out.append(line)
    return "".join(out)


def _compile(src):
    obj = os.path.join(OBJ, src[:-4] + ".o")
    if _stale(obj, [os.path.join(CSRC, src)] + _headers()):
        extra = ["-Rpass-analysis=kernel-resource-usage"] if src == "fastpath.hip" else []
        cmd = [HIPCC] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if extra:
            try:
                _check_bin_count_budget(r.stderr)
            except RuntimeError:
                os.remove(obj)
                raise
        rest = _without_remarks(r.stderr) if extra else r.stderr
        if rest.strip():
            sys.stderr.write(rest)
    return obj


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    if _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
