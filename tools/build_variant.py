"""Experimental build of the library: `python tools/build_variant.py NAME file.hip -DFOO=1 ...` recompiles the named source
files with the extra flags, links them with the regular objects of the others into rust-debruijn_amd/_exp/libNAME.so
(git-ignored, travels with gpurun); DBG_LIB=<that path> makes the Python mirror load it.  For A/B measurements only."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-debruijn_amd"))
import _build as B  # noqa: E402

name = sys.argv[1]
files = [a for a in sys.argv[2:] if a.endswith(".hip")]
flags = [a for a in sys.argv[2:] if not a.endswith(".hip")]
B.build()
exp = os.path.join(B.HERE, "_exp")
os.makedirs(exp, exist_ok=True)
objs = []
for src in B._sources():
    if src in files:
        o = os.path.join(exp, "%s_%s.o" % (name, src[:-4]))
        subprocess.check_call([B.HIPCC] + B.FLAGS + flags + ["-c", os.path.join(B.CSRC, src), "-o", o])
        objs.append(o)
    else:
        objs.append(os.path.join(B.OBJ, src[:-4] + ".o"))
lib = os.path.join(exp, "lib%s.so" % name)
subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
print(lib)
