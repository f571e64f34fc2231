import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import oracle_lib as O, refgen as R
from pkg import dbg
os.environ["DBG_COMPRESS"] = "device"
ctx = dbg.Context(0)
rng = np.random.default_rng(1)
c = [R.random_dna(rng, 60)]
t, _ = dbg.filter_kmers([(x, 0, None) for x in c], dbg.CountFilter(1), False, False, 4, k=31, ctx=ctx)
for name in ("saturating_add", "max", "add_mod_65535", "wrapping_add"):
    g = dbg.compress_kmers_with_hash(False, dbg.SimpleCompress(name), t, k=31, ctx=ctx)
    print(name, len(g), g.data[:5], g.sequences.length[:5], g.exts[:5])
