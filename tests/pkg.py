"""Imports the product package (its directory name carries a hyphen, so go through importlib)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
dbg = importlib.import_module("rust-debruijn_amd")
capi = importlib.import_module("rust-debruijn_amd._capi")
D = importlib.import_module("rust-debruijn_amd.distributed")
