"""CPU: integration/dbg_mi355x_sys.rs, the generated raw Rust binding of include/dbg_mi355x.h (tools/gen_rust_ffi.py).  There is no Rust
toolchain in this image, so the file is checked structurally: it is what the generator produces from the header as it stands; it binds
every symbol the library exports, once, with the header's parameter count; and the C layout that follows from the Rust field types
(#[repr(C)], natural alignment) equals the layout of the ctypes mirror the GPU tests call the library through -- size and every field
offset -- so that a field added to the header but not to one of the two mirrors cannot go unnoticed."""
import ctypes as C
import importlib.util
import os
import re

from pkg import capi, ROOT

spec = importlib.util.spec_from_file_location("gen_rust_ffi", os.path.join(ROOT, "tools", "gen_rust_ffi.py"))
G = importlib.util.module_from_spec(spec)
spec.loader.exec_module(G)

MIRROR = {"dbg_seqset": capi.SeqSet, "dbg_filter_params": capi.FilterParams, "dbg_kmer_table": capi.KmerTable, "dbg_msp_params": capi.MspParams,
          "dbg_msp_pieces": capi.MspPieces, "dbg_graph": capi.Graph, "dbg_label_classes": capi.LabelClasses, "dbg_edges": capi.Edges,
          "dbg_shard_plan": capi.ShardPlan, "dbg_transport": capi.Transport, "dbg_shard_params": capi.ShardParams,
          "dbg_shard_stats": capi.ShardStats, "dbg_synth_params": capi.SynthParams, "dbg_kernel_time": capi.KernelTime,
          "dbg_ctx_stats": capi.CtxStats}


def parsed():
    return G.parse(open(G.HEADER).read())


def test_generated_file_is_current():
    structs, enums, funcs = parsed()
    assert open(G.OUT).read() == G.emit(structs, enums, funcs), "run python tools/gen_rust_ffi.py"


def test_every_export_is_bound_once():
    structs, enums, funcs = parsed()
    names = [f[0] for f in funcs]
    assert sorted(names) == sorted(capi.EXPORTS) and len(set(names)) == len(names)
    text = open(G.OUT).read()
    for name, ret, params in funcs:
        m = re.search(r"pub fn %s\((.*?)\)( -> [^;]+)?;" % name, text)
        assert m, name
        assert len([a for a in m.group(1).split(",") if a.strip()]) == len(params)
    assert {k for k, _ in enums} >= {"DBG_COUNT_FILTER_SET", "DBG_SPEC_SCMAP_EQ", "DBG_REDUCE_TREE", "DBG_SERDE_BINCODE"}


def test_struct_layouts_match_the_ctypes_mirror():
    structs, _, _ = parsed()
    assert {n for n, _ in structs} == set(MIRROR)
    for name, fields in structs:
        cls = MIRROR[name]
        size, offs = G.c_layout(fields)
        assert size == C.sizeof(cls), (name, size, C.sizeof(cls))
        assert [n for n, _ in offs] == [f[0] for f in cls._fields_], name          # same fields, same order
        for fname, off in offs:
            assert getattr(cls, fname).offset == off, (name, fname)
