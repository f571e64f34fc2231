"""Generates integration/dbg_mi355x_sys.rs -- the raw Rust binding of include/dbg_mi355x.h (what `bindgen` would emit, written out so
that a maintainer of the crate can drop it in as a `-sys` module without a build-time dependency): every `#[repr(C)]` struct, every
enum constant, every `extern "C"` function, and the `dbg_transport` table with its function-pointer fields.

No Rust toolchain exists in this image, so the file cannot be compiled here; tests/test_rust_ffi.py instead checks it against the header
and against the ctypes mirror (rust-debruijn_amd/_capi.py): every declared symbol is bound exactly once with the header's parameter count,
every struct has the header's fields in the header's order, and the C layout computed from the Rust field types (size and offset of every
field under the C ABI's natural alignment) equals ctypes.sizeof / field offsets of the mirror structure.

    python tools/gen_rust_ffi.py            # rewrites integration/dbg_mi355x_sys.rs
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dbg_mi355x.h")
OUT = os.path.join(ROOT, "integration", "dbg_mi355x_sys.rs")

PRIM = {"uint64_t": "u64", "uint32_t": "u32", "uint16_t": "u16", "uint8_t": "u8", "int32_t": "i32", "int64_t": "i64", "int": "c_int",
        "double": "f64", "float": "f32", "char": "c_char", "void": "c_void", "size_t": "usize"}
SIZES = {"u64": 8, "u32": 4, "u16": 2, "u8": 1, "i32": 4, "i64": 8, "c_int": 4, "f64": 8, "f32": 4, "c_char": 1, "usize": 8}


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def rust_type(ctype):
    """C declarator type (without the name) -> Rust type"""
    t = " ".join(ctype.replace("*", " * ").split())
    toks = t.split()
    const = "const" in toks
    toks = [x for x in toks if x not in ("const", "struct")]
    base, stars = toks[0], toks.count("*")
    r = PRIM.get(base, base)                                   # dbg_* struct names pass through
    for i in range(stars):
        # `const T*` -> *const T ; `T*` -> *mut T ; for T** the inner level keeps the constness, outer levels are *mut
        r = ("*const " if (const and i == 0) else "*mut ") + r
    return r


def split_decl(decl):
    """'const uint64_t* words' -> ('const uint64_t*', 'words', None) ; 'uint32_t labels[64]' -> ('uint32_t', 'labels', 64)"""
    decl = decl.strip()
    m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)\s*(\[\s*(\d+)\s*\])?$", decl)
    return m.group(1).strip(), m.group(2), int(m.group(4)) if m.group(4) else None


def parse(text):
    text = strip_comments(text)
    structs, enums, funcs = [], [], []
    for m in re.finditer(r"typedef\s+struct\s*(\w*)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        name, body, fields = m.group(3), m.group(2), []
        # function-pointer fields: int (*name)(args);
        for part in [p.strip() for p in body.split(";") if p.strip()]:
            fp = re.match(r"^(\w[\w\s\*]*?)\(\s*\*\s*(\w+)\s*\)\s*\((.*)\)$", part, flags=re.S)
            if fp:
                args = [split_decl(a) for a in fp.group(3).split(",")]
                # (a `void` return is no return type in Rust: `c_void` is only ever a pointee)
                fields.append(dict(name=fp.group(2), fnptr=True, ret=None if fp.group(1).strip() == "void" else rust_type(fp.group(1)), args=[(n, rust_type(t)) for t, n, _ in args]))
                continue
            ctype, first, arr = split_decl(part.split(",")[0])
            names = [(first, arr)]
            for extra in part.split(",")[1:]:                   # `int32_t rank, world;`
                _, n2, a2 = split_decl(ctype + " " + extra.strip())
                names.append((n2, a2))
            for n, a in names:
                fields.append(dict(name=n, fnptr=False, rtype=rust_type(ctype), array=a))
        structs.append((name, fields))
    for m in re.finditer(r"enum\s*\{(.*?)\}\s*;", text, flags=re.S):
        for item in m.group(1).split(","):
            if "=" in item:
                k, v = item.split("=")
                enums.append((k.strip(), int(v.strip())))
    for m in re.finditer(r"(?m)^#define\s+(DBG_[A-Z0-9_]+)\s+(\d+)\b", text):
        if m.group(1) != "DBG_MI355X_H":
            enums.append((m.group(1), int(m.group(2))))
    body = re.sub(r"typedef\s+struct\s*\w*\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    for m in re.finditer(r"(?m)^\s*((?:const\s+)?\w+\s*\**)\s*(dbg_\w+)\s*\(([^;{]*?)\)\s*;", body, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        params = [] if args in ("void", "") else [split_decl(a) for a in args.split(",")]
        funcs.append((name, rust_type(ret) if ret != "void" else None, [(n, rust_type(t)) for t, n, _ in params]))
    return structs, enums, funcs


KEYWORDS = {"self": "self_", "type": "type_", "in": "in_", "ref": "ref_", "box": "box_", "move": "move_", "out": "out"}


def ident(n):
    return KEYWORDS.get(n, n)


def emit(structs, enums, funcs):
    o = []
    o.append("//! Raw FFI of the MI355X-native k-mer hot path (`include/dbg_mi355x.h`): generated by `tools/gen_rust_ffi.py`, do not edit.")
    o.append("//! Link with `-ldbg_mi355x` (INTEGRATION.md section 1).  Safe wrappers shaped like `filter_kmers` / `msp_sequence` /")
    o.append("//! `compress_kmers_with_hash` are sketched in INTEGRATION.md sections 3-6; every function returns 0 on success, and a")
    o.append("//! non-zero return is where the reference would have panicked (`dbg_last_error`).")
    o.append("#![allow(non_camel_case_types, non_snake_case, dead_code)]")
    o.append("use std::os::raw::{c_char, c_int, c_void};")
    o.append("")
    o.append("#[repr(C)] pub struct dbg_ctx { _private: [u8; 0] }      // opaque")
    o.append("")
    for k, v in enums:
        o.append("pub const %s: c_int = %d;" % (k, v))
    o.append("")
    for name, fields in structs:
        o.append("#[repr(C)]")
        o.append("#[derive(Clone, Copy)]")
        o.append("pub struct %s {" % name)
        for f in fields:
            if f["fnptr"]:
                args = ", ".join("%s: %s" % (ident(n), t) for n, t in f["args"])
                o.append("    pub %s: Option<unsafe extern \"C\" fn(%s)%s>," % (ident(f["name"]), args, " -> %s" % f["ret"] if f["ret"] else ""))
            elif f["array"]:
                o.append("    pub %s: [%s; %d]," % (ident(f["name"]), f["rtype"], f["array"]))
            else:
                o.append("    pub %s: %s," % (ident(f["name"]), f["rtype"]))
        o.append("}")
        o.append("")
    o.append("extern \"C\" {")
    for name, ret, params in funcs:
        args = ", ".join("%s: %s" % (ident(n), t) for n, t in params)
        o.append("    pub fn %s(%s)%s;" % (name, args, " -> %s" % ret if ret else ""))
    o.append("}")
    o.append("")
    return "\n".join(o)


def c_layout(fields):
    """(size, [(name, offset)]) of a #[repr(C)] struct from its Rust field types (natural alignment, LP64)"""
    off, align_max, out = 0, 1, []
    for f in fields:
        if f["fnptr"] or f["rtype"].startswith("*"):
            size, align, count = 8, 8, 1
        else:
            size = align = SIZES[f["rtype"]]
            count = f["array"] or 1
        off = (off + align - 1) // align * align
        out.append((f["name"], off))
        off += size * count
        align_max = max(align_max, align)
    return (off + align_max - 1) // align_max * align_max, out


def main():
    structs, enums, funcs = parse(open(HEADER).read())
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w").write(emit(structs, enums, funcs))
    print("%s: %d structs, %d constants, %d functions" % (os.path.relpath(OUT, ROOT), len(structs), len(enums), len(funcs)))


if __name__ == "__main__":
    main()
