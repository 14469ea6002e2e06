#!/usr/bin/env python3
"""Generate rust/pasture-amd-sys/src/lib.rs (extern "C" declarations) from include/pasture_amd.h.
The image has no Rust toolchain: the generated crate is UNCOMPILED; it exists so that the Rust side of the boundary is
concrete and stays in sync with the header (tests/test_capi_symbols.py checks that every declared symbol is listed)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hdr = open(os.path.join(ROOT, "include", "pasture_amd.h")).read()
hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)

TYPES = {"int": "c_int", "uint8_t": "u8", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize", "double": "f64", "float": "f32", "int64_t": "i64", "void": "c_void",
         "char": "c_char", "pst_layout": "pst_layout", "pst_buffer": "pst_buffer", "pst_converter": "pst_converter",
         "pst_datatype": "pst_datatype", "pst_member": "pst_member", "pst_transform": "pst_transform", "pst_mapping_info": "pst_mapping_info",
         "pst_point_converter": "pst_point_converter", "pst_comm": "pst_comm", "pst_comm_id": "pst_comm_id", "pst_jit_stats": "pst_jit_stats", "pst_voxel_plan": "pst_voxel_plan", "pst_normals_plan": "pst_normals_plan"}
# C parameter names that are Rust keywords (`self` as a parameter NAME of an extern fn is a compile error)
RENAME = {"type": "ty", "self": "this", "in": "input", "ref": "reference", "fn": "func", "move": "mv", "match": "matched"}


def rust_type(c: str) -> str:
    c = c.strip()
    m = re.match(r"^(const\s+)?(\w+)\s*((?:\*\s*(?:const\s*)?)*)$", c)
    assert m, c
    const, base, ptrs = bool(m.group(1)), m.group(2), m.group(3)
    t = TYPES[base]
    stars = ptrs.count("*")
    if stars == 0:
        return t
    out = t
    # innermost pointer carries the constness of the pointee; `void* const*` => *const *mut c_void
    pieces = re.findall(r"\*\s*(const)?", ptrs)
    for i, pc in enumerate(pieces):
        pointee_const = const if i == 0 else bool(pieces[i - 1])
        out = ("*const " if pointee_const else "*mut ") + out
    return out


decls = []
for m in re.finditer(r"\n(const char\*|int)\s+(pst_\w+)\s*\(([^;]*?)\)\s*;", hdr):
    ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
    params = []
    if args != "void":
        for a in args.split(","):
            a = a.strip()
            am = re.match(r"^(.*?)(\w+)(\[\d*\])?$", a)
            ctype, pname, arr = am.group(1).strip(), am.group(2), am.group(3)
            if arr:
                ctype += "*"
            params.append(f"{RENAME.get(pname, pname)}: {rust_type(ctype)}")
    r = "*const c_char" if ret.startswith("const char") else "c_int"
    decls.append(f"    pub fn {name}({', '.join(params)}) -> {r};")

# ---- the #[repr(C)] structs: ONE model, from which both the Rust text and a C file of _Static_assert(sizeof / offsetof) are generated.
# The C file is compiled against include/pasture_amd.h by tests/test_capi_symbols.py: without rustc it is the mechanical check that the
# layout the Rust declarations imply (repr(C): fields in order, each at the next multiple of its alignment, size rounded up to the
# struct's alignment) is the layout the C compiler gives the header's structs.
PRIM = {"u8": (1, 1), "u32": (4, 4), "i32": (4, 4), "u64": (8, 8), "f64": (8, 8), "*const c_char": (8, 8)}
STRUCTS = [
    ("pst_comm_id", "Clone, Copy", [("bytes", "[u8; 128]")]),
    ("pst_datatype", "Clone, Copy, Debug, Default", [("kind", "u32"), ("reserved", "u32"), ("size_param", "u64"), ("align_param", "u64"), ("uuid", "[u8; 16]")]),
    ("pst_member", "Clone, Copy", [("name", "*const c_char"), ("datatype", "pst_datatype"), ("offset", "u64"), ("size", "u64")]),
    ("pst_transform", "Clone, Copy", [("kind", "u32"), ("shift", "u32"), ("datatype", "pst_datatype"), ("scale", "[f64; 3]"), ("offset", "[f64; 3]"), ("mask", "u64")]),
    ("pst_mapping_info", "Clone, Copy", [("source_name", "*const c_char"), ("target_name", "*const c_char"), ("source_datatype", "pst_datatype"),
                                         ("target_datatype", "pst_datatype"), ("source_offset", "u64"), ("target_offset", "u64"), ("has_converter", "i32"),
                                         ("transform_kind", "u32"), ("apply_to_source", "i32"), ("reserved", "i32")]),
    ("pst_jit_stats", "Clone, Copy, Debug, Default", [("compiled", "u64"), ("disk_hits", "u64"), ("memory_hits", "u64"), ("failures", "u64"), ("launches", "u64"),
                                                      ("compile_seconds", "f64")]),
]
LAYOUT = {}


def size_align(t):
    if t in PRIM:
        return PRIM[t]
    m = re.match(r"^\[(\w+); (\d+)\]$", t)
    if m:
        sz, al = size_align(m.group(1))
        return sz * int(m.group(2)), al
    return LAYOUT[t]["size"], LAYOUT[t]["align"]


for sname, _derive, fields in STRUCTS:
    off, align, offs = 0, 1, []
    for fname, ftype in fields:
        sz, al = size_align(ftype)
        off = (off + al - 1) // al * al
        offs.append((fname, off, sz))
        off += sz
        align = max(align, al)
    LAYOUT[sname] = {"size": (off + align - 1) // align * align, "align": align, "fields": offs}

rust_structs = "\n".join(
    f"#[repr(C)] #[derive({derive})]\npub struct {sname} {{ " + ", ".join(f"pub {fn}: {ft}" for fn, ft in fields) + " }"
    for sname, derive, fields in STRUCTS)

body = """//! Raw FFI declarations for libpasture_amd.so — GENERATED by tools/gen_rust_sys.py from include/pasture_amd.h.
//! UNCOMPILED: the build image has no Rust toolchain (see INTEGRATION.md).  The struct layouts below are pinned against the header by
//! tests/abi/rust_layout_asserts.c (generated from the same model, compiled by tests/test_capi_symbols.py).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] pub struct pst_layout { _private: [u8; 0] }
#[repr(C)] pub struct pst_buffer { _private: [u8; 0] }
#[repr(C)] pub struct pst_converter { _private: [u8; 0] }
#[repr(C)] pub struct pst_point_converter { _private: [u8; 0] }
#[repr(C)] pub struct pst_comm { _private: [u8; 0] }
#[repr(C)] pub struct pst_voxel_plan { _private: [u8; 0] }
#[repr(C)] pub struct pst_normals_plan { _private: [u8; 0] }

""" + rust_structs + """

pub const PST_OK: c_int = 0;
pub const PST_XF_AFFINE: u32 = 1;
pub const PST_XF_BITFIELD: u32 = 2;
pub const PST_STORAGE_INTERLEAVED: u32 = 0;
pub const PST_STORAGE_COLUMNAR: u32 = 1;
pub const PST_MEM_DEVICE: u32 = 0;
pub const PST_MEM_PINNED_HOST: u32 = 1;

#[link(name = "pasture_amd")]
extern "C" {
""" + "\n".join(decls) + "\n}\n"
open(os.path.join(ROOT, "rust", "pasture-amd-sys", "src", "lib.rs"), "w").write(body)

asserts = ["/* GENERATED by tools/gen_rust_sys.py -- the layout rust/pasture-amd-sys declares (#[repr(C)]) against the header's structs.",
           "   Compiled (not run) by tests/test_capi_symbols.py: gcc -std=c11 -c -Iinclude tests/abi/rust_layout_asserts.c */",
           "#include <stddef.h>", '#include "pasture_amd.h"', ""]
for sname, _derive, _fields in STRUCTS:
    lay = LAYOUT[sname]
    asserts.append(f'_Static_assert(sizeof({sname}) == {lay["size"]}, "sizeof({sname})");')
    asserts.append(f'_Static_assert(_Alignof({sname}) == {lay["align"]}, "alignof({sname})");')
    for fname, off, sz in lay["fields"]:
        asserts.append(f'_Static_assert(offsetof({sname}, {fname}) == {off}, "offsetof({sname}, {fname})");')
        asserts.append(f'_Static_assert(sizeof((({sname}*)0)->{fname}) == {sz}, "sizeof({sname}.{fname})");')
    asserts.append("")
os.makedirs(os.path.join(ROOT, "tests", "abi"), exist_ok=True)
open(os.path.join(ROOT, "tests", "abi", "rust_layout_asserts.c"), "w").write("\n".join(asserts))
print(len(decls), "declarations,", len(STRUCTS), "struct layouts")
