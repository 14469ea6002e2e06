"""CPU twin of a device expression (test infrastructure): the SAME expression text wrapped into a host function with the semantics
include/pasture_amd.h documents (names v, x y z, c, i, p0 .. p3; result converted to T with Rust `as` -- the oracle's rust_as), compiled by
g++ with -ffp-contract=off into a throw-away shared object.  Parity of the hipRTC-compiled device function is asserted against it bit for bit."""
import ctypes
import hashlib
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_CT = {"u8": "uint8_t", "i8": "int8_t", "u16": "uint16_t", "i16": "int16_t", "u32": "uint32_t", "i32": "int32_t", "u64": "uint64_t", "i64": "int64_t", "f32": "float", "f64": "double"}
_NP = {"u8": np.uint8, "i8": np.int8, "u16": np.uint16, "i16": np.int16, "u32": np.uint32, "i32": np.int32, "u64": np.uint64, "i64": np.int64, "f32": np.float32, "f64": np.float64}
_CACHE = {}
_DIR = tempfile.mkdtemp(prefix="pst_expr_twin_")


def _split(expr):
    out, cur, depth = [], "", 0
    for ch in expr:
        depth += ch in "([" 
        depth -= ch in ")]"
        if ch == ";" and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    if len(out) > 1 and not out[-1].strip():
        out.pop()
    return out


def _build(source):
    key = hashlib.sha1(source.encode()).hexdigest()
    if key in _CACHE:
        return _CACHE[key]
    cpp, so = os.path.join(_DIR, key + ".cpp"), os.path.join(_DIR, key + ".so")
    with open(cpp, "w") as f:
        f.write(source)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I", os.path.join(ROOT, "oracle"), "-o", so, cpp])
    _CACHE[key] = ctypes.CDLL(so)
    return _CACHE[key]


def map_twin(src_ct, dst_ct, ncomp, pre, expr):
    """-> f(values [n][ncomp] of src type, first_index, params) -> [n][ncomp] of dst type"""
    comps = _split(expr)
    assert len(comps) in (1, ncomp)
    t = '#include <cmath>\n#include <cstdint>\n#include <cstring>\n#include "pasture_oracle.hpp"\nusing orc::rust_as;\n'
    t += f"typedef {_CT[src_ct]} TS;\ntypedef {_CT[dst_ct]} TD;\ntypedef {'TS' if pre else 'TD'} TI;\n"
    for c in range(ncomp):
        t += (f"static inline TI pst_expr_{c}(const TI v, const TI x, const TI y, const TI z, const int c, const uint64_t i, const double* p0, const double* p1, "
              f"const double* p2, const double* p3) {{\n  (void)v; (void)x; (void)y; (void)z; (void)c; (void)i; (void)p0; (void)p1; (void)p2; (void)p3;\n"
              f"  return rust_as<TI>(\n{comps[0 if len(comps) == 1 else c]}\n  );\n}}\n")
    t += ('extern "C" void twin(const TS* src, TD* dst, uint64_t n, uint64_t first, const double* p0, const double* p1, const double* p2, const double* p3) {\n'
          f"  const int NC = {ncomp};\n  for (uint64_t e = 0; e < n; ++e) {{\n    const uint64_t i = first + e;\n    TI in[3];\n"
          f"    for (int c = 0; c < NC; ++c) in[c] = {'(TI)src[e * NC + c]' if pre else '(TI)rust_as<TD>(src[e * NC + c])'};\n"
          "    for (int c = NC; c < 3; ++c) in[c] = in[0];\n    TI r[3];\n")
    for c in range(ncomp):
        t += f"    r[{c}] = pst_expr_{c}(in[{c}], in[0], in[1], in[2], {c}, i, p0, p1, p2, p3);\n"
    t += "    for (int c = 0; c < NC; ++c) dst[e * NC + c] = rust_as<TD>(r[c]);\n  }\n}\n"
    lib = _build(t)

    def run(values, first=0, params=()):
        values = np.ascontiguousarray(values, dtype=_NP[src_ct]).reshape(-1, ncomp)
        out = np.zeros(values.shape, dtype=_NP[dst_ct])
        ps = [np.ascontiguousarray(p, dtype=np.float64) for p in params] + [None] * (4 - len(params))
        lib.twin(values.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(len(values)), ctypes.c_uint64(first),
                 *[p.ctypes.data_as(ctypes.c_void_p) if p is not None else None for p in ps])
        return out if ncomp > 1 else out.reshape(-1)
    return run


def pred_twin(attrs, expr):
    """attrs: [(name, ct, ncomp)] -> f({name: array}, first_index, params) -> uint8 mask"""
    t = '#include <cmath>\n#include <cstdint>\ntemplate <typename T> struct PstV3 { T x, y, z; };\n'
    params = ", ".join(f"const {('PstV3<' + _CT[ct] + '>') if nc == 3 else _CT[ct]} {name}" for name, ct, nc in attrs)
    t += (f"static inline bool pst_pred({params}{', ' if attrs else ''}const uint64_t i, const double* p0, const double* p1, const double* p2, const double* p3) {{\n"
          f"  (void)i; (void)p0; (void)p1; (void)p2; (void)p3;\n  return (bool)(\n{expr}\n  );\n}}\n")
    t += 'extern "C" void twin(const void* const* cols, uint8_t* mask, uint64_t n, uint64_t first, const double* p0, const double* p1, const double* p2, const double* p3) {\n  for (uint64_t e = 0; e < n; ++e) {\n'
    args = []
    for a, (name, ct, nc) in enumerate(attrs):
        T = _CT[ct]
        if nc == 3:
            t += f"    const PstV3<{T}> v{a} = {{((const {T}*)cols[{a}])[3 * e], ((const {T}*)cols[{a}])[3 * e + 1], ((const {T}*)cols[{a}])[3 * e + 2]}};\n"
        else:
            t += f"    const {T} v{a} = ((const {T}*)cols[{a}])[e];\n"
        args.append(f"v{a}")
    t += f"    mask[e] = pst_pred({', '.join(args)}{', ' if args else ''}first + e, p0, p1, p2, p3) ? 1 : 0;\n  }}\n}}\n"
    lib = _build(t)

    def run(columns, n, first=0, params=()):
        arrs = [np.ascontiguousarray(columns[name], dtype=_NP[ct]) for name, ct, nc in attrs]
        ptrs = (ctypes.c_void_p * max(1, len(arrs)))(*[a.ctypes.data for a in arrs])
        mask = np.zeros(n, dtype=np.uint8)
        ps = [np.ascontiguousarray(p, dtype=np.float64) for p in params] + [None] * (4 - len(params))
        lib.twin(ptrs, mask.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(n), ctypes.c_uint64(first),
                 *[p.ctypes.data_as(ctypes.c_void_p) if p is not None else None for p in ps])
        return mask
    return run
