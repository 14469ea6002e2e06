"""Device expressions (include/pasture_amd.h "device expressions", pasture_amd/csrc/expr.cpp): user-written closures as source text.

What the reference takes as closures -- Fn(T) -> T transformations (buffer_conversion.rs:13-36, 194-234), transform_attribute's
Fn(usize, T) -> T (point_buffer.rs:391-404), filter's Fn(usize) -> bool (point_buffer.rs:1064-1136) -- the library takes as C++ expression
text and compiles with hipRTC.  CPU suite: the generated translation units compile for gfx950 (no device), shape errors are reported.
GPU suite: the reference's own scenarios (`+ 42.0` before and after the conversion, buffer_conversion.rs:764-846; positions overwritten by
index, point_buffer.rs:2007-2043) with numpy expectations, and a differential suite against the SAME text compiled by g++ (tests/expr_twin.py),
bit for bit, over datatypes, storage pairings, apply_to_source, parameters and slices."""
import ctypes
import os

import numpy as np
import pytest

import expr_twin
from harness import BUFFER_KINDS, PAIRINGS, custom_point_type_big, make_buffer, random_records
from pasture_amd import conversion as cv, las
from pasture_amd._capi import PastureError
from pasture_amd.algorithms import transform_attribute_expr
from pasture_amd.buffers import HashMapBuffer, VectorBuffer
from pasture_amd.conversion import BufferLayoutConverter
from pasture_amd.layout import PointAttributeDataType as T, PointAttributeDefinition, PointLayout, attributes as A

SCALARS = {"u8": T.U8, "i8": T.I8, "u16": T.U16, "i16": T.I16, "u32": T.U32, "i32": T.I32, "u64": T.U64, "i64": T.I64, "f32": T.F32, "f64": T.F64}
VEC3 = {"u8": T.Vec3u8, "u16": T.Vec3u16, "i32": T.Vec3i32, "f32": T.Vec3f32, "f64": T.Vec3f64}

# expression texts: (text, works on integer T, works on float T).  No libm functions whose last bit differs between libraries (sin, exp, pow):
# arithmetic, sqrt, floor / fabs / fmin / fmax, comparisons, casts, bit operations.
EXPRS = [
    ("v + 42.0", True, True),
    ("v * 0.001 + 500000.0", True, True),
    ("(v > x ? v : x) - z * 0.5", True, True),
    ("sqrt(fabs((double)v)) + (double)c", True, True),
    ("floor((double)v / 3.0) + (double)(i % 7)", True, True),
    ("fmin((double)x, fmax((double)y, (double)z)) * 2.0 - 1.0", True, True),
    ("p0[i % 13] * (double)v + p1[c]", True, True),
    ("c == 0 ? y : (c == 1 ? z : x)", True, True),
    ("(v ^ (v >> 1)) + 3", True, False),
    ("(uint32_t)v * 2654435761u + (uint32_t)i", True, False),
    ("v == v ? -v : 0.0", False, True),
    ("x * 2.0 ; y - z ; (double)c + floor((double)z)", True, True),
]


# ---- CPU suite: generator + headers under hipRTC, shape errors ---------------------------------------------------------------------------
@pytest.mark.parametrize("text,ints,floats", EXPRS)
def test_transformation_sources_compile_with_hiprtc(hip, text, ints, floats):
    for name, dt in list(SCALARS.items()) + [("v" + k, v) for k, v in VEC3.items()]:
        is_float = name.endswith("f32") or name.endswith("f64")
        if (is_float and not floats) or (not is_float and not ints) or (";" in text and not name.startswith("v")):
            continue
        for pre in (False, True):
            src = cv.expr_source("transform", text, src_datatype=dt, dst_datatype=dt, apply_to_source=pre, api=hip)
            assert "pst_jit_expr_map" in src and text.split(";")[0].strip() in src
            if name in ("u8", "f64", "vf64", "vu16", "i64"):  # (a compile takes ~0.3 s: a spread of types, not all of them)
                assert cv.jit_compile_source(src, api=hip)[:4] == b"\x7fELF"


def test_predicate_source_names_only_the_referenced_attributes(hip):
    layout = las.point_layout_from_las_point_format(las.Format(3), False, api=hip)
    src = cv.expr_source("predicate", "Classification == 2 && Position3D.z < 120.0 && (i & 1) == 0 && p0[i] > 0.5", layout=layout, api=hip)
    assert "const PstV3<double> Position3D, const uint8_t Classification, const uint64_t i" in src
    assert "Intensity" not in src and "ColorRGB" not in src
    assert cv.jit_compile_source(src, api=hip)[:4] == b"\x7fELF"
    src = cv.expr_source("predicate", "ColorRGB.x > ColorRGB.y || GpsTime < 1e5", layout=layout, api=hip)
    assert "const double GpsTime, const PstV3<uint16_t> ColorRGB" in src


def test_expression_shape_errors(hip):
    for bad in ("", "v; }", "v + 1 # x", 'v + "a"', "v;\nw", "x ; y"):  # statements / preprocessor / strings / two components for a Vec3
        with pytest.raises(PastureError) as e:
            cv.expr_source("transform", bad, src_datatype=T.Vec3f64, dst_datatype=T.Vec3f64, api=hip)
        assert e.value.code == 7, bad
    with pytest.raises(PastureError) as e:  # scalar <-> Vec3 has no conversion, Vec4u8 no components the expression could name
        cv.expr_source("transform", "v", src_datatype=T.Vec4u8, dst_datatype=T.Vec4u8, api=hip)
    assert e.value.code == 7
    layout = PointLayout.from_attributes([A.POSITION_3D, A.INTENSITY], api=hip)
    conv = BufferLayoutConverter.for_layouts(layout, layout)
    with pytest.raises(PastureError) as e:
        conv.set_custom_mapping_with_expression(A.INTENSITY, A.INTENSITY, "v ; v", False)
    assert e.value.code == 7
    with pytest.raises(PastureError) as e:  # attribute lookups keep the reference's panics
        conv.set_custom_mapping_with_expression(A.CLASSIFICATION, A.INTENSITY, "v", False)
    assert e.value.code == 4
    with pytest.raises(PastureError) as e:
        cv.jit_compile_source(cv.expr_source("transform", "v +* 2", src_datatype=T.F64, dst_datatype=T.F64, api=hip), api=hip)
    assert "error" in str(e.value)


@pytest.mark.parametrize("src_kind,dst_kind", [("V", "V"), ("V", "H"), ("H", "V")])
@pytest.mark.parametrize("apply_to_source", [False, True])
def test_expression_mappings_are_part_of_the_plan_specialised_kernel(hip, src_kind, dst_kind, apply_to_source):
    """Round 6 (review: closures are applied INSIDE the reference's conversion loop, buffer_conversion.rs:569-590): wherever a side is interleaved
    the expression is a device function of the plan's own translation unit -- one kernel, no strided pass of its own.  CPU: the generated unit of the
    reference's `+ 42.0` scenario (CustomPointTypeBig -> [POSITION_3D]) and of a converting Vec3 mapping with three component texts carries the
    text(s) and compiles for gfx950."""
    big = custom_point_type_big(hip)
    custom = PointLayout.from_attributes([A.POSITION_3D], api=hip)
    conv = BufferLayoutConverter.for_layouts_with_default(big, custom)
    conv.set_custom_mapping_with_expression(A.POSITION_3D, A.POSITION_3D, "v + 42.0", apply_to_source)
    unit = conv.jit_source(BUFFER_KINDS[src_kind], BUFFER_KINDS[dst_kind])
    assert "v + 42.0" in unit and "static __forceinline__ TI expr(" in unit and "pst_jit_convert" in unit
    assert cv.jit_compile_source(unit, api=hip)[:4] == b"\x7fELF"
    sa, da = PointAttributeDefinition("Value", T.Vec3i32), PointAttributeDefinition("Value", T.Vec3f32)
    sl = PointLayout.from_attributes_packed([A.CLASSIFICATION, sa, A.GPS_TIME], 1, api=hip)
    dl = PointLayout.from_attributes_packed([A.GPS_TIME, da, A.CLASSIFICATION], 1, api=hip)
    conv = BufferLayoutConverter.for_layouts(sl, dl)
    conv.set_custom_mapping_with_expression(sa, da, "x * 2.0 ; y - z ; (double)c + floor((double)z) + (double)(i % 7)", apply_to_source)
    unit = conv.jit_source(BUFFER_KINDS[src_kind], BUFFER_KINDS[dst_kind])
    assert "if constexpr (M == 1 && C == 2) return rust_as<TI>(" in unit and "y - z" in unit
    assert cv.jit_compile_source(unit, api=hip)[:4] == b"\x7fELF"
    # without an expression the unit is what it was (the in-tree instantiations are matched by their text)
    plain = BufferLayoutConverter.for_layouts(sl, dl).jit_source(BUFFER_KINDS[src_kind], BUFFER_KINDS[dst_kind])
    assert "expr(" not in plain


def test_predicates_are_part_of_the_compaction_kernels(hip):
    """Round 6 (review: filter takes ANY predicate and evaluates it in its own loop, point_buffer.rs:1064-1136): for layouts that take the streaming
    compaction kernel the predicate is a device function of the count pass (matches per tile, straight from the columns it names) and of the scatter
    pass (evaluated again on the values the kernel holds in registers) -- no byte mask.  CPU: both translation units carry the text and compile."""
    layout = las.point_layout_from_las_point_format(las.Format(3), False, api=hip)
    text = "Classification == 2 && Position3D.z < 120.0 && (i & 1) == 0 && p0[i % 7] > 0.5 && ColorRGB.y >= ColorRGB.x"
    count = cv.expr_source("predicate-count", text, layout=layout, api=hip)
    assert "pst_jit_pred_count" in count and "counts[tile] = c" in count and text in count
    assert cv.jit_compile_source(count, api=hip)[:4] == b"\x7fELF"
    for kind in ("predicate-filter-columns", "predicate-filter-records"):
        unit = cv.expr_source(kind, text, layout=layout, api=hip)
        assert "has_pred = true" in unit and "pred_mask(" in unit and text in unit and "pst_jit_filter" in unit
        assert "const PstV3<double> a0 = {" in unit  # Position3D cut out of the lane's words
        assert cv.jit_compile_source(unit, api=hip)[:4] == b"\x7fELF"
    # a layout whose points do not fit four to a lane has no such kernel: the predicate keeps its byte mask there
    wide = PointLayout.from_attributes([A.POSITION_3D, A.GPS_TIME, A.COLOR_RGB, A.WAVEFORM_PARAMETERS, A.NORMAL, A.POINT_ID, A.INTENSITY,
                                        PointAttributeDefinition("Extra", T.Vec3f64), PointAttributeDefinition("Extra2", T.Vec3f64)], api=hip)
    assert cv.expr_source("predicate-filter-columns", "Intensity > 3", layout=wide, api=hip) == ""


def test_expression_text_must_nest_and_predicates_cannot_name_reserved_attributes(hip):
    """Round-5 advisor findings: `x) , (y` would escape the cast the text is pasted into; an attribute called `i` / `p0` / `int` would end as a
    duplicate-parameter or keyword error inside generated code.  Both are PST_ERR_UNSUPPORTED_TRANSFORM with a message that says so."""
    for bad in ("x) , (y", "(v + 1", "v + 1)", "p0[i) + (1]", "((v)"):
        with pytest.raises(PastureError) as e:
            cv.expr_source("transform", bad, src_datatype=T.Vec3f64, dst_datatype=T.Vec3f64, api=hip)
        assert e.value.code == 7 and ("close" in str(e.value) or "open" in str(e.value)), (bad, str(e.value))
    for name in ("i", "p0", "int", "double", "return"):
        layout = PointLayout.from_attributes([A.POSITION_3D, PointAttributeDefinition(name, T.U16)], api=hip)
        with pytest.raises(PastureError) as e:
            cv.expr_source("predicate", f"{name} > 3", layout=layout, api=hip)
        assert e.value.code == 7 and "reserve" in str(e.value), (name, str(e.value))
    layout = PointLayout.from_attributes([A.POSITION_3D, PointAttributeDefinition("sqrt", T.F64)], api=hip)
    with pytest.raises(PastureError) as e:
        cv.expr_source("predicate", "sqrt (Position3D.x) > 2.0", layout=layout, api=hip)
    assert e.value.code == 7 and "used as a function" in str(e.value)
    assert "const double sqrt" in cv.expr_source("predicate", "sqrt > 2.0", layout=layout, api=hip)  # (as a value it is an attribute like any other)


def test_twin_semantics_are_rust_as():
    """The g++ twin itself against numpy on the documented rules: the result is converted to T with Rust `as`."""
    f = expr_twin.map_twin("u16", "u8", 1, True, "v * 2.0 + 0.5")  # T = u16: 400.5 -> 400; then u16 -> u8 truncates: 144
    assert f(np.array([0, 1, 200, 40000], dtype=np.uint16)).tolist() == [0, 2, 144, 255 & 65535 & 0xFF]
    g = expr_twin.map_twin("f64", "i8", 1, False, "v * 1000.0")  # T = i8 (post): f64 -> i8 saturates first, then the product saturates again
    assert g(np.array([0.5, -3.0, 1e9, np.nan])).tolist() == [0, -128, 127, 0]
    h = expr_twin.map_twin("f64", "f64", 3, False, "x + y + z ; (double)i ; (double)c")
    assert h(np.arange(6.0).reshape(2, 3), first=10).tolist() == [[3.0, 10.0, 2.0], [12.0, 11.0, 2.0]]


# ---- GPU suite -----------------------------------------------------------------------------------------------------------------------------
def _device_array(values):
    import torch
    t = torch.as_tensor(np.ascontiguousarray(values, dtype=np.float64), device="cuda")
    return t, t.data_ptr()


@pytest.mark.gpu
@pytest.mark.parametrize("src_kind,dst_kind", PAIRINGS)
@pytest.mark.parametrize("apply_to_source", [False, True])
def test_reference_plus_42_scenarios_through_the_expression_path(hip, src_kind, dst_kind, apply_to_source):
    """buffer_converter_transformed_{target,source}_attribute_generic, buffer_conversion.rs:764-846: CustomPointTypeBig -> [POSITION_3D] with
    `|p| p.add_scalar(42.0)`, for the reference's four buffer pairings; expected = the closure applied to the source positions."""
    big = custom_point_type_big(hip)
    rec = random_records(big, 16, 5)
    src = make_buffer(src_kind, big, rec)
    custom = PointLayout.from_attributes([A.POSITION_3D], api=hip)
    conv = BufferLayoutConverter.for_layouts_with_default(big, custom)
    conv.set_custom_mapping_with_expression(A.POSITION_3D, A.POSITION_3D, "v + 42.0", apply_to_source)
    out = conv.convert(src, BUFFER_KINDS[dst_kind])
    assert out.point_layout() == custom
    assert np.array_equal(out.view_attribute(A.POSITION_3D), rec["Position3D"] + 42.0)


@pytest.mark.gpu
@pytest.mark.parametrize("src_kind,dst_kind", [("V", "V"), ("V", "H"), ("H", "V")])
@pytest.mark.parametrize("apply_to_source", [False, True])
def test_plus_42_at_scale_runs_as_one_fused_kernel(hip, src_kind, dst_kind, apply_to_source):
    """The same scenario on 2^18 points (whole tiles, aligned records): pst_last_plan_kinds names ONE family -- the plan-specialised kernel with the
    closure inside --, the positions are the source's + 42.0, every other attribute of a wider target arrives unchanged; with a ragged point count the
    tail (less than one tile) goes through the expression's strided kernel and the interpreter, same values; PST_EXPR_FUSE is the A/B switch."""
    big = custom_point_type_big(hip)
    target = PointLayout.from_attributes_packed([A.POSITION_3D, A.GPS_TIME, A.CLASSIFICATION], 1, api=hip)
    for n in (1 << 18, (1 << 18) + 77):
        rec = random_records(big, n, 5)
        src = make_buffer(src_kind, big, rec)
        conv = BufferLayoutConverter.for_layouts_with_default(big, target)
        conv.set_custom_mapping_with_expression(A.POSITION_3D, A.POSITION_3D, "v + 42.0 + (double)(i & 1)", apply_to_source)
        dst = BUFFER_KINDS[dst_kind].new_from_layout(target)
        dst.resize(n)
        conv.convert_into(src, dst)
        kinds = cv.last_plan_kinds(hip)
        assert kinds == (["jit"] if n % 256 == 0 else ["interpreted", "jit", "expression"]), kinds  # ("expression": the closure's own strided pass over the tail)
        want = rec["Position3D"] + 42.0 + (np.arange(n) & 1)[:, None]
        assert np.array_equal(dst.view_attribute(A.POSITION_3D), want)
        assert np.array_equal(dst.view_attribute(A.GPS_TIME), rec["GpsTime"]) and np.array_equal(dst.view_attribute(A.CLASSIFICATION), rec["Classification"])
        # a sub-range that starts mid-buffer: the expression's index is the point's index in the SOURCE buffer
        part = BUFFER_KINDS[dst_kind].new_from_layout(target)
        part.resize(4096)
        conv.convert_into_range(src, range(5001, 5001 + 4096), part, range(0, 4096))
        assert np.array_equal(part.view_attribute(A.POSITION_3D), want[5001:5001 + 4096])


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["V", "H"])
def test_reference_transform_attribute_by_index(hip, kind):
    """test_transform_attribute_generic, point_buffer.rs:2007-2043: positions overwritten with `overwrite_data[index].position` -- the closure
    captures an array and uses its index argument: p0[3 * i + c]."""
    big = custom_point_type_big(hip)
    test_data, overwrite = random_records(big, 16, 1), random_records(big, 16, 2)
    buf = make_buffer(kind, big, test_data)
    keep, ptr = _device_array(overwrite["Position3D"])
    transform_attribute_expr(buf, A.POSITION_3D, "p0[3 * i + c]", [ptr])
    assert np.array_equal(buf.view_attribute(A.POSITION_3D), overwrite["Position3D"])
    for a in big.attributes():  # nothing else moved
        if a.name() != "Position3D":
            assert np.array_equal(buf.view_attribute(a.attribute_definition()), test_data[a.name()])
    del keep


def _bits(a):
    """The bytes two results are compared by: bit for bit, except that every NaN counts as THE NaN.  Which NaN an operation returns (sign, payload)
    is unspecified in Rust (the reference's closures: "the bit pattern of a NaN result is non-deterministic") and differs between this GPU and the
    twin's x86 host -- `y - z` with a NaN z is y + (-z) on gfx950 and comes back with the sign flipped, SSE's subss keeps the operand's sign."""
    a = np.ascontiguousarray(a)
    if a.dtype.kind == "f":
        a = np.where(np.isnan(a), np.array(np.nan, dtype=a.dtype), a)
    return np.ascontiguousarray(a).tobytes()


def _values(ct, n, ncomp, rng):
    npdt = expr_twin._NP[ct]
    shape = (n, ncomp) if ncomp > 1 else (n,)
    if npdt in (np.float32, np.float64):
        v = (rng.random(shape) - 0.3) * 1000.0
        v.reshape(-1)[:: 97] = np.nan  # NaN -> 0 and friends
        v.reshape(-1)[1:: 89] = 1e300 if npdt is np.float64 else 3e38
        return v.astype(npdt)
    info = np.iinfo(npdt)
    return rng.integers(info.min, info.max, size=shape, dtype=npdt, endpoint=True)


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(EXPRS)))
@pytest.mark.parametrize("kind", ["V", "H"])
def test_transform_attribute_expressions_against_the_gxx_twin(hip, case, kind):
    """Every expression text on every scalar / Vec3 datatype it applies to, in place on interleaved (packed: odd offsets) and columnar
    buffers, 5 003 points (ragged against every block size), parameters and the point index included: bit for bit against the g++ twin."""
    text, ints, floats = EXPRS[case]
    rng = np.random.default_rng(100 + case)
    n = 5003
    p0, p1 = rng.random(13) * 10.0, rng.random(3) - 0.5
    k0, ptr0 = _device_array(p0)
    k1, ptr1 = _device_array(p1)
    for name, dt, ncomp in [(k, v, 1) for k, v in SCALARS.items()] + [(k, v, 3) for k, v in VEC3.items()]:
        is_float = name in ("f32", "f64")
        if (is_float and not floats) or (not is_float and not ints) or (";" in text and ncomp == 1):
            continue
        attr = PointAttributeDefinition("Value", dt)
        layout = PointLayout.from_attributes_packed([A.CLASSIFICATION, attr, A.INTENSITY], 1, api=hip)
        vals = _values(name, n, ncomp, rng)
        buf = BUFFER_KINDS[kind].new_from_layout(layout)
        buf.resize(n)
        buf.set_attribute_range(attr, range(0, n), vals)
        transform_attribute_expr(buf, attr, text, [ptr0, ptr1])
        want = expr_twin.map_twin(name, name, ncomp, False, text)(vals, 0, [p0, p1])
        got = buf.view_attribute(attr)
        assert _bits(got) == _bits(want), f"{text!r} on {'Vec3' if ncomp == 3 else ''}{name}: {np.flatnonzero((got != want).reshape(n, -1).any(axis=1))[:5]}"
        # a slice_mut sees its OWN indices (the closure's index is the index in the buffer the call is made on)
        view = buf.slice(range(1000, 1100))
        before = view.view_attribute(attr)
        transform_attribute_expr(view, attr, text, [ptr0, ptr1])
        assert _bits(view.view_attribute(attr)) == _bits(expr_twin.map_twin(name, name, ncomp, False, text)(before, 0, [p0, p1]))
    del k0, k1


CONVERSIONS = [("u8", "f64", 1), ("f64", "u8", 1), ("i32", "f32", 3), ("f64", "f32", 3), ("u16", "i32", 3), ("f32", "i16", 1), ("i64", "u16", 1), ("f64", "f64", 3), ("u32", "u32", 1)]


@pytest.mark.gpu
@pytest.mark.parametrize("src_kind,dst_kind", PAIRINGS)
@pytest.mark.parametrize("apply_to_source", [False, True])
def test_mapping_expressions_with_conversion_against_the_gxx_twin(hip, src_kind, dst_kind, apply_to_source):
    """set_custom_mapping_with_expression between DIFFERENT datatypes: T is the source datatype before the `as` conversion (apply_to_source) or
    the target datatype after it; four storage pairings, packed layouts with other attributes around the mapped one, a sub-range conversion
    (the index the expression sees is the point's index in the source buffer)."""
    rng = np.random.default_rng(7 + apply_to_source)
    n = 4099
    for s_ct, d_ct, ncomp in CONVERSIONS:
        s_dt = (VEC3 if ncomp == 3 else SCALARS)[s_ct]
        d_dt = (VEC3 if ncomp == 3 else SCALARS)[d_ct]
        is_float_T = (s_ct if apply_to_source else d_ct) in ("f32", "f64")
        for text, ints, floats in EXPRS[:6] + EXPRS[7:8]:
            if (is_float_T and not floats) or (not is_float_T and not ints):
                continue
            sa, da = PointAttributeDefinition("Value", s_dt), PointAttributeDefinition("Value", d_dt)
            sl = PointLayout.from_attributes_packed([A.CLASSIFICATION, sa, A.GPS_TIME], 1, api=hip)
            dl = PointLayout.from_attributes_packed([A.GPS_TIME, da, A.CLASSIFICATION], 1, api=hip)
            vals = _values(s_ct, n, ncomp, rng)
            src = BUFFER_KINDS[src_kind].new_from_layout(sl)
            src.resize(n)
            src.set_attribute_range(sa, range(0, n), vals)
            gps = rng.random(n)
            src.set_attribute_range(A.GPS_TIME, range(0, n), gps)
            conv = BufferLayoutConverter.for_layouts(sl, dl)
            conv.set_custom_mapping_with_expression(sa, da, text, apply_to_source)
            out = conv.convert(src, BUFFER_KINDS[dst_kind])
            twin = expr_twin.map_twin(s_ct, d_ct, ncomp, apply_to_source, text)
            assert _bits(out.view_attribute(da)) == _bits(twin(vals, 0)), (text, s_ct, d_ct, apply_to_source)
            assert np.array_equal(out.view_attribute(A.GPS_TIME), gps)  # the other mappings went through the regular plan
            part = BUFFER_KINDS[dst_kind].new_from_layout(dl)
            part.resize(500)
            conv.convert_into_range(src, range(1234, 1734), part, range(0, 500))
            assert _bits(part.view_attribute(da)) == _bits(twin(vals[1234:1734], 1234))


def _ct_of(dt):
    for table, nc in ((SCALARS, 1), (VEC3, 3)):
        for name, v in table.items():
            if v == dt:
                return name, nc
    return None, 0


def _same_values(a, b):
    """Two results of the SAME library on the same input (the plan with and without the expression mappings): equal bit for bit outside NaNs."""
    if a.dtype.kind == "f":
        return np.array_equal(np.isnan(a), np.isnan(b)) and _bits(a) == _bits(b)
    return a.tobytes() == b.tobytes()


FUZZ = int(os.environ.get("PST_FUZZ_SCALE", "1"))


def random_expression_conversion(hip, seed):
    """One case of the fuzz below; False when the random layouts offer no attribute an expression could map."""
    import test_jit as tj
    sl, tl, plain, rng = tj.random_converter(hip, 52000 + seed)
    _, _, fused, _ = tj.random_converter(hip, 52000 + seed)  # the same converter again: it gets the expression mappings on top
    srcs = [a.attribute_definition() for a in sl.attributes()]
    tgts = [a.attribute_definition() for a in tl.attributes()]
    picks = {}
    for ti in rng.permutation(len(tgts))[: int(rng.integers(1, 4))]:
        t = tgts[ti]
        d_ct, nc = _ct_of(t.datatype())
        cands = [a for a in srcs if d_ct is not None and _ct_of(a.datatype())[1] == nc]
        if not cands:
            continue
        a = cands[rng.integers(0, len(cands))]
        s_ct = _ct_of(a.datatype())[0]
        on_source = bool(rng.random() < 0.5)
        is_float = (s_ct if on_source else d_ct) in ("f32", "f64")
        texts = [e[0] for e in EXPRS[:6] + EXPRS[7:] if (e[2] if is_float else e[1]) and (";" not in e[0] or nc == 3)]
        text = texts[rng.integers(0, len(texts))]
        fused.set_custom_mapping_with_expression(a, t, text, on_source)
        picks[t.name()] = (a, s_ct, d_ct, nc, on_source, text)
    if not picks:
        return False
    n = int(rng.choice([int(rng.integers(1, 3000)), int(rng.integers(3000, 70_000)), 1 << 16, (1 << 16) + int(rng.integers(1, 300))]))
    rec = tj.random_source_records(sl, n, rng)
    sk, dk = PAIRINGS[rng.integers(0, 4)]
    src = BUFFER_KINDS[sk].from_numpy(rec, sl)

    def check(got, want, first, lo, hi, d0):
        for t in tgts:
            g, w = got.view_attribute(t), want.view_attribute(t)
            if t.name() in picks:
                a, s_ct, d_ct, nc, on_source, text = picks[t.name()]
                twin = expr_twin.map_twin(s_ct, d_ct, nc, on_source, text)(rec[a.name()][lo:hi], first)
                assert _bits(g[d0:d0 + (hi - lo)]) == _bits(twin), (seed, t.name(), text, s_ct, d_ct, on_source, sk, dk, n, lo, hi)
                assert _same_values(g[:d0], w[:d0]) and _same_values(g[d0 + hi - lo:], w[d0 + hi - lo:]), (seed, t.name(), "outside the target range")
            else:
                assert _same_values(g, w), (seed, t.name(), "not an expression's target", sk, dk, n, lo, hi)

    out_fused = fused.convert(src, BUFFER_KINDS[dk])
    random_expression_conversion.plans.append((sk + dk, n, tuple(cv.last_plan_kinds(hip))))
    check(out_fused, plain.convert(src, BUFFER_KINDS[dk]), 0, 0, n, 0)
    # a sub-range: starts on a 16-point multiple (the specialised kernels' precondition for interleaved sides) or anywhere; the expression's index
    # is the point's index in the SOURCE buffer
    m = int(rng.integers(1, n + 1))
    lo = int(rng.integers(0, n - m + 1))
    d0 = int(rng.integers(0, 40))
    if rng.random() < 0.6:
        lo, d0 = lo - lo % 16, d0 - d0 % 16
    outs = []
    for conv in (fused, plain):
        dst = BUFFER_KINDS[dk].new_from_layout(tl)
        dst.resize(d0 + m + int(rng.integers(0, 20)) if conv is fused else outs[0].len())
        conv.convert_into_range(src, range(lo, lo + m), dst, range(d0, d0 + m))
        outs.append(dst)
    check(outs[0], outs[1], lo, lo, lo + m, d0)
    return True


random_expression_conversion.plans = []


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(24 * FUZZ))
def test_random_layouts_with_expression_mappings_against_the_twin_and_the_plain_plan(hip, seed):
    """Differential fuzz of round 6's fused path: test_jit's random converters (every datatype, packed / repr(C) layouts, unmapped target attributes,
    affine and bit-field mappings) with one to three EXPRESSION mappings on top, random storage pairing, point count and sub-range.  The expressions'
    targets equal the g++ twin bit for bit; every other target attribute -- same kernel when the plan is fused, the expression's entry beside it --
    equals what the converter WITHOUT the expression mappings writes (that one is pinned against the oracle by test_jit / test_gpu_parity)."""
    random_expression_conversion(hip, seed)


@pytest.mark.gpu
def test_expression_fuzz_ran_on_fused_kernels(hip):
    """Of the cases above with an interleaved side and at least one whole tile of points, a good share must have run as the plan-specialised kernel
    with the expressions in it (`jit` among pst_last_plan_kinds); columns -> columns never does (one launch per mapping is that pairing's form)."""
    plans = random_expression_conversion.plans
    if not plans:
        pytest.skip("runs after the fuzz cases")
    eligible = [p for p in plans if p[0] != "HH" and p[1] >= 4096]
    taken = [p for p in eligible if "jit" in p[2]]
    print(f"expression fuzz: {len(plans)} cases, {len(eligible)} with an interleaved side and >= 4096 points, {len(taken)} of them fused")
    assert all("jit" not in p[2] and "expression" in p[2] for p in plans if p[0] == "HH"), [p for p in plans if p[0] == "HH"][:6]
    assert len(taken) >= len(eligible) // 3, (len(taken), len(eligible), eligible[:6])


PREDICATES = [
    "Classification == 2 && Position3D.z < 50.0",
    "(Intensity & 1) == 0 || GpsTime > 0.75",
    "Position3D.x * Position3D.x + Position3D.y * Position3D.y < 250000.0",
    "i % 3 == 0",
    "p0[Classification] > 0.5 && ColorRGB.x >= ColorRGB.z",
    "true",
    "ReturnNumber > NumberOfReturns",
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(PREDICATES)))
@pytest.mark.parametrize("out_kind", ["V", "H"])
def test_filter_expressions_against_the_gxx_twin_and_the_mask_path(hip, case, out_kind):
    """HashMapBuffer::filter with the predicate as an expression over attribute names: the selected points equal (a) numpy selection with the
    g++ twin's mask and (b) the library's own mask-based filter given that mask, attribute by attribute."""
    text = PREDICATES[case]
    layout = las.point_layout_from_las_point_format(las.Format(3), False, api=hip)  # typed LAS-3: Position3D, flags, GpsTime, ColorRGB, ...
    n = 20_011
    rec = random_records(layout, n, 40 + case)
    rec["Position3D"] = np.random.default_rng(case).random((n, 3)) * np.array([1000.0, 1000.0, 100.0])
    rec["ReturnNumber"] %= 8
    rec["NumberOfReturns"] %= 8
    src = make_buffer("H", layout, rec)
    p0 = np.random.default_rng(9).random(256)
    keep, ptr = _device_array(p0)
    out = src.filter_expr(BUFFER_KINDS[out_kind], text, [ptr])
    assert cv.last_plan_kinds(hip) == ["jit"], cv.last_plan_kinds(hip)  # the streaming compaction kernel with the predicate inside (the ragged last tile: the gather kernel)
    attrs = [(a.name(), {v: k for k, v in SCALARS.items()}.get(a.datatype()) or {v: k for k, v in VEC3.items()}.get(a.datatype()), a.datatype().num_components())
             for a in layout.attributes() if a.name() in text]
    mask = expr_twin.pred_twin(attrs, text)({name: rec[name] for name, _, _ in attrs}, n, 0, [p0])
    assert out.len() == int(mask.sum()) and 0 < out.len() <= n
    ref = src.filter(BUFFER_KINDS[out_kind], mask)
    for a in layout.attributes():
        d = a.attribute_definition()
        assert np.array_equal(out.view_attribute(d), rec[a.name()][mask.astype(bool)])
        assert out.view_attribute(d).tobytes() == ref.view_attribute(d).tobytes()
    del keep


def _random_term(rng, name, ct, nc):
    comp = name + "." + "xyz"[rng.integers(0, 3)] if nc == 3 else name
    if ct in ("f32", "f64"):
        c = float(np.round(rng.uniform(-8000, 8000), 3))
        return [f"{comp} > {c}", f"{comp} * 0.5 < {c}", f"{comp} == {comp}", f"fabs((double){comp}) >= {abs(c)}"][rng.integers(0, 4)]
    info = np.iinfo(expr_twin._NP[ct])
    c = int(rng.integers(max(info.min, -(1 << 62)), min(info.max, 1 << 62), endpoint=True))
    lit = f"{c}ull" if ct == "u64" else f"{c}ll" if ct == "i64" else str(c)
    terms = [f"{comp} > {lit}", f"({comp} & {int(rng.integers(1, 8))}) == {int(rng.integers(0, 2))}", f"{comp} % 3 != 1", f"{comp} <= {lit}"]
    if nc == 3:
        terms.append(f"{name}.x < {name}.y")
    return terms[rng.integers(0, len(terms))]


def random_predicate_filter(hip, seed):
    """One case of the fuzz below."""
    import re
    import test_jit as tj
    rng = np.random.default_rng(61000 + seed)
    ALL = list(SCALARS.values()) + list(VEC3.values()) + [T.Vec4u8, T.ByteArray(5), T.ByteArray(16)]
    limit = 64 if rng.random() < 0.8 else 120  # beyond 64 bytes per point: no streaming form, the mask route
    attrs, total = [], 0
    for k in range(int(rng.integers(1, 13))):
        t = ALL[rng.integers(0, len(ALL))]
        if total + t.size() > limit:
            continue
        attrs.append(PointAttributeDefinition(f"a{k}", t))
        total += t.size()
    if not attrs:
        attrs = [PointAttributeDefinition("a0", T.U16)]
    layout = PointLayout.from_attributes(attrs, api=hip) if rng.integers(0, 3) == 0 else PointLayout.from_attributes_packed(attrs, 1, api=hip)
    n = int(rng.choice([1, 2047, 2048, 2049, 10_000, 33_333, 70_001]))
    rec = tj.random_source_records(layout, n, rng)
    src = HashMapBuffer.from_numpy(rec, layout)
    typed = [(a.name(),) + _ct_of(a.datatype()) for a in attrs if _ct_of(a.datatype())[0]]
    terms = []
    for _ in range(int(rng.integers(1, 4))):
        r = rng.random()
        if typed and r < 0.75:
            terms.append(_random_term(rng, *typed[rng.integers(0, len(typed))]))
        elif r < 0.9:
            terms.append(f"i % {int(rng.integers(2, 6))} == 0")
        else:
            terms.append("p0[i % 256] > 0.4")
    text = terms[0]
    for t in terms[1:]:
        text = f"({text}) {['&&', '||'][rng.integers(0, 2)]} {'!' if rng.random() < 0.2 else ''}({t})"
    p0 = rng.random(256)
    keep, ptr = _device_array(p0)
    kind = "VH"[rng.integers(0, 2)]
    out = src.filter_expr(BUFFER_KINDS[kind], text, [ptr])
    used = [(name, ct, nc) for name, ct, nc in typed if re.search(rf"\b{name}\b", text)]
    mask = expr_twin.pred_twin(used, text)({name: rec[name] for name, _, _ in used}, n, 0, [p0]).astype(bool)
    assert out.len() == int(mask.sum()), (seed, text, out.len(), int(mask.sum()))
    for a in attrs:
        got = out.view_attribute(a)
        assert np.ascontiguousarray(got).tobytes() == np.ascontiguousarray(rec[a.name()][mask]).tobytes(), (seed, text, a.name(), kind, n)
    del keep
    return True


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(24 * FUZZ))
def test_random_layouts_with_random_predicates_against_the_twin(hip, seed):
    """Differential fuzz of the predicate inside the compaction kernels (count pass + scatter pass of the layout's run-time compiled plan): random
    columnar layouts (every datatype, packed / repr(C), some beyond the 64 bytes the streaming form takes), random predicates over their attributes,
    the point index and a parameter array, point counts around the 2048-point tile; the selected points are numpy's selection with the g++ twin's
    mask, byte for byte, in both target kinds."""
    random_predicate_filter(hip, seed)


@pytest.mark.gpu
def test_expression_errors_at_run_time(hip):
    layout = PointLayout.from_attributes([A.POSITION_3D, A.CLASSIFICATION], api=hip)
    buf = HashMapBuffer.new_from_layout(layout)
    buf.resize(100)
    with pytest.raises(PastureError) as e:  # a text that does not compile: the compiler's log is the message
        transform_attribute_expr(buf, A.POSITION_3D, "v +* nothing")
    assert e.value.code == 7 and "error" in str(e.value)
    with pytest.raises(PastureError) as e:
        buf.filter_expr(HashMapBuffer, "NoSuchAttribute > 1")
    assert e.value.code == 7 and "NoSuchAttribute" in str(e.value)
    with pytest.raises(PastureError) as e:  # the attribute lookup keeps the reference's panic
        transform_attribute_expr(buf, A.INTENSITY, "v")
    assert e.value.code == 4
    vec = VectorBuffer.new_from_layout(layout)
    with pytest.raises(PastureError) as e:  # filter is defined on HashMapBuffer (point_buffer.rs:1064)
        vec.filter_expr(HashMapBuffer, "Classification == 1")
    assert e.value.code == 1
    empty = HashMapBuffer.new_from_layout(layout)
    assert empty.filter_expr(VectorBuffer, "Classification == 1").len() == 0
    transform_attribute_expr(empty, A.POSITION_3D, "v + 1.0")  # no points: still compiles the text, launches nothing
