"""Plan-specialised conversion kernels (pasture_amd/csrc/jit.cpp, jit_quad.hpp): the generator and the device headers under hipRTC on the
CPU (no GPU needed: hipRTC cross-compiles for gfx950), and on the GPU the specialised kernels against the oracle and against the interpreted
kernels.  The reference's converter is layout-generic (buffer_conversion.rs:112-234); whichever kernel family serves a plan, the target
bytes must be the oracle's."""
import os
import re
import subprocess
import sys
import time

import numpy as np
import pytest

from harness import BUFFER_KINDS
from pasture_amd import conversion as cv
from pasture_amd import las
from pasture_amd.buffers import HashMapBuffer, VectorBuffer
from pasture_amd.conversion import BufferLayoutConverter, Transform
from pasture_amd.layout import PointAttributeDataType as T, PointAttributeDefinition, PointLayout, attributes as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FUZZ = int(os.environ.get("PST_FUZZ_SCALE", "1"))
SC = [T.U8, T.I8, T.U16, T.I16, T.U32, T.I32, T.U64, T.I64, T.F32, T.F64]
V3 = [T.Vec3u8, T.Vec3u16, T.Vec3f32, T.Vec3i32, T.Vec3f64]
OPAQUE = [T.Vec4u8, T.ByteArray(5), T.ByteArray(16)]


def random_converter(api, seed, transforms=True):
    """Random source / target layouts (every datatype; packed or repr(C)), name-matched defaults with type changes, unmapped target
    attributes (read-modify-written records), custom mappings with affine / bit-field transformations on either side."""
    rng = np.random.default_rng(seed)
    n_src = int(rng.integers(1, 13))
    src_attrs = []
    for i in range(n_src):
        fam = rng.integers(0, 10)
        dt = SC[rng.integers(0, 10)] if fam < 6 else (V3[rng.integers(0, 5)] if fam < 9 else OPAQUE[rng.integers(0, 3)])
        src_attrs.append(PointAttributeDefinition(f"a{i}", dt))

    def other(dt):
        if dt in SC:
            return SC[rng.integers(0, 10)]
        if dt in V3:
            return V3[rng.integers(0, 5)]
        return dt
    order = rng.permutation(n_src)[: int(rng.integers(1, n_src + 1))]
    tgt_attrs = [PointAttributeDefinition(src_attrs[i].name(), other(src_attrs[i].datatype()) if rng.random() < 0.5 else src_attrs[i].datatype())
                 for i in order]
    with_default = rng.random() < 0.4
    if with_default and rng.random() < 0.7:
        tgt_attrs.append(PointAttributeDefinition("only_in_target", SC[rng.integers(0, 10)]))

    def make_layout(attrs):
        if rng.integers(0, 3) == 0:
            return PointLayout.from_attributes(attrs, api=api)
        return PointLayout.from_attributes_packed(attrs, int([1, 2, 4][rng.integers(0, 3)]), api=api)
    sl, tl = make_layout(src_attrs), make_layout(tgt_attrs)
    conv = (BufferLayoutConverter.for_layouts_with_default if with_default else BufferLayoutConverter.for_layouts)(sl, tl)
    for _ in range(int(rng.integers(0, 4)) if transforms else 0):
        t = tgt_attrs[rng.integers(0, len(tgt_attrs))]
        cands = [a for a in src_attrs if (a.datatype() in SC and t.datatype() in SC) or (a.datatype() in V3 and t.datatype() in V3)
                 or a.datatype() == t.datatype()]
        if not cands:
            continue
        a = cands[rng.integers(0, len(cands))]
        if rng.random() < 0.3:
            conv.set_custom_mapping(a, t)
            continue
        on_source = bool(rng.random() < 0.5)
        xt = a.datatype() if on_source else t.datatype()
        if xt in (T.F64, T.F32):
            conv.set_custom_mapping_with_transformation(a, t, Transform.affine(xt, (float(rng.uniform(-3, 3)),) * 3, (float(rng.uniform(-50, 50)),) * 3), on_source)
        elif xt in (T.Vec3f64, T.Vec3f32):
            conv.set_custom_mapping_with_transformation(a, t, Transform.affine(xt, tuple(rng.uniform(-3, 3, 3)), tuple(rng.uniform(-50, 50, 3))), on_source)
        elif xt in (T.U8, T.U16, T.U32, T.U64):
            bits = 8 * xt.size()
            conv.set_custom_mapping_with_transformation(a, t, Transform.bitfield(xt, int(rng.integers(0, bits)), int(rng.integers(1, 1 << min(bits, 16)))), on_source)
        else:
            conv.set_custom_mapping(a, t)
    return sl, tl, conv, rng


def random_source_records(sl, n, rng):
    rec = np.zeros(n, dtype=sl.numpy_record_dtype())
    for a in sl.attributes():
        npdt = a.datatype().numpy_dtype()
        nc = a.datatype().num_components()
        shape = (n, nc) if nc > 1 else (n,)
        if npdt.kind == "f":
            v = rng.uniform(-1e4, 1e4, shape)
            special = rng.random(shape) < 0.02
            v = np.where(special, rng.choice([np.nan, np.inf, -np.inf, 0.0, -0.0, 3e38, -1e300 if npdt.itemsize == 8 else -3e38, 0.5]), v)
            rec[a.name()] = v.astype(npdt)
        elif npdt.kind in "iu":
            info = np.iinfo(npdt)
            rec[a.name()] = rng.integers(info.min, info.max, size=shape, dtype=npdt, endpoint=True)
        else:
            rec[a.name()] = rng.integers(0, 256, size=(n, a.size()), dtype=np.uint8).view(npdt).reshape(shape if nc > 1 else (n,))
    return rec


def assert_same_columns(h, o):
    for k in o:
        a, b = h[k], o[k]
        if a.dtype.kind == "f":  # NaN payload bits of f64 -> f32 are not pinned by Rust `as`
            assert np.array_equal(np.isnan(a), np.isnan(b)), k
            fin = ~np.isnan(b)
            assert np.array_equal(a[fin].view(f"u{a.dtype.itemsize}"), b[fin].view(f"u{b.dtype.itemsize}")), k
        else:
            assert np.array_equal(a, b), k


# ---- CPU: the generator and the headers under hipRTC ---------------------------------------------------------------------------------------
def _scratch_bytes(code: bytes) -> int:
    """.private_segment_fixed_size of the (only) kernel in a code object: msgpack int behind the key in the metadata note."""
    i = code.find(b".private_segment_fixed_size")
    assert i >= 0
    b = code[i + len(b".private_segment_fixed_size"):]
    if b[0] <= 0x7f:
        return b[0]
    return {0xcc: lambda: b[1], 0xcd: lambda: int.from_bytes(b[1:3], "big"), 0xce: lambda: int.from_bytes(b[1:5], "big")}[b[0]]()


def test_in_tree_plan_kernels_use_no_scratch_memory():
    """The plan-specialised kernels instantiated in the library (conversions: convert_static.hip, compactions: filter.hip) keep their register images
    in registers: .private_segment_fixed_size == 0 for every one of them, read from the metadata of the built library.  (A plan type whose
    constants the optimiser does not fold compiles, passes every parity test -- and runs a hundred times slower.)"""
    from pasture_amd import _capi
    data = open(_capi.LIB_PATH, "rb").read()
    key = b".private_segment_fixed_size"
    seen = {}
    for family in (b"filter_stream_static_kernel", b"filter_stream_static3_kernel", b"quad_convert_static_kernel"):
        for m in re.finditer(rb"\.name[\xa0-\xbf\xd9\xda].?.?(_Z[0-9A-Za-z_]*" + family + rb"[0-9A-Za-z_]*)", data):
            j = data.find(key, m.end())
            assert 0 <= j - m.end() <= 4, "metadata layout changed"
            seen[m.group(1)] = _scratch_bytes(data[j:j + len(key) + 8])
    assert sum(1 for k in seen if b"filter_stream_static_kernel" in k) >= 16 and sum(1 for k in seen if b"quad_convert_static_kernel" in k) >= 19, len(seen)
    bad = {k.decode(): v for k, v in seen.items() if v != 0}
    assert not bad, bad


def test_knn_box_kernel_with_the_one_pass_fit_uses_no_scratch_memory():
    """knn_tile2_kernel<K, ..., FIT = 1> (the instances a volume-filling cloud runs on): .private_segment_fixed_size == 0.  Round 5 found the
    cubic solver's polynomial literals hoisted out of the chunk loop into vector registers and spilled -- 104 bytes of scratch per lane, 5.8 GB of
    write-back per 10^8-point launch, scratch reloads inside the fit; with the coefficients in constant memory (normals_device.hpp
    kThirdAngleTable) the kernels hold everything in registers, which was worth 6 % of the call (10 % on a LiDAR-like sheet).  A change that brings
    a spill back shows up here, without a GPU."""
    from pasture_amd import _capi
    data = open(_capi.LIB_PATH, "rb").read()
    key = b".private_segment_fixed_size"
    seen = {}
    for m in re.finditer(rb"\.name[\xa0-\xbf\xd9\xda].?.?(_Z[0-9A-Za-z_]*knn_tile2_kernelILi(?:8|16)E[0-9A-Za-z_]*)", data):
        j = data.find(key, m.end())
        assert 0 <= j - m.end() <= 4, "metadata layout changed"
        seen[m.group(1).decode()] = _scratch_bytes(data[j:j + len(key) + 8])
    # template arguments: K, THREADS, CAP, P3LDS, BATCH, WPE, WITH_KNN, ROT, FIT -- the plain one-pass instances: no lists, unrotated, FIT = 1
    plain = {k: v for k, v in seen.items() if k.endswith("ELb0ELb0ELi1EEEvNS_9Tile2ArgsE")}
    assert len(plain) >= 6, sorted(seen)
    assert all(v == 0 for v in plain.values()), plain
    # round 6: the reference-order fit (FIT = 0, the instances a LiDAR-like sheet runs on) keeps eight neighbours in registers and gathers the other
    # eight twice instead of holding sixteen positions across both passes -- it had 48 bytes of scratch per lane (2.7 GB of write-back per launch)
    in_order = {k: v for k, v in seen.items() if k.endswith("ELb0ELb0ELi0EEEvNS_9Tile2ArgsE")}
    assert len(in_order) >= 6, sorted(seen)
    assert all(v == 0 for v in in_order.values()), in_order


@pytest.mark.parametrize("seed", range(10))
def test_generated_plans_compile_under_hiprtc(hip, seed):
    """Host logic only: random converters -> the translation unit jit.cpp would hand to hipRTC -> compiled for gfx950 against the embedded
    headers.  Every quad image must end up in registers: no scratch memory."""
    sl, tl, conv, _ = random_converter(hip, 7000 + seed)
    compiled = 0
    for st, dt in ((VectorBuffer, HashMapBuffer), (HashMapBuffer, VectorBuffer), (VectorBuffer, VectorBuffer)):
        for with_bounds in (False, True):
            if with_bounds and not any(a.name() == "Position3D" for a in tl.attributes()):
                continue
            src = conv.jit_source(st, dt, with_bounds)
            if not src:
                continue  # records too large for the register images, or a plan another family serves
            assert "pstq::quad_convert_body<PstJitPlan>" in src
            code = cv.jit_compile_source(src, api=hip)
            assert code[:4] == b"\x7fELF"
            assert _scratch_bytes(code) == 0, src
            compiled += 1
    if sl.size_of_point_entry() + tl.size_of_point_entry() <= 200:
        assert compiled >= 3


def test_known_plan_source_text(hip):
    """The plan of the reference's bench layouts (layout_conversion_bench.rs:15-39) as the generator writes it: three `as` casts, records -> columns."""
    src_l = PointLayout.from_attributes_packed([A.POSITION_3D, A.CLASSIFICATION, A.INTENSITY, A.GPS_TIME], 1, api=hip)
    dst_l = PointLayout.from_attributes_packed([A.GPS_TIME, A.POSITION_3D.with_custom_datatype(T.Vec3f32), A.CLASSIFICATION.with_custom_datatype(T.U32),
                                                A.INTENSITY.with_custom_datatype(T.U8)], 1, api=hip)
    conv = BufferLayoutConverter.for_layouts(src_l, dst_l)
    text = conv.jit_source(VectorBuffer, HashMapBuffer)
    assert "src_aos = true, dst_aos = false" in text and "src_stride = 35, dst_stride = 0" in text
    rows = re.findall(r"\{([0-9, ]+)\},", text)
    assert len(rows) == 4
    first = [int(x) for x in rows[0].split(",")]
    assert first[:8] == [27, 0, 8, 8, 1, 9, 9, 0]  # GpsTime F64 @27 -> column, plain copy
    second = [int(x) for x in rows[1].split(",")]
    assert second[:8] == [0, 0, 24, 12, 3, 9, 8, 1]  # Position3D Vec3f64 @0 -> Vec3f32, converted
    # an identity between equal packed layouts is a byte copy of the records, never a compiled plan
    same = BufferLayoutConverter.for_layouts(src_l, src_l)
    assert same.jit_source(VectorBuffer, VectorBuffer) == ""
    # the LAS reader's plan has its own format-specialised kernels
    raw = las.point_layout_from_las_point_format(las.Format(0), True, api=hip)
    typed = las.point_layout_from_las_point_format(las.Format(0), False, api=hip)
    assert las.get_default_las_converter(raw, typed, (1, 1, 1), (0, 0, 0)).jit_source(VectorBuffer, HashMapBuffer) == ""


def test_static_plans_match_the_generator(hip):
    """pasture_amd/csrc/static_plans.inc (the in-tree instantiations for the reference's bench layouts) is what the generator writes today,
    and those converters are routed to the in-tree kernels without the run-time compiler."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_static_plans as G
    with open(os.path.join(ROOT, "pasture_amd", "csrc", "static_plans.inc")) as f:
        assert f.read() == G.render(), "stale static_plans.inc: run python tools/gen_static_plans.py and rebuild"
    before = cv.jit_stats(hip)
    for name, conv, st, dt, *rest in G.bench_converters():
        assert conv.prepare(st, dt, bool(rest and rest[0])) == cv.PLAN_STATIC, name
    after = cv.jit_stats(hip)
    assert after["compiled"] == before["compiled"] and after["disk_hits"] == before["disk_hits"]


def test_bad_source_reports_the_compiler_log(hip):
    from pasture_amd._capi import PastureError
    with pytest.raises(PastureError) as e:
        cv.jit_compile_source('#include "jit_quad.hpp"\nthis is not C++\n', api=hip)
    assert "error" in str(e.value)


def _filter_plan_source(sizes, columns, offsets=None, stride=None):
    """The translation unit filter.hip writes for a compaction plan (stream_source): attribute sizes in layout order; packed record offsets unless
    `offsets` / `stride` describe records with padding (then only the attributes' bytes of the staged target span are replaced)."""
    offs, o = [], 0
    for s in sizes:
        offs.append(0 if columns else o)
        o += s
    total = sum(sizes)
    covered = offsets is None
    if not covered:
        offs = list(offsets)
    stride = 0 if columns else (stride or total)
    piece_of = lambda v: 8 if v % 8 == 0 else 4 if v % 4 == 0 else 2 if v % 2 == 0 else 1
    pieces = [piece_of(z) if covered else min(piece_of(z), piece_of(o_) if o_ else 8) for z, o_ in zip(sizes, offs)]
    cap = min(2048, (52 * 1024 // (stride or total)) // 16 * 16)
    return ('#include "filter_stream.hpp"\nstruct PstFilterPlan {\n'
            f'  static constexpr int n = {len(sizes)};\n  static constexpr bool dst_columns = {"true" if columns else "false"}, covered = {"true" if covered else "false"}, has_pred = false;\n'
            f'  static constexpr uint32_t dst_stride = {stride}, cap = {cap};\n'
            '  __host__ __device__ static constexpr uint32_t size(int k) {\n    constexpr uint32_t t[n] = {' + ", ".join(map(str, sizes)) + '};\n    return t[k];\n  }\n'
            '  __host__ __device__ static constexpr uint32_t dst_off(int k) {\n    constexpr uint32_t t[n] = {' + ", ".join(map(str, offs)) + '};\n    return t[k];\n  }\n'
            '  __host__ __device__ static constexpr uint32_t piece(int k) {\n    constexpr uint32_t t[n] = {' + ", ".join(map(str, pieces)) + '};\n    return t[k];\n  }\n};\n'
            'extern "C" __global__ __launch_bounds__(512) void pst_jit_filter(const pstf::FilterArgs a) {\n  pstf::filter_stream_body<PstFilterPlan>(a);\n}\n')


@pytest.mark.parametrize("columns", [True, False])
@pytest.mark.parametrize("sizes", [[24, 2, 1, 1, 1, 1, 1, 1, 1, 2, 8, 6], [8, 6, 24, 1, 2], [1], [3, 5, 12, 16, 7], [4] * 16, [24, 24, 16]])
def test_compaction_plan_compiles_with_hiprtc(hip, sizes, columns):
    """filter_stream.hpp under hipRTC (no device needed): the streaming compaction kernel for attribute lists of every granule class -- typed LAS-3
    points, the bench layout, odd sizes, sixteen dwords, 64 bytes per point -- compiles for gfx950 without scratch memory."""
    code = cv.jit_compile_source(_filter_plan_source(sizes, columns), api=hip)
    assert code[:4] == b"\x7fELF"
    assert _scratch_bytes(code) == 0, (sizes, columns)


def test_compaction_plan_for_the_largest_las_format_compiles_with_hiprtc(hip):
    """Typed LAS-10 points (83 bytes, 20 attributes) into records: the largest plan the record side takes (96 bytes per point)."""
    sizes = [a.size() for a in las.point_layout_from_las_point_format(las.Format(10), False, api=hip).attributes()]
    assert sum(sizes) == 83
    code = cv.jit_compile_source(_filter_plan_source(sizes, False), api=hip)
    assert code[:4] == b"\x7fELF"
    assert _scratch_bytes(code) == 0


def test_compaction_plan_for_padded_records_compiles_with_hiprtc(hip):
    """A repr(C) record {f64, u16, (2 bytes of padding), u32, u8, (7)} : 24 bytes, 15 of them written: the read-modify-write variant."""
    code = cv.jit_compile_source(_filter_plan_source([8, 2, 4, 1], False, offsets=[0, 8, 12, 16], stride=24), api=hip)
    assert code[:4] == b"\x7fELF"
    assert _scratch_bytes(code) == 0


# ---- GPU -----------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture
def jit_sync(hip):
    cv.jit_set_mode("sync", api=hip)
    yield
    cv.jit_set_mode("env", api=hip)


def _run_case(api, seed, n, aligned, kinds, jit):
    """One random conversion of n points between buffers of the given kinds; range starts either at 16-point multiples (the specialised
    kernels' precondition for interleaved sides) or anywhere."""
    sl, tl, conv, rng = random_converter(api, seed)
    rec = random_source_records(sl, n, rng)
    src = BUFFER_KINDS[kinds[0]].from_numpy(rec, sl)
    pad = 16 * int(rng.integers(0, 3)) if aligned else int(rng.integers(0, 3))
    dst = BUFFER_KINDS[kinds[1]].new_from_layout(tl)
    dst.resize(n + 2 * pad)
    a0 = (16 * int(rng.integers(0, max(1, n // 64)))) if aligned else int(rng.integers(0, max(1, n // 4)))
    a0 = min(a0, n)
    conv.convert_into_range(src, range(a0, n), dst, range(pad + a0, pad + n))
    plan = cv.last_plan_kinds(api) if jit else []
    return {a.name(): dst.view_attribute(a.attribute_definition()) for a in tl.attributes()}, plan


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(90 * FUZZ))
def test_specialised_conversions_vs_oracle(hip, oracle, jit_sync, seed):
    """Differential fuzz with every eligible plan compiled (PST_JIT=sync): byte-identical to the oracle, full tiles on the specialised
    kernel and the ragged tail on the interpreter."""
    rng = np.random.default_rng(99000 + seed)
    n = int(rng.choice([256, 257, 1000, 4099, 20_011, 70_001]))
    aligned = bool(rng.random() < 0.75)
    kinds = [("V", "H"), ("H", "V"), ("V", "V")][seed % 3]
    # failures of the run-time compiler DURING the fuzz cases (sync mode: a plan is compiled inside the call that needs it).  Counted per case: texts
    # that are MEANT not to compile (tests/test_expressions.py) are failures of the same counter, and pytest does not run the modules back to back
    f0 = cv.jit_stats(hip)["failures"]
    h, plan = _run_case(hip, 99000 + seed, n, aligned, kinds, True)
    test_specialised_conversions_vs_oracle.failures += cv.jit_stats(hip)["failures"] - f0
    o, _ = _run_case(oracle, 99000 + seed, n, aligned, kinds, False)
    assert_same_columns(h, o)
    test_specialised_conversions_vs_oracle.plans.append(tuple(plan))


test_specialised_conversions_vs_oracle.plans = []
test_specialised_conversions_vs_oracle.failures = 0


@pytest.mark.gpu
def test_specialised_kernels_were_taken(hip):
    """Of the fuzz cases above a good share must really have run on compiled plans (aligned ranges, records that fit the images)."""
    plans = test_specialised_conversions_vs_oracle.plans
    if not plans:
        pytest.skip("runs after test_specialised_conversions_vs_oracle")
    taken = sum(1 for p in plans if "jit" in p)
    assert taken >= len(plans) // 3, (taken, len(plans))
    st = cv.jit_stats(hip)
    assert test_specialised_conversions_vs_oracle.failures == 0 and st["compiled"] + st["disk_hits"] >= taken // 2


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(16 * FUZZ))
def test_specialised_compaction_vs_oracle(hip, oracle, jit_sync, seed):
    """Differential fuzz of filter / filter_into with the streaming compaction kernel of every eligible layout compiled at run time (filter_stream.hpp;
    PST_JIT=sync): random packed (every third: repr(C), padded) layouts of at most 64 bytes per point, both target kinds, full tiles on the compiled kernel and the ragged last tile
    on the gather kernel; a hint below the number of matches truncates inside a tile.  Byte-identical to the oracle."""
    rng = np.random.default_rng(77000 + seed)
    ALL = [T.U8, T.I8, T.U16, T.I16, T.U32, T.F32, T.U64, T.F64, T.Vec3u8, T.Vec3u16, T.Vec3f32, T.Vec3f64, T.Vec4u8, T.ByteArray(5), T.ByteArray(16), T.ByteArray(7)]
    attrs, total = [], 0
    for i in range(int(rng.integers(1, 13))):
        t = ALL[rng.integers(0, len(ALL))]
        if total + t.size() > 64:
            continue
        attrs.append(PointAttributeDefinition(f"a{i}", t))
        total += t.size()
    n = int(rng.choice([2048, 2049, 10_000, 33_333, 70_001]))
    density = float(rng.choice([0.03, 0.5, 0.5, 0.97, 1.0]))
    mask = rng.random(n) < density
    kind = "VH"[seed % 2]
    k = int(mask.sum())
    hint = k if rng.random() < 0.5 else max(0, k - int(rng.integers(1, 3000)))

    def run(api):
        layout = PointLayout.from_attributes(attrs, api=api) if seed % 3 == 2 else PointLayout.from_attributes_packed(attrs, 1, api=api)
        src = HashMapBuffer.new_from_layout(layout)
        src.resize(n)
        src.synth_fill(seed, 5)
        out = src.filter(BUFFER_KINDS[kind], mask)
        kinds = cv.last_plan_kinds(api) if api is hip else None
        short = BUFFER_KINDS[kind].new_from_layout(layout)
        short.resize(k)
        if hint == k:
            assert src.filter_into(short, mask, hint) == k
        return out.len(), out.get_point_range(range(0, out.len())).tobytes(), short.get_point_range(range(0, k)).tobytes() if hint == k else b"", kinds
    hn, hb, hs, kinds = run(hip)
    on, ob, os_, _ = run(oracle)
    assert hn == on == k and hb == ob and hs == os_
    if k and not (seed % 3 == 2 and kind == "V"):  # (padded record targets with wide attributes stay on the gather kernel: a measured rule)
        assert kinds and kinds[0] in ("jit", "static"), kinds


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["V", "H"])
@pytest.mark.parametrize("types", [[T.U8], [T.U16], [T.Vec3u8], [T.U8, T.U8], [T.U32], [T.ByteArray(5)], [T.ByteArray(16), T.ByteArray(16), T.ByteArray(16), T.ByteArray(16)],
                                   [T.ByteArray(16)] * 4 + [T.Vec3f64, T.U64]])
def test_specialised_compaction_of_tiny_and_largest_records(hip, oracle, jit_sync, types, kind):
    """Records of one, two, three bytes (shorter than the dword grid a record image is shifted to), of the 64-byte limit of the column side and of the
    96-byte limit of the record side, sparse and dense masks."""
    n = 70_001
    for density in (0.03, 0.97):
        mask = np.random.default_rng(len(types) + int(density * 100)).random(n) < density

        def run(api):
            layout = PointLayout.from_attributes_packed([PointAttributeDefinition(f"a{i}", t) for i, t in enumerate(types)], 1, api=api)
            src = HashMapBuffer.new_from_layout(layout)
            src.resize(n)
            src.synth_fill(3, 5)
            out = src.filter(BUFFER_KINDS[kind], mask)
            kinds = cv.last_plan_kinds(api) if api is hip else None  # (before get_point_range: reading columns as records is a conversion)
            return out.len(), out.get_point_range(range(0, out.len())).tobytes(), kinds
        hn, hb, kinds = run(hip)
        on, ob, _ = run(oracle)
        assert hn == on and hb == ob, (types, density)
        if kind == "V" or sum(t.size() for t in types) <= 64:  # (96 bytes per point into records, 64 into columns)
            assert kinds[0] in ("jit", "static"), kinds


@pytest.mark.gpu
@pytest.mark.parametrize("density", [0.4, 1.0])
def test_specialised_compaction_keeps_the_padding_of_repr_c_records(hip, jit_sync, density):
    """filter_into a VectorBuffer of a repr(C) layout (padding between and after the attributes) on the compiled streaming kernel: the target's
    padding bytes and the records beyond the matches keep what they held (the reference writes attribute bytes only, point_buffer.rs:1110-1131)."""
    from harness import random_records
    attrs = [PointAttributeDefinition("t", T.F32), PointAttributeDefinition("i", T.U16), PointAttributeDefinition("c", T.Vec3u8), PointAttributeDefinition("k", T.U8),
             PointAttributeDefinition("p", T.Vec3f32), PointAttributeDefinition("b", T.ByteArray(5)), PointAttributeDefinition("f", T.U8), PointAttributeDefinition("g", T.U16),
             PointAttributeDefinition("h", T.U8)]  # (narrow attributes: padded records take the streaming kernel below four bytes per attribute)
    layout = PointLayout.from_attributes(attrs, api=hip)
    assert layout.size_of_point_entry() > sum(a.size() for a in layout.attributes())  # (there IS padding)
    n = 70_001
    rec = random_records(layout, n, 9)
    mask = np.random.default_rng(3).random(n) < density
    k = int(mask.sum())
    src = HashMapBuffer.from_numpy(rec, layout)
    dst = VectorBuffer.new_from_layout(layout)
    dst.resize(k + 3)
    raw = np.full((k + 3, layout.size_of_point_entry()), 0xAB, np.uint8)
    dst.set_point_range(range(0, k + 3), raw.view(layout.numpy_record_dtype()).reshape(-1))
    assert src.filter_into(dst, mask) == k
    assert cv.last_plan_kinds(hip)[0] == "jit", cv.last_plan_kinds(hip)
    got = np.ascontiguousarray(dst.get_point_range(range(0, k + 3))).view(np.uint8).reshape(k + 3, -1)
    covered = np.zeros(layout.size_of_point_entry(), bool)
    for a in layout.attributes():
        covered[a.offset():a.offset() + a.size()] = True
    assert (got[:, ~covered] == 0xAB).all() and (got[k:] == 0xAB).all()
    exp = np.ascontiguousarray(rec[mask]).view(np.uint8).reshape(k, -1)
    assert np.array_equal(got[:k][:, covered], exp[:, covered])


@pytest.mark.gpu
def test_compaction_kernel_arrives_in_the_background(hip):
    """PST_JIT=async: a large filter of a layout without an in-tree kernel shows its plan to the compiler thread and runs on the gather kernels;
    once the code object is there the same call takes the streaming kernel -- with identical bytes."""
    cv.jit_set_mode("async", api=hip)
    try:
        n = (1 << 20) + 1234
        layout = PointLayout.from_attributes_packed([A.POSITION_3D, PointAttributeDefinition("odd", T.ByteArray(7)), A.INTENSITY, A.GPS_TIME, A.CLASSIFICATION], 1, api=hip)
        src = HashMapBuffer.new_from_layout(layout)
        src.resize(n)
        src.synth_fill(4321, 0)
        mask = np.random.default_rng(8).random(n) < 0.5
        first = src.filter(VectorBuffer, mask)
        seen = [cv.last_plan_kinds(hip)]
        deadline = time.time() + 60
        while time.time() < deadline:
            again = src.filter(VectorBuffer, mask)
            seen.append(cv.last_plan_kinds(hip))
            if "jit" in seen[-1]:
                break
            time.sleep(0.05)
        assert "jit" in seen[-1], seen[-5:]
        assert seen[0] == ["interpreted"] or "jit" in seen[0]  # (a warm disk cache may serve the very first call)
        k = int(mask.sum())
        assert first.len() == again.len() == k
        assert first.get_point_range(range(0, k)).tobytes() == again.get_point_range(range(0, k)).tobytes()
    finally:
        cv.jit_set_mode("env", api=hip)


@pytest.mark.gpu
@pytest.mark.parametrize("kinds", [("V", "H"), ("H", "V"), ("V", "V")])
def test_las_reader_plan_into_a_custom_layout_specialised(hip, oracle, jit_sync, kinds):
    """The reader's "different layout" case (raw_readers.rs:820-905) with the affine position mapping and the bit fields of
    get_default_las_converter, plus the fused AABB: specialised kernel == oracle, bounds == calculate_bounds of the result."""
    from pasture_amd.algorithms import calculate_bounds
    n = 300_011

    def run(api):
        raw = las.point_layout_from_las_point_format(las.Format(1), True, api=api)
        tgt = PointLayout.from_attributes_packed([A.INTENSITY.with_custom_datatype(T.U32), A.POSITION_3D, A.RETURN_NUMBER, A.GPS_TIME,
                                                  A.NUMBER_OF_RETURNS.with_custom_datatype(T.U16), A.CLASSIFICATION], 1, api=api)
        conv = las.get_default_las_converter(raw, tgt, (0.001, 0.01, 0.25), (500000.0, -5400000.0, 100.0))
        src_v = VectorBuffer.new_from_layout(raw)
        src_v.resize(n)
        src_v.synth_fill(11, 5)
        if kinds[0] == "H":
            src = BufferLayoutConverter.for_layouts(raw, raw).convert(src_v, HashMapBuffer)
        else:
            src = src_v
        dst = BUFFER_KINDS[kinds[1]].new_from_layout(tgt)
        dst.resize(n)
        if api is hip:
            bounds = conv.convert_into_with_bounds(src, dst)
            plan = cv.last_plan_kinds(api)
        else:
            conv.convert_into(src, dst)
            bounds, plan = calculate_bounds(dst), []
        return {a.name(): dst.view_attribute(a.attribute_definition()) for a in tgt.attributes()}, bounds, plan
    (h, hb, plan), (o, ob, _) = run(hip), run(oracle)
    assert "jit" in plan, plan
    assert_same_columns(h, o)
    assert hb == ob


@pytest.mark.gpu
def test_async_mode_interprets_first_then_takes_the_compiled_plan(hip, oracle):
    """PST_JIT=async: a large call shows the plan to the compiler thread and is interpreted; once the code object is there the same call takes
    it -- with identical bytes."""
    cv.jit_set_mode("async", api=hip)
    try:
        n = (1 << 20) + 77
        lay = PointLayout.from_attributes_packed([A.POSITION_3D, A.INTENSITY, PointAttributeDefinition("weird", T.ByteArray(5)), A.GPS_TIME], 1, api=hip)
        tgt = PointLayout.from_attributes_packed([A.GPS_TIME, PointAttributeDefinition("weird", T.ByteArray(5)), A.INTENSITY.with_custom_datatype(T.F32),
                                                  A.POSITION_3D.with_custom_datatype(T.Vec3f32)], 1, api=hip)
        src = VectorBuffer.new_from_layout(lay)
        src.resize(n)
        src.synth_fill(1234, 0)
        conv = BufferLayoutConverter.for_layouts(lay, tgt)
        first = conv.convert(src, VectorBuffer)
        seen = [cv.last_plan_kinds(hip)]
        deadline = time.time() + 60
        while time.time() < deadline:
            again = conv.convert(src, VectorBuffer)
            seen.append(cv.last_plan_kinds(hip))
            if "jit" in seen[-1]:
                break
            time.sleep(0.05)
        assert "jit" in seen[-1], seen[-5:]
        assert seen[0] == ["interpreted"] or "jit" in seen[0]  # (a warm disk cache may serve the very first call)
        a = first.get_point_range(range(0, n)).tobytes()
        b = again.get_point_range(range(0, n)).tobytes()
        assert a == b
    finally:
        cv.jit_set_mode("env", api=hip)


@pytest.mark.gpu
def test_disk_cache_serves_a_second_process(tmp_path):
    """Code objects are cached on disk per hash of source + headers + options: a second process loads instead of compiling."""
    prog = (
        "import sys; sys.path.insert(0, %r)\n"
        "import pasture_amd as pa\n"
        "from pasture_amd import conversion as cv\n"
        "from pasture_amd.layout import PointLayout, attributes as A, PointAttributeDataType as T\n"
        "a = PointLayout.from_attributes_packed([A.POSITION_3D, A.INTENSITY, A.CLASSIFICATION], 1)\n"
        "b = PointLayout.from_attributes_packed([A.CLASSIFICATION, A.POSITION_3D.with_custom_datatype(T.Vec3f32), A.INTENSITY], 1)\n"
        "c = pa.BufferLayoutConverter.for_layouts(a, b)\n"
        "print(c.prepare(pa.VectorBuffer, pa.HashMapBuffer), cv.jit_stats())\n" % ROOT)
    env = dict(os.environ, PST_JIT_CACHE_DIR=str(tmp_path / "jit"), PST_JIT="async")
    out1 = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=300)
    out2 = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=300)
    assert out1.returncode == 0 and out2.returncode == 0, out1.stderr + out2.stderr
    assert out1.stdout.startswith("2 ") and "'compiled': 1" in out1.stdout and "'disk_hits': 0" in out1.stdout, out1.stdout
    assert out2.stdout.startswith("2 ") and "'compiled': 0" in out2.stdout and "'disk_hits': 1" in out2.stdout, out2.stdout
    files = list((tmp_path / "jit").glob("*.pstco"))
    assert len(files) == 1
    # A cache file that is not what the library wrote under that name -- truncated, overwritten, or the intact file of ANOTHER plan copied over it -- is
    # recognised before the loader sees it (hipModuleLoadData takes a bare pointer) and compiled again.  (tools/exp_jit_cache_damage.py, end of round 6:
    # the truncated file failed the call, the other plan's image under this name faulted the GPU.)
    intact = files[0].read_bytes()
    other = prog.replace("[A.CLASSIFICATION, A.POSITION_3D.with_custom_datatype(T.Vec3f32), A.INTENSITY]", "[A.INTENSITY, A.CLASSIFICATION, A.POSITION_3D]")
    env2 = dict(env, PST_JIT_CACHE_DIR=str(tmp_path / "jit2"))
    assert subprocess.run([sys.executable, "-c", other], env=env2, capture_output=True, text=True, timeout=300).returncode == 0
    foreign = list((tmp_path / "jit2").glob("*.pstco"))[0].read_bytes()
    assert foreign != intact
    for what, data in (("truncated", intact[: len(intact) // 2]), ("header only", intact[:80]), ("flipped byte", intact[:200] + bytes([intact[200] ^ 0x40]) + intact[201:]),
                       ("another plan's file", foreign), ("empty", b""), ("bare code object", intact[80:])):
        files[0].write_bytes(data)
        out3 = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=300)
        assert out3.returncode == 0, (what, out3.stderr[-2000:])
        assert out3.stdout.startswith("2 ") and "'compiled': 1" in out3.stdout and "'disk_hits': 0" in out3.stdout and "'failures': 0" in out3.stdout, (what, out3.stdout)
        out4 = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=300)  # ... and written again: the next process loads it
        assert out4.returncode == 0 and "'compiled': 0" in out4.stdout and "'disk_hits': 1" in out4.stdout, (what, out4.stdout, out4.stderr[-1000:])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 3, 7])
def test_full_size_1e8_specialised_equals_interpreted(hip, seed):
    """10^8 points, random packed layouts with about a third of the datatypes changed, the three pairings with an interleaved side: the
    specialised kernel and the interpreted one leave byte-identical targets (compared on the device), and columns -> records -> columns
    through the specialised kernels is the identity on every mapped column."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import exp_jit_layouts as X
    n = 100_000_000
    sl, tl = X.random_layouts(seed)
    for pairing in ("VH", "HV", "VV"):
        ST, DT = BUFFER_KINDS[pairing[0]], BUFFER_KINDS[pairing[1]]
        src = ST.new_from_layout(sl)
        src.resize(n)
        src.synth_fill(4242 + seed, 0)
        conv = BufferLayoutConverter.for_layouts(sl, tl)
        outs = {}
        for mode in ("off", "sync"):
            cv.jit_set_mode(mode, api=hip)
            try:
                dst = DT.new_from_layout(tl)
                dst.resize(n)
                conv.convert_into(src, dst)
                outs[mode] = (dst, cv.last_plan_kinds(hip))
            finally:
                cv.jit_set_mode("env", api=hip)
        assert "jit" in outs["sync"][1] and "jit" not in outs["off"][1], (outs["sync"][1], outs["off"][1])
        for a, b in zip(X.raw_bytes(outs["off"][0]), X.raw_bytes(outs["sync"][0])):
            assert torch.equal(a, b), (seed, pairing)
        del outs, src, dst


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1023, 1024, 4099, 1_300_001])
@pytest.mark.parametrize("fmt", [0, 1, 6])
def test_in_place_transform_on_packed_records_takes_the_specialised_kernel(hip, oracle, jit_sync, fmt, n):
    """transform_attribute on a VectorBuffer without padding runs as a whole-record records -> records plan on the plan-specialised kernel
    (algorithms.cpp): same bytes as the oracle's in-place loop, every other attribute untouched, ragged tail included."""
    from pasture_amd import las
    from pasture_amd.algorithms import transform_attribute
    from pasture_amd.conversion import Transform
    from pasture_amd.layout import PointAttributeDataType as T, attributes as A
    xf_args = (T.Vec3f64, (1.0001, 0.9999, 2.0), (1.0, -2.0, 0.5))
    out = {}
    for name, api in (("hip", hip), ("oracle", oracle)):
        layout = las.point_layout_from_las_point_format(las.Format(fmt), False, api=api)
        buf = VectorBuffer.new_from_layout(layout)
        buf.resize(n)
        buf.synth_fill(5, 0)
        transform_attribute(buf, A.POSITION_3D, Transform.affine(*xf_args))
        if name == "hip":
            assert "jit" in cv.last_plan_kinds(hip) or "static" in cv.last_plan_kinds(hip), cv.last_plan_kinds(hip)
        out[name] = buf.get_point_range(range(0, n))
    assert np.array_equal(out["hip"], out["oracle"])


@pytest.mark.gpu
def test_family_autotune_measures_once_and_changes_no_byte(hip):
    """Round-4 review, item 6: two kernel families can serve interleaved LAS-shaped plans and trade places from box to box.  The converter
    measures them on the first conversion of >= 2^22 points (pst_converter_family_choice) and keeps the winner: the choice is one of the two,
    both timings are positive, the call after it takes the chosen family, smaller calls and plans with one family do not measure, and the
    converted bytes are the source's fields whichever family ran."""
    from pasture_amd import conversion as cv
    from pasture_amd.buffers import HashMapBuffer, VectorBuffer
    n = (1 << 22) + 5
    typed = las.point_layout_from_las_point_format(las.Format(0), False, api=hip)
    raw = las.point_layout_from_las_point_format(las.Format(0), True, api=hip)
    # (a) typed LAS-0 records -> 10 columns
    src = VectorBuffer.new_from_layout(typed)
    src.resize(n)
    src.synth_fill(5, 0)
    dst = HashMapBuffer.new_from_layout(typed)
    dst.resize(n)
    conv = BufferLayoutConverter.for_layouts(typed, typed)
    conv.prepare(VectorBuffer, HashMapBuffer)
    assert conv.family_choice(HashMapBuffer)[0] == -1
    conv.convert_into_range(src, range(0, 1000), dst, range(0, 1000))  # below 2^22 points: no measurement
    assert conv.family_choice(HashMapBuffer)[0] == -1
    conv.convert_into(src, dst)
    choice, ms = conv.family_choice(HashMapBuffer)
    assert choice in (0, 1) and ms[0] > 0 and ms[1] > 0, (choice, ms)
    kinds = cv.last_plan_kinds(hip)
    assert kinds == (["las-specialised"] if choice == 0 else ["static"]), (choice, kinds)
    m = 50_000
    for lo in (0, n - m):
        rec = src.get_point_range(range(lo, lo + m)).reshape(-1).view(typed.numpy_record_dtype())
        for a in typed.attributes():
            assert np.array_equal(dst.get_attribute_range(a.attribute_definition(), range(lo, lo + m)), rec[a.name()]), a.name()
    conv.convert_into(src, dst)
    assert conv.family_choice(HashMapBuffer)[0] == choice and cv.last_plan_kinds(hip) == kinds
    assert conv.family_choice(VectorBuffer)[0] == -1  # (per target storage)
    # (a') the same converter the other way round: 10 columns -> typed records (the LAS transposer against the plan-specialised kernel)
    conv.prepare(HashMapBuffer, VectorBuffer)
    back = VectorBuffer.new_from_layout(typed)
    back.resize(n)
    conv.convert_into(dst, back)
    assert conv.family_choice(VectorBuffer)[0] == -1  # records from RECORDS: another slot (round 6), untouched
    bchoice, bms = conv.family_choice(VectorBuffer, source_type=HashMapBuffer)
    assert bchoice in (0, 1) and bms[0] > 0 and bms[1] > 0, (bchoice, bms)
    assert cv.last_plan_kinds(hip) == (["las-specialised"] if bchoice == 0 else ["static"]) or (bchoice == 1 and cv.last_plan_kinds(hip) == ["jit"])
    for lo in (0, n - m):
        assert np.array_equal(back.get_point_range(range(lo, lo + m)), src.get_point_range(range(lo, lo + m)))
    # (b) raw LAS-0 records -> typed records: the decoder against the plan-specialised kernel
    rsrc = VectorBuffer.new_from_layout(raw)
    rsrc.resize(n)
    rsrc.synth_fill(6, 0)
    rdst = VectorBuffer.new_from_layout(typed)
    rdst.resize(n)
    rconv = las.get_default_las_converter(raw, typed, (0.001, 0.001, 0.001), (5.0, 6.0, 7.0))
    rconv.prepare(VectorBuffer, VectorBuffer)
    rconv.convert_into(rsrc, rdst)
    rchoice, rms = rconv.family_choice(VectorBuffer)
    assert rchoice in (0, 1) and rms[0] > 0 and rms[1] > 0, (rchoice, rms)
    want = VectorBuffer.new_from_layout(typed)
    want.resize(m)
    rconv.convert_into_range(rsrc, range(n - m, n), want, range(0, m))  # small call: not measured, default order of preference
    assert np.array_equal(rdst.get_point_range(range(n - m, n)), want.get_point_range(range(0, m)))
    # (c) a plan with one family only: raw records -> columns always takes the LAS decoder
    cdst = HashMapBuffer.new_from_layout(typed)
    cdst.resize(n)
    rconv.convert_into(rsrc, cdst)
    assert rconv.family_choice(HashMapBuffer)[0] == 2


@pytest.mark.gpu
def test_family_autotune_stays_out_of_a_graph_capture(hip):
    """The measurement waits on the host once; a stream that is being captured into a hipGraph must not be measured on (hipEventSynchronize inside a
    capture invalidates it): the first large conversion of a fresh converter inside a capture takes the default family and leaves the choice
    open; the first large call outside a capture then measures."""
    import ctypes
    import torch
    n = (1 << 22) + 64
    typed = las.point_layout_from_las_point_format(las.Format(0), False, api=hip)
    src = VectorBuffer.new_from_layout(typed)
    src.resize(n)
    src.synth_fill(9, 0)
    dst = HashMapBuffer.new_from_layout(typed)
    dst.resize(n)
    conv = BufferLayoutConverter.for_layouts(typed, typed)
    conv.prepare(VectorBuffer, HashMapBuffer)
    g = torch.cuda.CUDAGraph()
    main = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    try:
        with torch.cuda.graph(g):
            hip.set_stream(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            conv.convert_into_range_async(src, range(0, n), dst, range(0, n))
    finally:
        hip.set_stream(ctypes.c_void_p(main))
    assert conv.family_choice(HashMapBuffer)[0] == -1
    del g  # (never replayed: a conversion's plan entries travel from host memory, so conversions are not among the calls the header documents as capturable)
    torch.cuda.synchronize()
    conv.convert_into(src, dst)
    assert conv.family_choice(HashMapBuffer)[0] in (0, 1)
    m = 20_000
    rec = src.get_point_range(range(n - m, n)).reshape(-1).view(typed.numpy_record_dtype())
    for a in typed.attributes():
        assert np.array_equal(dst.get_attribute_range(a.attribute_definition(), range(n - m, n)), rec[a.name()]), a.name()
