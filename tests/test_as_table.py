"""The `as` table (attribute_conversion.rs:184-271) driven through BufferLayoutConverter on single-attribute columnar
buffers.  Expectations come from tests/rust_as_ref.py (independent Python statement of Rust `as`) and from the literal
cases of SURVEY.md Appendix C.  CPU suite: pins the oracle.  GPU suite: pins the HIP kernels (bit-exact)."""
import itertools

import numpy as np
import pytest

from pasture_amd._capi import PasturePanic
from pasture_amd.buffers import HashMapBuffer, VectorBuffer
from pasture_amd.conversion import BufferLayoutConverter
from pasture_amd.layout import PointAttributeDataType as T, PointAttributeDefinition, PointLayout

from rust_as_ref import edge_values, rust_as_array

SCALARS = [T.U8, T.U16, T.U32, T.U64, T.I8, T.I16, T.I32, T.I64, T.F32, T.F64]
VEC3 = [T.Vec3u8, T.Vec3u16, T.Vec3i32, T.Vec3f32, T.Vec3f64]


def convert_column(api, values, from_t, to_t, src_kind=HashMapBuffer, dst_kind=HashMapBuffer):
    fl = PointLayout.from_attributes([PointAttributeDefinition("X", from_t)], api=api)
    tl = PointLayout.from_attributes([PointAttributeDefinition("X", to_t)], api=api)
    n = values.shape[0]
    src = src_kind.new_from_layout(fl)
    src.resize(n)
    src.set_attribute_range(PointAttributeDefinition("X", from_t), range(0, n), values)
    conv = BufferLayoutConverter.for_layouts(fl, tl)
    dst = conv.convert(src, dst_kind)
    return dst.view_attribute(PointAttributeDefinition("X", to_t))


def same_bits(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()


def nan_safe_equal(a, b):
    """bit-exact except that NaN payloads may differ (f64 NaN -> f32 NaN keeps 'a NaN', Rust guarantees no payload)."""
    if a.dtype.kind != "f":
        return same_bits(a, b)
    return a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b)) and same_bits(np.where(np.isnan(a), 0, a), np.where(np.isnan(b), 0, b))


@pytest.mark.parametrize("from_t,to_t", [(f, t) for f, t in itertools.permutations(SCALARS, 2)], ids=lambda t: str(t))
def test_scalar_as_pairs(api, from_t, to_t):
    rng = np.random.default_rng(1234 + from_t.kind * 31 + to_t.kind)
    npf = from_t.numpy_dtype()
    if npf.kind == "f":
        mags = 10.0 ** rng.uniform(-3, 25 if npf.itemsize == 8 else 20, size=200)
        with np.errstate(over="ignore"):
            rnd = (mags * rng.choice([-1.0, 1.0], size=200)).astype(npf)
    else:
        info = np.iinfo(npf)
        rnd = rng.integers(info.min, info.max, size=200, dtype=npf, endpoint=True)
    vals = np.concatenate([edge_values(npf), rnd])
    got = convert_column(api, vals, from_t, to_t)
    want = rust_as_array(vals, to_t.numpy_dtype())
    assert nan_safe_equal(got, want), f"{from_t}->{to_t}: first mismatch at {np.flatnonzero(got != want)[:5]}"


@pytest.mark.parametrize("from_t,to_t", [(f, t) for f, t in itertools.permutations(VEC3, 2)], ids=lambda t: str(t))
def test_vec3_as_pairs(api, from_t, to_t):
    npf = from_t.numpy_dtype()
    ev = edge_values(npf)
    ev = ev[: (ev.size // 3) * 3].reshape(-1, 3)
    got = convert_column(api, ev, from_t, to_t)
    want = rust_as_array(ev, to_t.numpy_dtype())
    assert nan_safe_equal(got, want)


def test_appendix_c_literals(api):
    """SURVEY.md Appendix C — the Rust-language answers written out as literals."""
    def one(v, f, t):
        return convert_column(api, np.array(v, dtype=f.numpy_dtype()), f, t).tolist()
    assert one([255.9, 256.0, -0.9, -1.0, float("nan"), float("inf")], T.F64, T.U8) == [255, 255, 0, 0, 0, 255]
    assert one([2147483647.5, -2147483649.0, float("nan")], T.F64, T.I32) == [2147483647, -2147483648, 0]
    assert one([1e30, -1.0], T.F32, T.I64) == [9223372036854775807, -1]
    assert one([1e30, -1.0], T.F64, T.U64) == [18446744073709551615, 0]
    assert one([511, 512, -1], T.I32, T.U8) == [255, 0, 255]
    assert one([511, 512], T.U16, T.U8) == [255, 0]
    assert one([-1], T.I64, T.I16) == [-1]
    assert one([-1], T.I8, T.U32) == [4294967295]
    assert one([-1], T.I16, T.U64) == [18446744073709551615]
    assert one([2 ** 64 - 1], T.U64, T.F32) == [18446744073709551616.0]
    assert one([16777217], T.I64, T.F32) == [16777216.0]
    assert one([2 ** 53 + 1], T.U64, T.F64) == [9007199254740992.0]
    r = convert_column(api, np.array([1e39, 1e-46, 1e-40, 0.1, float("nan")], dtype=np.float64), T.F64, T.F32)
    assert np.isposinf(r[0]) and r[1] == 0.0 and not np.signbit(r[1])
    assert r[2] == np.float32(1e-40) and r[2] != 0.0  # subnormal must not be flushed to zero
    assert r[3] == np.float32(0.1) and np.isnan(r[4])


def test_vec3u16_to_vec3u8_wraparound(api):
    """The colour wrap-around pinned by the LAS reader tests (raw_readers.rs:869-879, test_util.rs:136-149)."""
    colors = np.array([[i, (i + 1) << 4, (i + 2) << 8] for i in range(10)], dtype=np.uint16)
    got = convert_column(api, colors, T.Vec3u16, T.Vec3u8)
    assert got.tolist() == [[i, ((i + 1) << 4) & 255, 0] for i in range(10)]


@pytest.mark.parametrize("from_t,to_t", [(T.Vec4u8, T.U32), (T.U32, T.Vec4u8), (T.F64, T.Vec3f64), (T.Vec3f64, T.F64), (T.Vec3u8, T.U8),
                                         (T.ByteArray(4), T.U32), (T.ByteArray(4), T.ByteArray(5)), (T.U8, T.Vec3u8)], ids=lambda t: str(t))
def test_invalid_conversions_panic(api, from_t, to_t):
    """attribute_conversion.rs:267-269: 'Invalid conversion X -> Y'."""
    fl = PointLayout.from_attributes([PointAttributeDefinition("X", from_t)], api=api)
    tl = PointLayout.from_attributes([PointAttributeDefinition("X", to_t)], api=api)
    with pytest.raises(PasturePanic) as e:
        BufferLayoutConverter.for_layouts(fl, tl)
    assert "Invalid conversion" in str(e.value)


@pytest.mark.parametrize("pair", [("V", "V"), ("V", "H"), ("H", "V")])
def test_type_change_through_interleaved_buffers(api, pair):
    """The reference has no test with a datatype-changing converter on interleaved buffers; the bench layouts
    (layout_conversion_bench.rs:15-39: Vec3f64->Vec3f32, u8->u32, u16->u8, f64 copy) exercise exactly that."""
    from harness import BUFFER_KINDS, make_buffer, random_records
    from pasture_amd.layout import attributes as A
    src_l = PointLayout.from_attributes_packed([A.POSITION_3D, A.CLASSIFICATION, A.INTENSITY, A.GPS_TIME], 1, api=api)
    dst_l = PointLayout.from_attributes_packed([A.GPS_TIME, A.POSITION_3D.with_custom_datatype(T.Vec3f32),
                                                A.CLASSIFICATION.with_custom_datatype(T.U32), A.INTENSITY.with_custom_datatype(T.U8)], 1, api=api)
    assert src_l.size_of_point_entry() == 35 and dst_l.size_of_point_entry() == 25
    rec = random_records(src_l, 1000, seed=5)
    rec["Position3D"] *= 1000.0
    src = make_buffer(pair[0], src_l, rec)
    out = BufferLayoutConverter.for_layouts(src_l, dst_l).convert(src, BUFFER_KINDS[pair[1]])
    assert same_bits(out.view_attribute(A.GPS_TIME), rec["GpsTime"])
    assert same_bits(out.view_attribute(A.POSITION_3D.with_custom_datatype(T.Vec3f32)), rec["Position3D"].astype(np.float32))
    assert same_bits(out.view_attribute(A.CLASSIFICATION.with_custom_datatype(T.U32)), rec["Classification"].astype(np.uint32))
    assert same_bits(out.view_attribute(A.INTENSITY.with_custom_datatype(T.U8)), (rec["Intensity"] & 255).astype(np.uint8))
