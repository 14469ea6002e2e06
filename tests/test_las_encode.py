"""LAS record encoder = the write half of the LAS pipeline: RawLASWriter::write_points_default_layout
(pasture-io/src/las/raw_writers.rs:203-363) with write_position_as_las_position / write_las_bit_attributes
(write_helpers.rs:10-55).  Golden vectors: the typed reference data of test_util.rs (las_expected.py), encoded with the
fixtures' scale 1 / offset 0, must reproduce the fixtures' raw point records byte for byte (that is what the reference's
writer tests check by reading the written file back, raw_writers.rs tests / las_writer tests)."""
import os

import numpy as np
import pytest

import las_expected as E
from harness import BUFFER_KINDS
from pasture_amd import las
from pasture_amd._capi import PasturePanic
from pasture_amd.buffers import HashMapBuffer, VectorBuffer
from pasture_amd.layout import attributes as A

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "las")
F64_MAX = 1.7976931348623157e308


def reference_points(fmt: int, layout) -> np.ndarray:
    """test_data_* of test_util.rs as records of the format's default layout."""
    F = las.Format(fmt)
    rec = np.zeros(E.N, dtype=layout.numpy_record_dtype())
    rec[A.POSITION_3D.name()] = E.POSITIONS
    rec[A.INTENSITY.name()] = E.INTENSITIES
    rec[A.RETURN_NUMBER.name()] = E.RETURN_NUMBERS_EXTENDED if F.is_extended else E.RETURN_NUMBERS
    rec[A.NUMBER_OF_RETURNS.name()] = E.NUMBER_OF_RETURNS_EXTENDED if F.is_extended else E.NUMBER_OF_RETURNS
    if F.is_extended:
        rec[A.CLASSIFICATION_FLAGS.name()] = E.CLASSIFICATION_FLAGS
        rec[A.SCANNER_CHANNEL.name()] = E.SCANNER_CHANNELS
        rec[A.SCAN_ANGLE.name()] = E.SCAN_ANGLES_EXTENDED
    else:
        rec[A.SCAN_ANGLE_RANK.name()] = E.SCAN_ANGLE_RANKS
    rec[A.SCAN_DIRECTION_FLAG.name()] = E.SCAN_DIRECTION_FLAGS
    rec[A.EDGE_OF_FLIGHT_LINE.name()] = E.EDGE_OF_FLIGHT_LINES
    rec[A.CLASSIFICATION.name()] = E.CLASSIFICATIONS
    rec[A.USER_DATA.name()] = E.USER_DATA
    rec[A.POINT_SOURCE_ID.name()] = E.POINT_SOURCE_IDS
    if F.has_gps_time:
        rec[A.GPS_TIME.name()] = E.GPS_TIMES
    if F.has_color:
        rec[A.COLOR_RGB.name()] = E.COLORS
    if F.has_nir:
        rec[A.NIR.name()] = E.NIRS
    if F.has_waveform:
        rec[A.WAVE_PACKET_DESCRIPTOR_INDEX.name()] = E.WAVEPACKET_INDEX
        rec[A.WAVEFORM_DATA_OFFSET.name()] = E.WAVEPACKET_OFFSET
        rec[A.WAVEFORM_PACKET_SIZE.name()] = E.WAVEPACKET_SIZE
        rec[A.RETURN_POINT_WAVEFORM_LOCATION.name()] = E.WAVEPACKET_LOCATION
        rec[A.WAVEFORM_PARAMETERS.name()] = E.WAVEPACKET_PARAMETERS
    return rec


def layouts(fmt, api):
    F = las.Format(fmt)
    return las.point_layout_from_las_point_format(F, False, api=api), las.point_layout_from_las_point_format(F, True, api=api)


def raw_bytes(buf) -> np.ndarray:
    n = buf.len()
    return np.ascontiguousarray(buf.get_point_range(range(0, n))).view(np.uint8).reshape(n, -1)


def numpy_encode(rec, fmt, scale, offset) -> np.ndarray:
    """Independent numpy restatement used to cross-check oracle and HIP on random data (positions assumed in range)."""
    F = las.Format(fmt)
    n = len(rec)
    parts = []
    local = (rec[A.POSITION_3D.name()] - np.asarray(offset)) / np.asarray(scale)
    parts.append(np.trunc(local).astype(np.int64).astype("<i4").view(np.uint8).reshape(n, 12))
    parts.append(rec[A.INTENSITY.name()].astype("<u2").view(np.uint8).reshape(n, 2))
    rn, nr = rec[A.RETURN_NUMBER.name()], rec[A.NUMBER_OF_RETURNS.name()]
    sd, eof = rec[A.SCAN_DIRECTION_FLAG.name()], rec[A.EDGE_OF_FLIGHT_LINE.name()]
    if F.is_extended:
        cf, sc = rec[A.CLASSIFICATION_FLAGS.name()], rec[A.SCANNER_CHANNEL.name()]
        parts.append(((rn & 15) | ((nr & 15) << 4)).astype(np.uint8).reshape(n, 1))
        parts.append(((cf & 15) | ((sc & 3) << 4) | ((sd & 1) << 6) | ((eof & 1) << 7)).astype(np.uint8).reshape(n, 1))
    else:
        parts.append(((rn & 7) | ((nr & 7) << 3) | ((sd & 1) << 6) | ((eof & 1) << 7)).astype(np.uint8).reshape(n, 1))
    parts.append(rec[A.CLASSIFICATION.name()].reshape(n, 1))
    if F.is_extended:
        parts.append(rec[A.USER_DATA.name()].reshape(n, 1))
        parts.append(rec[A.SCAN_ANGLE.name()].astype("<i2").view(np.uint8).reshape(n, 2))
    else:
        parts.append(rec[A.SCAN_ANGLE_RANK.name()].view(np.uint8).reshape(n, 1))
        parts.append(rec[A.USER_DATA.name()].reshape(n, 1))
    parts.append(rec[A.POINT_SOURCE_ID.name()].astype("<u2").view(np.uint8).reshape(n, 2))
    tail = [(F.has_gps_time, A.GPS_TIME), (F.has_color, A.COLOR_RGB), (F.has_nir, A.NIR), (F.has_waveform, A.WAVE_PACKET_DESCRIPTOR_INDEX),
            (F.has_waveform, A.WAVEFORM_DATA_OFFSET), (F.has_waveform, A.WAVEFORM_PACKET_SIZE), (F.has_waveform, A.RETURN_POINT_WAVEFORM_LOCATION),
            (F.has_waveform, A.WAVEFORM_PARAMETERS)]
    for present, attr in tail:
        if present:
            parts.append(np.ascontiguousarray(rec[attr.name()]).view(np.uint8).reshape(n, -1))
    return np.concatenate(parts, axis=1)


def random_typed(layout, n, seed):
    from harness import random_records
    rec = random_records(layout, n, seed)
    rng = np.random.default_rng(seed + 1)
    rec[A.POSITION_3D.name()] = rng.uniform(-5000.0, 5000.0, size=(n, 3))
    return rec


@pytest.mark.parametrize("fmt", range(11))
@pytest.mark.parametrize("src_kind", ["V", "H"])
def test_encode_reference_data_reproduces_fixture_records(api, fmt, src_kind):
    typed, raw = layouts(fmt, api)
    f = las.read_las_records(os.path.join(GOLDEN, f"10_points_format_{fmt}.las"))
    src = BUFFER_KINDS[src_kind].from_numpy(reference_points(fmt, typed), typed)
    dst = VectorBuffer.new_from_layout(raw)
    dst.resize(10)
    bounds, counts = las.encode_points(src, fmt, f.scale, f.offset, dst)
    fixture = np.ascontiguousarray(f.records).view(np.uint8).reshape(10, -1)
    assert np.array_equal(raw_bytes(dst), fixture)
    assert bounds == E.BOUNDS  # header bounds of the fixture files, test_util.rs:46-48
    rn = E.RETURN_NUMBERS_EXTENDED if fmt >= 6 else E.RETURN_NUMBERS
    assert counts == [int((rn == r).sum()) for r in range(1, 16)]


@pytest.mark.parametrize("fmt", range(11))
def test_encode_then_read_round_trip(api, fmt):
    """write -> read: records produced by the encoder, read back with get_default_las_converter, are the input (positions on the grid)."""
    typed, raw = layouts(fmt, api)
    n = 777
    rec = random_typed(typed, n, 100 + fmt)
    scale, offset = (0.01, 0.01, 0.01), (10.0, -20.0, 30.0)
    grid = np.round(rec[A.POSITION_3D.name()] * 100.0)
    rec[A.POSITION_3D.name()] = grid * np.asarray(scale) + np.asarray(offset)  # exactly what the reader reconstructs
    if fmt >= 6:
        for name, m in [(A.RETURN_NUMBER, 15), (A.NUMBER_OF_RETURNS, 15), (A.CLASSIFICATION_FLAGS, 15), (A.SCANNER_CHANNEL, 3)]:
            rec[name.name()] &= m
    else:
        for name in (A.RETURN_NUMBER, A.NUMBER_OF_RETURNS):
            rec[name.name()] &= 7
    for name in (A.SCAN_DIRECTION_FLAG, A.EDGE_OF_FLIGHT_LINE):
        rec[name.name()] &= 1
    src = HashMapBuffer.from_numpy(rec, typed)
    dst = VectorBuffer.new_from_layout(raw)
    dst.resize(n)
    las.encode_points(src, fmt, scale, offset, dst)
    back = HashMapBuffer.new_from_layout(typed)
    back.resize(n)
    las.get_default_las_converter(raw, typed, scale, offset).convert_into(dst, back)
    for a in typed.attributes():
        got, want = back.view_attribute(a.attribute_definition()), rec[a.name()]
        if a.name() == A.POSITION_3D.name():
            # (w-o)/s truncates toward zero, so a grid value may come back one step low; never more
            steps = np.round((want - got) / np.asarray(scale))
            assert np.all((steps == 0) | (steps == 1) | (steps == -1))
        else:
            assert np.array_equal(got, want), a.name()


@pytest.mark.parametrize("fmt", [0, 1, 3, 6, 8, 10])
def test_encode_random_matches_numpy(api, fmt):
    typed, raw = layouts(fmt, api)
    n = 5003
    rec = random_typed(typed, n, 7 + fmt)
    scale, offset = (0.001, 0.002, 0.004), (1.5, -2.5, 100.0)
    src = VectorBuffer.from_numpy(rec, typed)
    dst = VectorBuffer.new_from_layout(raw)
    dst.resize(n + 5)
    hb = [0.0, 0.0, 0.0, 1.0, 1.0, 1.0]  # the header already holds bounds from earlier writes
    bounds, counts = las.encode_points(src, fmt, scale, offset, dst, target_first=3, header_bounds=hb, max_return=5 if fmt < 6 else 15)
    out = raw_bytes(dst)
    assert not out[:3].any() and not out[n + 3:].any()
    assert np.array_equal(out[3:n + 3], numpy_encode(rec, fmt, scale, offset))
    pos = rec[A.POSITION_3D.name()]
    assert bounds == (tuple(np.minimum(pos.min(axis=0), 0.0)), tuple(np.maximum(pos.max(axis=0), 1.0)))
    rn = rec[A.RETURN_NUMBER.name()]
    assert counts == [int((rn == r).sum()) for r in range(1, (5 if fmt < 6 else 15) + 1)]


def test_position_out_of_bounds_panics(api):
    """write_helpers.rs:17-22."""
    typed, raw = layouts(0, api)
    rec = random_typed(typed, 100, 3)
    rec[A.POSITION_3D.name()][57, 1] = 3.0e9
    dst = VectorBuffer.new_from_layout(raw)
    dst.resize(100)
    with pytest.raises(PasturePanic, match="out of bounds given the current LAS offset and scale"):
        las.encode_points(VectorBuffer.from_numpy(rec, typed), 0, (1.0, 1.0, 1.0), (0.0, 0.0, 0.0), dst)
    rec[A.POSITION_3D.name()][57, 1] = float("nan")  # NaN as i64 = 0: in range (Rust `as`)
    las.encode_points(VectorBuffer.from_numpy(rec, typed), 0, (1.0, 1.0, 1.0), (0.0, 0.0, 0.0), dst)
    assert raw_bytes(dst)[57, 4:8].view("<i4")[0] == 0


def test_wrong_layouts_and_ranges(api):
    typed, raw = layouts(1, api)
    typed0, raw0 = layouts(0, api)
    src = VectorBuffer.from_numpy(random_typed(typed, 10, 1), typed)
    dst = VectorBuffer.new_from_layout(raw)
    dst.resize(5)
    with pytest.raises(PasturePanic):
        las.encode_points(src, 1, (1, 1, 1), (0, 0, 0), dst)  # target too short
    dst.resize(10)
    with pytest.raises(PasturePanic):
        las.encode_points(src, 0, (1, 1, 1), (0, 0, 0), dst)  # source is not format 0's default layout
    empty = VectorBuffer.new_from_layout(typed)
    b, c = las.encode_points(empty, 1, (1, 1, 1), (0, 0, 0), dst)
    assert b == ((F64_MAX,) * 3, (-F64_MAX,) * 3) and c == [0] * 15


@pytest.mark.parametrize("fmt", [0, 3, 7])
@pytest.mark.parametrize("src_kind", ["V", "H"])
def test_write_points_custom_layout(api, fmt, src_kind):
    """write_points_custom_layout (raw_writers.rs:365-603): positions as Vec3f32, classification as u32 (wraps through `as`),
    intensity present, everything else missing -> Default::default(); points-by-return stay zero (reference quirk)."""
    from pasture_amd.layout import PointAttributeDataType as T, PointLayout
    typed, raw = layouts(fmt, api)
    n = 1000
    pos32 = A.POSITION_3D.with_custom_datatype(T.Vec3f32)
    cls32 = A.CLASSIFICATION.with_custom_datatype(T.U32)
    custom = PointLayout.from_attributes([cls32, pos32, A.INTENSITY, A.RETURN_NUMBER, A.NORMAL], api=api)
    rng = np.random.default_rng(5)
    rec = np.zeros(n, dtype=custom.numpy_record_dtype())
    rec[pos32.name()] = rng.uniform(-100, 100, size=(n, 3)).astype(np.float32)
    rec[cls32.name()] = rng.integers(0, 2**32, size=n, dtype=np.uint32)
    rec[A.INTENSITY.name()] = rng.integers(0, 65536, size=n, dtype=np.uint16)
    rec[A.RETURN_NUMBER.name()] = rng.integers(0, 8, size=n, dtype=np.uint8)
    src = BUFFER_KINDS[src_kind].from_numpy(rec, custom)
    dst = VectorBuffer.new_from_layout(raw)
    dst.resize(n)
    scale, offset = (0.01, 0.01, 0.01), (0.0, 0.0, 0.0)
    bounds, counts = las.write_points(src, fmt, scale, offset, dst)
    exp = np.zeros(n, dtype=typed.numpy_record_dtype())
    exp[A.POSITION_3D.name()] = rec[pos32.name()].astype(np.float64)
    exp[A.CLASSIFICATION.name()] = rec[cls32.name()].astype(np.uint8)
    exp[A.INTENSITY.name()] = rec[A.INTENSITY.name()]
    exp[A.RETURN_NUMBER.name()] = rec[A.RETURN_NUMBER.name()]
    assert np.array_equal(raw_bytes(dst), numpy_encode(exp, fmt, scale, offset))
    p = exp[A.POSITION_3D.name()]
    assert bounds == (tuple(p.min(axis=0)), tuple(p.max(axis=0)))
    assert counts == [0] * 15
    # the default layout dispatches to the default-layout writer (counts are kept there)
    d = BUFFER_KINDS[src_kind].from_numpy(exp, typed)
    _, counts2 = las.write_points(d, fmt, scale, offset, dst)
    assert counts2[:7] == [int((exp[A.RETURN_NUMBER.name()] == r).sum()) for r in range(1, 8)]


_DIV_SCALES = [
    ((0.001, 0.01, 0.1), (500000.0, 5400000.0, 100.0)),
    ((1.0 / 3.0, 0.0254, 1e-7), (0.0, -12.5, 1e-3)),
    ((0.125, 3.0, 7.0), (1.0, 2.0, 3.0)),
    ((0.0009765625 * float(np.nextafter(2.0, 0.0)), 1e-3, 1e-2), (0.0, 0.0, 0.0)),  # a significand of all ones: the division sequence
    ((-0.01, 2.5e-4, 1.0), (3.0, 3.0, 3.0)),                                            # a negative scale divides like any other
    ((1e-200, 0.3, 1e150), (0.0, 0.5, 0.0)),                                            # outside the reciprocal form's window
    ("random", (123.456, -7.0, 0.0)),
]


@pytest.mark.parametrize("case", range(len(_DIV_SCALES)))
@pytest.mark.parametrize("src_kind", ["V", "H"])
def test_reciprocal_division_is_the_ieee_quotient(api, case, src_kind, monkeypatch):
    """write_helpers.rs:15-17: `((p - offset) / scale) as i64` truncates the IEEE quotient.  The HIP encoder forms it from a host-side
    reciprocal and two fused correction steps (las_encode.hip quotient_by_reciprocal); positions ON the LAS grid sit within an ulp of an
    integer quotient, where the last bit decides the record.  Adversarial inputs: grid points k * scale + offset over the whole i32 range and
    their neighbours a few ulps away, against numpy's IEEE division; the HIP path runs both forms (the division instruction sequence = the
    default, and PST_LAS_RECIPROCAL_DIV=1 = host reciprocal + two fused correction steps) and both must give these bytes."""
    typed, raw = layouts(0, api)
    n = 200_000 + 37
    rng = np.random.default_rng(1000 + case)
    scale, offset = _DIV_SCALES[case]
    if scale == "random":
        scale = tuple(float(v) for v in rng.uniform(1e-4, 10.0, size=3))
    rec = random_typed(typed, n, 50 + case)
    k = np.empty((n, 3), dtype=np.int64)
    k[: n // 2] = rng.integers(-(2**31) + 8, 2**31 - 8, size=(n // 2, 3))
    k[n // 2:] = rng.integers(-100000, 100000, size=(n - n // 2, 3))
    s, o = np.asarray(scale), np.asarray(offset)
    with np.errstate(over="ignore"):
        p = k.astype(np.float64) * s + o
    for step in range(3):  # every fourth point 1, 2, 3 ulps up / down
        up = np.nextafter(p, np.inf)
        down = np.nextafter(p, -np.inf)
        sel = rng.integers(0, 4, size=p.shape)
        p = np.where(sel == 0, up, np.where(sel == 1, down, p))
    with np.errstate(over="ignore", invalid="ignore"):
        local = (p - o) / s
    ok = np.isfinite(local) & (np.abs(local) < 2147483000.0)
    p = np.where(ok, p, o)  # (in range only: numpy_encode has no checked narrowing)
    rec[A.POSITION_3D.name()] = p
    want = numpy_encode(rec, 0, scale, offset)
    src = BUFFER_KINDS[src_kind].from_numpy(rec, typed)
    for exact in ("0", "1"):
        monkeypatch.setenv("PST_LAS_RECIPROCAL_DIV", exact)
        dst = VectorBuffer.new_from_layout(raw)
        dst.resize(n)
        las.encode_points(src, 0, scale, offset, dst)
        got = raw_bytes(dst)
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert bad.size == 0, (exact, bad[:5], p[bad[:5]], got[bad[:5], :12].view("<i4"), want[bad[:5], :12].view("<i4"))
