"""Layouts with far more attributes than one kernel launch takes (PST_PLAN_MAX_ENTRIES mappings per conversion launch, kMaxFilterAttrs attributes per
compaction launch): 150 attributes of mixed datatypes, reordered and partly cast, through the four conversion pairings, compaction into both buffer
kinds (byte mask and expression), append, min-max of the last attribute -- against numpy on the same random records.  And attributes of zero bytes
(ByteArray(0): legal in the reference, point_layout.rs:57) between ordinary ones."""
import numpy as np
import pytest

from harness import BUFFER_KINDS, PAIRINGS, random_records
from pasture_amd.algorithms import minmax_attribute
from pasture_amd.buffers import HashMapBuffer
from pasture_amd.conversion import BufferLayoutConverter
from pasture_amd.layout import PointAttributeDataType as T, PointAttributeDefinition, PointLayout, attributes as A

pytestmark = pytest.mark.gpu
SCALARS = [T.U8, T.I8, T.U16, T.I16, T.U32, T.I32, T.U64, T.I64, T.F32, T.F64]
WIDER = {T.U8: T.U16, T.I8: T.I32, T.U16: T.F32, T.I16: T.I64, T.U32: T.U64, T.F32: T.F64}


def _layouts(api, n_attrs):
    src_defs, dst_defs = [], []
    for i in range(n_attrs):
        dt = SCALARS[(i * 7) % len(SCALARS)] if i % 11 else T.Vec3f64
        d = PointAttributeDefinition(f"a{i:03d}", dt)
        src_defs.append(d)
        dst_defs.append(d.with_custom_datatype(WIDER[dt]) if i % 3 == 0 and dt in WIDER else d)
    return (PointLayout.from_attributes_packed(src_defs, 1, api=api), PointLayout.from_attributes_packed(dst_defs[::-1], 1, api=api), src_defs, dst_defs)


@pytest.mark.parametrize("pair", PAIRINGS)
def test_150_attributes_reordered_and_cast(hip, pair):
    n = 10_007
    sl, tl, src_defs, dst_defs = _layouts(hip, 150)
    rec = random_records(sl, n, 11)
    src = BUFFER_KINDS[pair[0]].from_numpy(rec, sl)
    conv = BufferLayoutConverter.for_layouts(sl, tl)
    for s, d in zip(src_defs, dst_defs):
        if s.datatype() != d.datatype():
            conv.set_custom_mapping(s, d)
    out = conv.convert(src, BUFFER_KINDS[pair[1]])
    for s, d in zip(src_defs, dst_defs):
        want = rec[s.name()].astype(d.datatype().numpy_dtype())
        assert out.get_attribute_range(d, range(0, n)).tobytes() == np.ascontiguousarray(want).tobytes(), (s.name(), str(s.datatype()), str(d.datatype()))


@pytest.mark.parametrize("out_kind", ["V", "H"])
def test_150_attributes_compaction_append_minmax(hip, out_kind):
    n = 30_011
    sl, _, src_defs, _ = _layouts(hip, 150)
    rec = random_records(sl, n, 12)
    src = HashMapBuffer.from_numpy(rec, sl)
    mask = np.random.default_rng(3).random(n) < 0.4
    kept = src.filter(BUFFER_KINDS[out_kind], mask)
    first = src_defs[1]  # a scalar
    by_expr = src.filter_expr(BUFFER_KINDS[out_kind], f"{first.name()} % 2 == 0")
    sel = rec[first.name()] % 2 == 0
    assert kept.len() == int(mask.sum()) and by_expr.len() == int(sel.sum())
    for d in src_defs:
        assert kept.get_attribute_range(d, range(0, kept.len())).tobytes() == np.ascontiguousarray(rec[d.name()][mask]).tobytes(), d.name()
        assert by_expr.get_attribute_range(d, range(0, by_expr.len())).tobytes() == np.ascontiguousarray(rec[d.name()][sel]).tobytes(), d.name()
    kept.append(by_expr)
    last = src_defs[-1]
    assert kept.get_attribute_range(last, range(0, kept.len())).tobytes() == np.concatenate([rec[last.name()][mask], rec[last.name()][sel]]).tobytes()
    lo, hi = minmax_attribute(src, last)
    assert lo == rec[last.name()].min() and hi == rec[last.name()].max()


@pytest.mark.parametrize("pair", PAIRINGS)
def test_zero_byte_attributes_between_ordinary_ones(hip, pair):
    n = 5_003
    empty = PointAttributeDefinition("Nothing", T.ByteArray(0))
    also = PointAttributeDefinition("NothingElse", T.ByteArray(0))
    sl = PointLayout.from_attributes_packed([A.INTENSITY, empty, A.POSITION_3D, also, A.CLASSIFICATION], 1, api=hip)
    tl = PointLayout.from_attributes_packed([also, A.CLASSIFICATION, A.POSITION_3D, empty, A.INTENSITY], 1, api=hip)
    assert sl.size_of_point_entry() == 27 == tl.size_of_point_entry()
    rec = np.zeros(n, dtype=np.dtype({"names": ["Intensity", "Position3D", "Classification"], "formats": ["<u2", ("<f8", (3,)), "u1"], "offsets": [0, 2, 26], "itemsize": 27}))
    rng = np.random.default_rng(2)
    rec["Intensity"] = rng.integers(0, 65535, n)
    rec["Position3D"] = rng.random((n, 3))
    rec["Classification"] = rng.integers(0, 255, n)
    src = BUFFER_KINDS[pair[0]].new_from_layout(sl)
    src.resize(n)
    for d in (A.INTENSITY, A.POSITION_3D, A.CLASSIFICATION):
        src.set_attribute_range(d, range(0, n), rec[d.name()])
    out = BufferLayoutConverter.for_layouts(sl, tl).convert(src, BUFFER_KINDS[pair[1]])
    for d in (A.INTENSITY, A.POSITION_3D, A.CLASSIFICATION):
        assert out.get_attribute_range(d, range(0, n)).tobytes() == np.ascontiguousarray(rec[d.name()]).tobytes(), d.name()
    if pair[0] == "H":
        kept = src.filter(BUFFER_KINDS[pair[1]], rec["Classification"] < 100)
        assert kept.get_attribute_range(A.INTENSITY, range(0, kept.len())).tobytes() == rec["Intensity"][rec["Classification"] < 100].tobytes()
