"""Points far larger than the kernels' LDS tiles: ByteArray(n) attributes of 70 KB and 1 MB (the reference takes any u64 length,
point_layout.rs:57), through every conversion pairing, the fused AABB, compaction and append -- against numpy on the same random records; and the
refusal of layouts whose points reach 4 GiB (the kernels address a point's bytes with 32-bit strides)."""
import numpy as np
import pytest

from harness import BUFFER_KINDS, PAIRINGS, random_records
from pasture_amd._capi import PastureError
from pasture_amd.algorithms import calculate_bounds
from pasture_amd.buffers import HashMapBuffer, VectorBuffer
from pasture_amd.conversion import BufferLayoutConverter, Transform
from pasture_amd.layout import PointAttributeDataType as T, PointAttributeDefinition, PointLayout, attributes as A

pytestmark = pytest.mark.gpu
SCALE, OFFSET = (0.001, 0.002, 0.004), (1.5, -2.5, 100.0)


def _layout(api, blob_bytes, blob_first):
    blob = PointAttributeDefinition("Blob", T.ByteArray(blob_bytes))
    attrs = [blob, A.POSITION_3D, A.INTENSITY] if blob_first else [A.POSITION_3D, A.INTENSITY, blob]
    return PointLayout.from_attributes_packed(attrs, 1, api=api), blob


@pytest.mark.parametrize("pair", PAIRINGS)
@pytest.mark.parametrize("blob_bytes,n,blob_first", [(70_001, 301, False), (70_000, 257, True), (1_000_003, 37, False)])
def test_conversions_of_points_with_large_byte_arrays(hip, blob_bytes, n, blob_first, pair):
    layout, blob = _layout(hip, blob_bytes, blob_first)
    other, _ = _layout(hip, blob_bytes, not blob_first)  # same attributes, the other order
    rec = random_records(layout, n, blob_bytes % 97)
    rec["Position3D"] = np.random.default_rng(1).random((n, 3)) * 1000.0
    src = BUFFER_KINDS[pair[0]].from_numpy(rec, layout)
    conv = BufferLayoutConverter.for_layouts(layout, other)
    conv.set_custom_mapping_with_transformation(A.POSITION_3D, A.POSITION_3D, Transform.affine(T.Vec3f64, SCALE, OFFSET), False)
    dst = BUFFER_KINDS[pair[1]].new_from_layout(other)
    dst.resize(n)
    fused = conv.convert_into_with_bounds(src, dst)
    want_pos = (rec["Position3D"] * np.array(SCALE)) + np.array(OFFSET)
    assert dst.get_attribute_range(A.POSITION_3D, range(0, n)).tobytes() == want_pos.tobytes()
    assert dst.get_attribute_range(A.INTENSITY, range(0, n)).tobytes() == rec["Intensity"].tobytes()
    assert dst.get_attribute_range(blob, range(0, n)).tobytes() == np.ascontiguousarray(rec["Blob"]).tobytes()
    assert fused.min() == tuple(want_pos.min(axis=0)) and fused.max() == tuple(want_pos.max(axis=0))
    assert calculate_bounds(dst) == fused
    # a sub-range in the middle, into a zeroed target: the neighbours stay zero
    part = BUFFER_KINDS[pair[1]].new_from_layout(other)
    part.resize(n)
    conv.convert_into_range(src, range(3, n - 5), part, range(4, n - 4))
    got = part.get_attribute_range(blob, range(0, n))
    assert not got[:4].any() and not got[n - 4:].any() and got[4:n - 4].tobytes() == np.ascontiguousarray(rec["Blob"][3:n - 5]).tobytes()


@pytest.mark.parametrize("out_kind", ["V", "H"])
def test_compaction_and_append_of_points_with_large_byte_arrays(hip, out_kind):
    layout, blob = _layout(hip, 70_001, False)
    n = 203
    rec = random_records(layout, n, 5)
    src = HashMapBuffer.from_numpy(rec, layout)
    mask = (np.arange(n) % 3 != 1)
    kept = src.filter(BUFFER_KINDS[out_kind], mask)
    assert kept.len() == int(mask.sum())
    assert kept.get_attribute_range(blob, range(0, kept.len())).tobytes() == np.ascontiguousarray(rec["Blob"][mask]).tobytes()
    assert kept.get_attribute_range(A.INTENSITY, range(0, kept.len())).tobytes() == rec["Intensity"][mask].tobytes()
    # filter_into a longer target with Some(num_matches) larger than the count: the points behind the matches keep what they held
    target = BUFFER_KINDS[out_kind].from_numpy(rec[::-1].copy(), layout)
    k = int(mask.sum())
    assert src.filter_into(target, mask, k + 7) == k
    got = target.get_attribute_range(blob, range(0, n))
    assert got[:k].tobytes() == np.ascontiguousarray(rec["Blob"][mask]).tobytes() and got[k:].tobytes() == np.ascontiguousarray(rec["Blob"][::-1][k:]).tobytes()
    assert target.get_attribute_range(A.INTENSITY, range(0, n)).tobytes() == np.concatenate([rec["Intensity"][mask], rec["Intensity"][::-1][k:]]).tobytes()
    # the predicate as an expression (points this wide keep the byte-mask path)
    sel = rec["Intensity"] > 30000
    by_expr = src.filter_expr(BUFFER_KINDS[out_kind], "Intensity > 30000")
    assert by_expr.len() == int(sel.sum()) and by_expr.get_attribute_range(blob, range(0, by_expr.len())).tobytes() == np.ascontiguousarray(rec["Blob"][sel]).tobytes()
    both = BUFFER_KINDS[out_kind].from_numpy(rec[:50], layout)
    both.append(kept)
    assert both.len() == 50 + kept.len()
    assert both.get_attribute_range(blob, range(0, both.len())).tobytes() == np.ascontiguousarray(np.concatenate([rec["Blob"][:50], rec["Blob"][mask]])).tobytes()


@pytest.mark.parametrize("kind", ["V", "H"])
def test_swap_of_points_wider_than_the_scratch(hip, kind):
    """BorrowedMutBuffer::swap (point_buffer.rs:229) on 2.5-MB points: the exchange goes through 1 MiB of device scratch piece by piece."""
    layout, blob = _layout(hip, 2_500_003, False)
    rec = random_records(layout, 5, 8)
    buf = BUFFER_KINDS[kind].from_numpy(rec, layout)
    buf.swap(1, 4)
    want = rec.copy()
    want[[1, 4]] = want[[4, 1]]
    assert buf.get_attribute_range(blob, range(0, 5)).tobytes() == np.ascontiguousarray(want["Blob"]).tobytes()
    assert buf.get_attribute_range(A.POSITION_3D, range(0, 5)).tobytes() == np.ascontiguousarray(want["Position3D"]).tobytes()


def test_points_of_four_gib_are_refused_where_they_meet_device_memory(hip):
    """ByteArray(2^32): a layout like any other on the host (sizes and offsets are u64, point_layout.rs:57), PST_ERR_UNSUPPORTED for buffers and converters."""
    huge = PointLayout.from_attributes_packed([A.POSITION_3D, PointAttributeDefinition("Blob", T.ByteArray(1 << 32))], 1, api=hip)
    assert huge.size_of_point_entry() == (1 << 32) + 24
    small = PointLayout.from_attributes([A.POSITION_3D], api=hip)
    for make in (lambda: VectorBuffer.new_from_layout(huge), lambda: HashMapBuffer.new_from_layout(huge),
                 lambda: BufferLayoutConverter.for_layouts_with_default(small, huge), lambda: BufferLayoutConverter.for_layouts(huge, small)):
        with pytest.raises(PastureError, match="4 GiB"):
            make()
