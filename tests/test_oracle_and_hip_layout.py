"""PointLayout known answers — every literal below is asserted by the reference's own tests / doc-tests.
Runs against the oracle (CPU suite) and against the product's host logic; neither needs a GPU."""
import pytest

from pasture_amd._capi import PasturePanic
from pasture_amd.layout import FieldAlignment, PointAttributeDataType as T, PointAttributeDefinition, PointLayout, attributes as A
from pasture_amd import las


@pytest.fixture(params=["oracle", "hip_host"])
def lapi(request, oracle):
    if request.param == "oracle":
        return oracle
    from pasture_amd import product_api
    return product_api()  # layouts are host-only logic: no device needed


def offsets(layout):
    return [(a.name(), a.offset()) for a in layout.attributes()]


def test_datatype_sizes_and_alignments():
    # point_layout.rs:72-126, pasture-derive/src/lib.rs:36-56
    expect = {T.U8: (1, 1), T.I8: (1, 1), T.U16: (2, 2), T.I16: (2, 2), T.U32: (4, 4), T.I32: (4, 4), T.U64: (8, 8), T.I64: (8, 8),
              T.F32: (4, 4), T.F64: (8, 8), T.Vec3u8: (3, 1), T.Vec3u16: (6, 2), T.Vec3f32: (12, 4), T.Vec3i32: (12, 4),
              T.Vec3f64: (24, 8), T.Vec4u8: (4, 1)}
    for dt, (s, a) in expect.items():
        assert dt.size() == s and dt.min_alignment() == a
    assert T.ByteArray(7).size() == 7 and T.ByteArray(7).min_alignment() == 1


def test_from_attributes_packed_doc(lapi):  # point_layout.rs:680-691
    l1 = PointLayout.from_attributes_packed([A.INTENSITY, A.POSITION_3D], 1, api=lapi)
    assert l1.at(0).offset() == 0 and l1.at(1).offset() == 2
    l4 = PointLayout.from_attributes_packed([A.INTENSITY, A.POSITION_3D], 4, api=lapi)
    assert l4.at(1).offset() == 4


def test_from_members_and_alignment_doc(lapi):  # :711-717
    l = PointLayout.from_members_and_alignment([A.INTENSITY.at_offset_in_type(2), A.POSITION_3D.at_offset_in_type(8)], 8, api=lapi)
    assert l.at(0).offset() == 2 and l.at(1).offset() == 8 and l.size_of_point_entry() == 32


def test_add_attribute_doc(lapi):  # :767-776, :905-912, :920-927
    l = PointLayout.default(lapi)
    l.add_attribute(A.INTENSITY, FieldAlignment.Default)
    l.add_attribute(A.POSITION_3D, FieldAlignment.Default)
    assert offsets(l) == [("Intensity", 0), ("Position3D", 8)]
    l2 = PointLayout.from_attributes([A.POSITION_3D, A.INTENSITY], api=lapi)
    assert offsets(l2) == [("Position3D", 0), ("Intensity", 24)]
    assert l2.size_of_point_entry() == 32 and l2.alignment() == 8
    assert l2.index_of(A.POSITION_3D) == 0 and l2.index_of(A.INTENSITY) == 1 and l2.index_of(A.CLASSIFICATION) is None


def test_index_of_reordered_doc(lapi):  # :945-949
    l = PointLayout.from_members_and_alignment([A.INTENSITY.at_offset_in_type(24), A.POSITION_3D.at_offset_in_type(0)], 8, api=lapi)
    assert l.index_of(A.INTENSITY) == 0 and l.index_of(A.POSITION_3D) == 1


def test_derive_vs_runtime_layout(lapi):  # point_layout.rs:1058-1070 (TestPoint1: packed Vec3f64, Vec3u16, u16)
    l = PointLayout.from_attributes_packed([A.POSITION_3D, A.COLOR_RGB, A.INTENSITY], 1, api=lapi)
    assert offsets(l) == [("Position3D", 0), ("ColorRGB", 24), ("Intensity", 30)]
    assert l.size_of_point_entry() == 32 and l.alignment() == 1


def test_appendix_a_layouts(lapi):
    from harness import custom_point_type_big, custom_point_type_small
    small = custom_point_type_small(lapi)
    assert offsets(small) == [("Position3D", 0), ("Classification", 24)] and small.size_of_point_entry() == 25
    big = custom_point_type_big(lapi)
    assert offsets(big) == [("GpsTime", 0), ("ColorRGB", 8), ("Position3D", 14), ("Classification", 38), ("Intensity", 39)]
    assert big.size_of_point_entry() == 41
    # "different layout" target of pasture-io/src/las/raw_readers.rs:820-828 (repr(C) rules)
    diff = PointLayout.from_attributes([A.POSITION_3D.with_custom_datatype(T.Vec3f32), A.CLASSIFICATION.with_custom_datatype(T.U32),
                                        A.COLOR_RGB.with_custom_datatype(T.Vec3u8), A.POINT_SOURCE_ID, A.WAVEFORM_PARAMETERS], api=lapi)
    assert offsets(diff) == [("Position3D", 0), ("Classification", 12), ("ColorRGB", 16), ("PointSourceID", 20), ("WaveformParameters", 24)]
    assert diff.size_of_point_entry() == 36 and diff.alignment() == 4


def test_las_layout_sizes(lapi):
    # las_layout.rs:278 (exact binary record sizes) and las_types.rs const_assert sizes :37,:93,:150,:212,:283,:358,:427,:465,:505,:551,:601
    exact = [20, 28, 26, 34, 57, 63, 30, 36, 38, 59, 67]
    typed = [35, 43, 41, 49, 72, 78, 46, 52, 54, 75, 83]
    for f in range(11):
        assert las.point_layout_from_las_point_format(las.Format(f), True, api=lapi).size_of_point_entry() == exact[f]
        assert las.point_layout_from_las_point_format(las.Format(f), False, api=lapi).size_of_point_entry() == typed[f]
    raw0 = las.point_layout_from_las_point_format(las.Format(0), True, api=lapi)
    assert offsets(raw0) == [("LASLocalPosition", 0), ("Intensity", 12), ("LASBasicFlags", 14), ("Classification", 15),
                             ("ScanAngleRank", 16), ("UserData", 17), ("PointSourceID", 18)]
    f0 = las.point_layout_from_las_point_format(las.Format(0), False, api=lapi)
    assert offsets(f0) == [("Position3D", 0), ("Intensity", 24), ("ReturnNumber", 26), ("NumberOfReturns", 27), ("ScanDirectionFlag", 28),
                           ("EdgeOfFlightLine", 29), ("Classification", 30), ("ScanAngleRank", 31), ("UserData", 32), ("PointSourceID", 33)]


def test_layout_panics(lapi):
    l = PointLayout.from_attributes([A.POSITION_3D], api=lapi)
    with pytest.raises(PasturePanic):  # duplicate name :783-788
        l.add_attribute(A.POSITION_3D.with_custom_datatype(T.Vec3f32), FieldAlignment.Default)
    with pytest.raises(PasturePanic):  # non-unique names :725-730
        PointLayout.from_members_and_alignment([A.INTENSITY.at_offset_in_type(0), A.INTENSITY.at_offset_in_type(8)], 8, api=lapi)
    with pytest.raises(PasturePanic):  # overlap :732-745
        PointLayout.from_members_and_alignment([A.INTENSITY.at_offset_in_type(0), A.POSITION_3D.at_offset_in_type(1)], 8, api=lapi)
    with pytest.raises(PasturePanic):  # Packed(3) after a Default attribute => align 3 is not a power of two (Layout::from_size_align)
        l.add_attribute(A.INTENSITY, FieldAlignment.Packed(3))


def test_layout_equality(lapi):
    a = PointLayout.from_attributes([A.POSITION_3D, A.INTENSITY], api=lapi)
    b = PointLayout.from_attributes([A.POSITION_3D, A.INTENSITY], api=lapi)
    c = PointLayout.from_attributes_packed([A.POSITION_3D, A.INTENSITY], 1, api=lapi)  # same members' names, different size/align
    assert a == b and a != c and a.compare_without_offsets(c)
    # runtime Packed(n) quirk: layout alignment = min(n, current) => 1 when built from default()
    assert c.alignment() == 1 and c.size_of_point_entry() == 26


def test_points_of_four_gib_are_a_layout_but_not_a_device_buffer():
    """Host-only part of tests/test_big_records.py (runs without a GPU): ByteArray(2^32) builds a layout with u64 sizes like the reference's
    (point_layout.rs:57), and the product refuses it where it would meet 32-bit strides -- buffers and converters -- before any device is touched."""
    from pasture_amd import product_api
    from pasture_amd._capi import PastureError
    from pasture_amd.buffers import HashMapBuffer, VectorBuffer
    from pasture_amd.conversion import BufferLayoutConverter
    from pasture_amd.layout import PointAttributeDataType as T, PointAttributeDefinition, PointLayout, attributes as A
    api = product_api()
    huge = PointLayout.from_attributes_packed([A.POSITION_3D, PointAttributeDefinition("Blob", T.ByteArray(1 << 32))], 1, api=api)
    assert huge.size_of_point_entry() == (1 << 32) + 24
    just_under = PointLayout.from_attributes_packed([PointAttributeDefinition("Blob", T.ByteArray((1 << 32) - 1))], 1, api=api)
    small = PointLayout.from_attributes([A.POSITION_3D], api=api)
    for make in (lambda: VectorBuffer.new_from_layout(huge), lambda: HashMapBuffer.new_from_layout(huge),
                 lambda: BufferLayoutConverter.for_layouts_with_default(small, huge), lambda: BufferLayoutConverter.for_layouts(huge, small)):
        with pytest.raises(PastureError, match="4 GiB"):
            make()
    assert VectorBuffer.new_from_layout(just_under).len() == 0  # (an empty buffer: nothing is allocated)
