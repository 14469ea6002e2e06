"""ExternalMemoryBuffer over memory that does NOT start on a nice boundary (point_buffer.rs:1479-1708: "for mmap or GPU buffers").
The production shape: a whole LAS file resident in device memory, the point records wrapped where they lie -- behind a 375-byte
header (375 % 16 = 7).  Differential: every result over an odd-based external buffer equals the library's own result over a freshly
allocated (2 MiB-aligned) buffer holding the same bytes, with the run-time compiler in its default mode and in `sync` mode (where the
plan-specialised kernels are at hand and the host has to keep them off bases they do not take)."""
import numpy as np
import pytest

from harness import random_records
from pasture_amd import conversion as cv
from pasture_amd import las
from pasture_amd.algorithms import calculate_bounds, transform_attribute
from pasture_amd.buffers import ExternalMemoryBuffer, HashMapBuffer, VectorBuffer
from pasture_amd.conversion import BufferLayoutConverter, Transform
from pasture_amd.layout import PointAttributeDataType as T, PointLayout, attributes as A

pytestmark = pytest.mark.gpu

SCALE, OFFSET = (0.001, 0.001, 0.001), (500000.0, 5400000.0, 100.0)
OFFSETS = (1, 4, 7, 8, 13)  # bytes in front of the first record (7 = a LAS 1.2 header)


def _layouts(api):
    big = PointLayout.from_attributes_packed([A.GPS_TIME, A.COLOR_RGB, A.POSITION_3D, A.CLASSIFICATION, A.INTENSITY.with_custom_datatype(T.I16)], 1, api=api)
    return {
        "xyz24": PointLayout.from_attributes([A.POSITION_3D], api=api),
        "las0_35": las.point_layout_from_las_point_format(las.Format(0), False, api=api),
        "big41": big,
        "las3_61": las.point_layout_from_las_point_format(las.Format(3), False, api=api),
    }


def _records(layout, n, seed):
    rec = random_records(layout, n, seed)
    rec["Position3D"] = np.random.default_rng(seed).random((n, 3)) * np.array([1000.0, 1000.0, 100.0])
    return rec


def _external(raw_bytes: np.ndarray, off: int, layout):
    """A device tensor with `off` bytes of 0xEE in front of and behind the records; the wrapped buffer and the tensor."""
    import torch
    t = torch.full((off + raw_bytes.size + 32,), 0xEE, dtype=torch.uint8, device="cuda")
    t[off:off + raw_bytes.size] = torch.from_numpy(raw_bytes.reshape(-1)).cuda()
    return ExternalMemoryBuffer(t[off:off + raw_bytes.size], layout), t


def _columns(buf):
    return {a.name(): buf.view_attribute(a.attribute_definition()).tobytes() for a in buf.point_layout().attributes()}


@pytest.mark.parametrize("mode", ["env", "sync"])
@pytest.mark.parametrize("name", ["xyz24", "las0_35", "big41", "las3_61"])
@pytest.mark.parametrize("n", [5, 70_001])
def test_conversions_and_bounds_over_odd_based_external_memory(hip, name, n, mode):
    layout = _layouts(hip)[name]
    S = layout.size_of_point_entry()
    rec = _records(layout, n, 3 + n % 7)
    raw = np.ascontiguousarray(rec).view(np.uint8).reshape(n, S)
    aligned = VectorBuffer.from_numpy(rec, layout)
    conv = BufferLayoutConverter.for_layouts(layout, layout)
    xf = BufferLayoutConverter.for_layouts(layout, layout)
    xf.set_custom_mapping_with_transformation(A.POSITION_3D, A.POSITION_3D, Transform.affine(T.Vec3f64, SCALE, OFFSET), False)
    cv.jit_set_mode(mode, hip)
    try:
        want_cols = _columns(conv.convert(aligned, HashMapBuffer))
        want_b = calculate_bounds(aligned)
        want_recs = VectorBuffer.new_from_layout(layout)
        want_recs.resize(n)
        want_fused = xf.convert_into_with_bounds(aligned, want_recs)
        want_xf = np.ascontiguousarray(want_recs.get_point_range(range(0, n))).tobytes()
        for off in OFFSETS:
            src, t_src = _external(raw, off, layout)
            assert src.len() == n
            # records at an odd base -> columns; AABB; records -> records (+ affine, fused AABB) into another odd base
            assert _columns(conv.convert(src, HashMapBuffer)) == want_cols, (off, "records -> columns")
            assert calculate_bounds(src) == want_b, off
            dst, t_dst = _external(np.zeros((n, S), dtype=np.uint8), (off * 5 + 3) % 16, layout)
            assert xf.convert_into_with_bounds(src, dst) == want_fused, off
            got = np.ascontiguousarray(dst.get_point_range(range(0, n))).tobytes()
            assert got == want_xf, (off, "records -> records")
            o2 = (off * 5 + 3) % 16
            assert bool((t_dst[:o2] == 0xEE).all()) and bool((t_dst[o2 + n * S:] == 0xEE).all()), (off, "bytes around the target")
            # columns -> records at an odd base
            cols = conv.convert(aligned, HashMapBuffer)
            dst2, t_dst2 = _external(np.zeros((n, S), dtype=np.uint8), off, layout)
            conv.convert_into(cols, dst2)
            assert np.ascontiguousarray(dst2.get_point_range(range(0, n))).tobytes() == raw.tobytes(), (off, "columns -> records")
            assert bool((t_dst2[:off] == 0xEE).all()) and bool((t_dst2[off + n * S:] == 0xEE).all()), (off, "bytes around the target")
            # in place on the odd-based records
            transform_attribute(src, A.POSITION_3D, Transform.affine(T.Vec3f64, SCALE, OFFSET))
            assert np.ascontiguousarray(src.get_point_range(range(0, n))).tobytes() == want_xf, (off, "in place")
            assert bool((t_src[:off] == 0xEE).all()) and bool((t_src[off + n * S:] == 0xEE).all()), (off, "bytes around the source")
    finally:
        cv.jit_set_mode("env", hip)


@pytest.mark.parametrize("mode", ["env", "sync"])
@pytest.mark.parametrize("fmt", [0, 1, 3, 6])
def test_las_file_image_in_device_memory_records_behind_the_header(hip, fmt, mode):
    """raw_readers.rs:299-352 on a file image: raw records wrapped behind a 375-byte header (formats 6+: 375 as well here) -> the typed
    layout, columnar and interleaved, with the fused AABB; and the writer's direction: typed points -> raw records at that odd base."""
    n = 50_003
    raw_l = las.point_layout_from_las_point_format(las.Format(fmt), True, api=hip)
    typed = las.point_layout_from_las_point_format(las.Format(fmt), False, api=hip)
    R = raw_l.size_of_point_entry()
    aligned = VectorBuffer.new_from_layout(raw_l)
    aligned.resize(n)
    aligned.synth_fill(5 + fmt, 0)
    raw = np.ascontiguousarray(aligned.get_point_range(range(0, n))).view(np.uint8).reshape(n, R)
    conv = las.get_default_las_converter(raw_l, typed, SCALE, OFFSET)
    cv.jit_set_mode(mode, hip)
    try:
        for kind in (HashMapBuffer, VectorBuffer):
            want = kind.new_from_layout(typed)
            want.resize(n)
            want_b = conv.convert_into_with_bounds(aligned, want)
            src, _ = _external(raw, 375, raw_l)
            out = kind.new_from_layout(typed)
            out.resize(n)
            assert conv.convert_into_with_bounds(src, out) == want_b
            assert _columns(out) == _columns(want), kind.__name__
            # ranged, as the chunked reader calls it (raw_readers.rs:309-349)
            out2 = kind.new_from_layout(typed)
            out2.resize(n)
            for first in range(0, n, 16_384):
                m = min(16_384, n - first)
                conv.convert_into_range(src, range(first, first + m), out2, range(first, first + m))
            assert _columns(out2) == _columns(want), kind.__name__
        typed_pts = conv.convert(aligned, HashMapBuffer)
        back_aligned = VectorBuffer.new_from_layout(raw_l)
        back_aligned.resize(n)
        want_hdr = las.encode_points(typed_pts, fmt, SCALE, OFFSET, back_aligned)
        back, t_back = _external(np.zeros((n, R), dtype=np.uint8), 375, raw_l)
        assert las.encode_points(typed_pts, fmt, SCALE, OFFSET, back) == want_hdr
        assert np.ascontiguousarray(back.get_point_range(range(0, n))).tobytes() == np.ascontiguousarray(back_aligned.get_point_range(range(0, n))).tobytes()
        assert bool((t_back[:375] == 0xEE).all()) and bool((t_back[375 + n * R:] == 0xEE).all())
    finally:
        cv.jit_set_mode("env", hip)


def test_external_memory_the_device_cannot_reach_is_refused_and_mapped_host_memory_works(hip):
    """The reference's ExternalMemoryBuffer wraps ordinary host memory (point_buffer.rs:1479-1497).  Passed to this library as it is, such a pointer would
    fault the GPU at the first kernel (and end the process); pst_buffer_wrap_external[_columns] asks the HIP runtime about both ends of the range
    and refuses what it does not know.  Host memory the device maps (pinned) is taken, and gives the same results as device memory."""
    import torch
    from pasture_amd._capi import PastureError
    from pasture_amd.buffers import ExternalColumnsBuffer
    layout = las.point_layout_from_las_point_format(las.Format(0), False, api=hip)
    n = 20_011
    rec = _records(layout, n, 5)
    raw = np.ascontiguousarray(rec).view(np.uint8).reshape(-1).copy()
    with pytest.raises(PastureError, match="not known to the HIP runtime"):
        ExternalMemoryBuffer(raw.ctypes.data, layout, nbytes=raw.size)  # numpy's (malloc'ed) memory
    xyz = PointLayout.from_attributes([A.POSITION_3D], api=hip)
    col = np.zeros((n, 3))
    with pytest.raises(PastureError, match="not known to the HIP runtime"):
        ExternalColumnsBuffer([col.ctypes.data], xyz, n)
    # pinned host memory: the device maps it
    pinned = torch.from_numpy(raw).pin_memory()
    src = ExternalMemoryBuffer(pinned.data_ptr(), layout, nbytes=raw.size)
    aligned = VectorBuffer.from_numpy(rec, layout)
    conv = BufferLayoutConverter.for_layouts(layout, layout)
    assert _columns(conv.convert(src, HashMapBuffer)) == _columns(conv.convert(aligned, HashMapBuffer))
    assert calculate_bounds(src) == calculate_bounds(aligned)
    out_host = torch.zeros(raw.size, dtype=torch.uint8).pin_memory()
    dst = ExternalMemoryBuffer(out_host.data_ptr(), layout, nbytes=raw.size)
    conv.convert_into(conv.convert(aligned, HashMapBuffer), dst)
    torch.cuda.synchronize()
    assert out_host.numpy().tobytes() == raw.tobytes()  # the records arrived in host memory, written by the kernels
    dst.swap(0, n - 1)  # (copies inside mapped host memory take the right copy kind)
    torch.cuda.synchronize()
    got = out_host.numpy().reshape(n, 35)
    assert got[0].tobytes() == raw.reshape(n, 35)[n - 1].tobytes() and got[n - 1].tobytes() == raw.reshape(n, 35)[0].tobytes()
