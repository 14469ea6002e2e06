"""The production caller's contract for the hot path: LAS point layouts and `get_default_las_converter`.

Reference: pasture-io/src/las/las_layout.rs:37-125 (layouts), las_types.rs:10-601 (LasPointFormat0..10),
raw_readers.rs:31-167 (converter construction).  File parsing (headers, VLRs, LAZ) is out of scope; a tiny reader
for uncompressed point records exists only so the reference's fixtures can be used as golden vectors.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from .conversion import BufferLayoutConverter, Transform
from .layout import FieldAlignment, PointAttributeDataType as T, PointAttributeDefinition, PointLayout, attributes as A

# las_layout.rs:37-49
ATTRIBUTE_BASIC_FLAGS = PointAttributeDefinition.custom("LASBasicFlags", T.U8)
ATTRIBUTE_EXTENDED_FLAGS = PointAttributeDefinition.custom("LASExtendedFlags", T.U16)
ATTRIBUTE_LOCAL_LAS_POSITION = PointAttributeDefinition.custom("LASLocalPosition", T.Vec3i32)


@dataclass(frozen=True)
class Format:
    """las::point::Format (crate `las`, not vendored) restricted to what las_layout.rs reads from it."""
    number: int

    @property
    def is_extended(self):
        return self.number >= 6

    @property
    def has_gps_time(self):
        return self.number in (1, 3, 4, 5) or self.number >= 6

    @property
    def has_color(self):
        return self.number in (2, 3, 5, 7, 8, 10)

    @property
    def has_nir(self):
        return self.number in (8, 10)

    @property
    def has_waveform(self):
        return self.number in (4, 5, 9, 10)


def point_layout_from_las_point_format(fmt: Format, exact_binary_representation: bool, api=None) -> PointLayout:
    """las_layout.rs:64-125."""
    if not 0 <= fmt.number <= 10:
        raise ValueError(f"Unsupported LAS point format {fmt.number}")
    P1 = FieldAlignment.Packed(1)
    layout = PointLayout.default(api)
    if exact_binary_representation:  # :70-107
        layout.add_attribute(ATTRIBUTE_LOCAL_LAS_POSITION, P1)
        layout.add_attribute(A.INTENSITY, P1)
        layout.add_attribute(ATTRIBUTE_EXTENDED_FLAGS if fmt.is_extended else ATTRIBUTE_BASIC_FLAGS, P1)
        layout.add_attribute(A.CLASSIFICATION, P1)
        if fmt.is_extended:
            layout.add_attribute(A.USER_DATA, P1)
            layout.add_attribute(A.SCAN_ANGLE, P1)
        else:
            layout.add_attribute(A.SCAN_ANGLE_RANK, P1)
            layout.add_attribute(A.USER_DATA, P1)
        layout.add_attribute(A.POINT_SOURCE_ID, P1)
    else:  # LasPointFormatN::layout(): repr(C, packed) structs, las_types.rs
        layout.add_attribute(A.POSITION_3D, P1)
        layout.add_attribute(A.INTENSITY, P1)
        layout.add_attribute(A.RETURN_NUMBER, P1)
        layout.add_attribute(A.NUMBER_OF_RETURNS, P1)
        if fmt.is_extended:
            layout.add_attribute(A.CLASSIFICATION_FLAGS, P1)
            layout.add_attribute(A.SCANNER_CHANNEL, P1)
        layout.add_attribute(A.SCAN_DIRECTION_FLAG, P1)
        layout.add_attribute(A.EDGE_OF_FLIGHT_LINE, P1)
        layout.add_attribute(A.CLASSIFICATION, P1)
        if fmt.is_extended:
            layout.add_attribute(A.USER_DATA, P1)
            layout.add_attribute(A.SCAN_ANGLE, P1)
        else:
            layout.add_attribute(A.SCAN_ANGLE_RANK, P1)
            layout.add_attribute(A.USER_DATA, P1)
        layout.add_attribute(A.POINT_SOURCE_ID, P1)
    if fmt.has_gps_time:
        layout.add_attribute(A.GPS_TIME, P1)
    if fmt.has_color:
        layout.add_attribute(A.COLOR_RGB, P1)
    if fmt.has_nir:
        layout.add_attribute(A.NIR, P1)
    if fmt.has_waveform:
        layout.add_attribute(A.WAVE_PACKET_DESCRIPTOR_INDEX, P1)
        layout.add_attribute(A.WAVEFORM_DATA_OFFSET, P1)
        layout.add_attribute(A.WAVEFORM_PACKET_SIZE, P1)
        layout.add_attribute(A.RETURN_POINT_WAVEFORM_LOCATION, P1)
        layout.add_attribute(A.WAVEFORM_PARAMETERS, P1)
    return layout


def get_default_las_converter(raw_las_layout: PointLayout, target_layout: PointLayout, scale: Tuple[float, float, float],
                              offset: Tuple[float, float, float]) -> BufferLayoutConverter:
    """raw_readers.rs:31-167 with `las_header.transforms()` passed as (scale, offset)."""
    converter = BufferLayoutConverter.for_layouts_with_default(raw_las_layout, target_layout)  # :36-37
    pos = target_layout.get_attribute_by_name(A.POSITION_3D.name())
    if pos is not None:  # :39-58
        if pos.datatype() == T.Vec3f64:
            converter.set_custom_mapping_with_transformation(ATTRIBUTE_LOCAL_LAS_POSITION, pos.attribute_definition(),
                                                             Transform.affine(T.Vec3f64, scale, offset), False)
        elif pos.datatype() == T.Vec3f32:
            converter.set_custom_mapping_with_transformation(ATTRIBUTE_LOCAL_LAS_POSITION, pos.attribute_definition(),
                                                             Transform.affine(T.Vec3f32, scale, offset), False)
        else:
            raise ValueError(f"Invalid datatype {pos.datatype()} for POSITION_3D attribute. Only Vec3f64 and Vec3f32 are supported!")

    def bits(flags_attr, target_name, shift, mask):
        t = target_layout.get_attribute_by_name(target_name)
        if t is not None:
            converter.set_custom_mapping_with_transformation(flags_attr, t.attribute_definition(),
                                                             Transform.bitfield(flags_attr.datatype(), shift, mask), True)

    if raw_las_layout.has_attribute(ATTRIBUTE_BASIC_FLAGS):  # :61-103
        bits(ATTRIBUTE_BASIC_FLAGS, A.RETURN_NUMBER.name(), 0, 0b111)
        bits(ATTRIBUTE_BASIC_FLAGS, A.NUMBER_OF_RETURNS.name(), 3, 0b111)
        bits(ATTRIBUTE_BASIC_FLAGS, A.SCAN_DIRECTION_FLAG.name(), 6, 0b1)
        bits(ATTRIBUTE_BASIC_FLAGS, A.EDGE_OF_FLIGHT_LINE.name(), 7, 0b1)
    else:  # :104-164
        bits(ATTRIBUTE_EXTENDED_FLAGS, A.RETURN_NUMBER.name(), 0, 0b1111)
        bits(ATTRIBUTE_EXTENDED_FLAGS, A.NUMBER_OF_RETURNS.name(), 4, 0b1111)
        bits(ATTRIBUTE_EXTENDED_FLAGS, A.CLASSIFICATION_FLAGS.name(), 8, 0b1111)
        bits(ATTRIBUTE_EXTENDED_FLAGS, A.SCANNER_CHANNEL.name(), 12, 0b11)
        bits(ATTRIBUTE_EXTENDED_FLAGS, A.SCAN_DIRECTION_FLAG.name(), 14, 0b1)
        bits(ATTRIBUTE_EXTENDED_FLAGS, A.EDGE_OF_FLIGHT_LINE.name(), 15, 0b1)
    return converter


def encode_points(points, point_format: int, scale, offset, target, target_first: int = 0, header_bounds=None, max_return: int = 15):
    """RawLASWriter::write_points_default_layout (pasture-io/src/las/raw_writers.rs:203-363): typed LAS points in the format's
    default layout -> exact-binary records in `target[target_first:]`.  Returns (bounds, points_by_return): the header's
    updated {min xyz, max xyz} (starting from `header_bounds`, default f64::MAX / f64::MIN like the writer) and the number of
    points per return number 1..max_return."""
    import ctypes as C
    b = (C.c_double * 6)(*(header_bounds if header_bounds is not None else [1.7976931348623157e308] * 3 + [-1.7976931348623157e308] * 3))
    counts = (C.c_uint64 * 15)()
    sc, of = (C.c_double * 3)(*scale), (C.c_double * 3)(*offset)
    points.api.las_encode_points(points._h, point_format, sc, of, target._h, target_first, b, counts, max_return)
    return (tuple(b[:3]), tuple(b[3:])), list(counts)[:max_return]


def write_points(points, point_format: int, scale, offset, target, target_first: int = 0, header_bounds=None, max_return: int = 15):
    """RawLASWriter::write (raw_writers.rs:606-613): the format's default layout takes `encode_points` directly; any other
    layout takes write_points_custom_layout (:365-603), whose per-attribute readers (read_helpers.rs:302-345: same datatype ->
    copy, other datatype -> the `as` converter, attribute missing -> Default::default()) are exactly a
    `BufferLayoutConverter::for_layouts_with_default(source layout, default layout)`; the records are then encoded from the
    converted (device-resident, columnar) points.  Reference quirk kept: the custom-layout writer never increments its
    points-by-return map (:379-387 vs :256-258), so the counts it adds to the header are all zero."""
    from .buffers import HashMapBuffer
    default_layout = point_layout_from_las_point_format(Format(point_format), False, api=points.api)
    if points.point_layout() == default_layout:
        return encode_points(points, point_format, scale, offset, target, target_first, header_bounds, max_return)
    converter = BufferLayoutConverter.for_layouts_with_default(points.point_layout(), default_layout)
    staged = converter.convert(points, HashMapBuffer)
    bounds, counts = encode_points(staged, point_format, scale, offset, target, target_first, header_bounds, max_return)
    return bounds, [0] * len(counts)


def read_records_into(records, point_format: int, scale, offset, target, chunk_points: int = 4 << 20):
    """Host side of a device LAS reader (RawLASReader::read_into_custom_layout, raw_readers.rs:299-352, with the chunk loop of
    :266-289): raw point records in HOST memory -> `target` (typed layout, device buffer), in chunks that are copied over PCIe on
    a second stream while the previous chunk is being decoded.  `records`: a 1-D uint8 torch tensor on the CPU (pinned memory
    gives the full link rate; pageable tensors are staged through pinned buffers first) whose length is a multiple of the
    record size.  Returns the number of points read.  torch is used for streams / events / pinned memory only."""
    import ctypes as C

    import torch
    from .buffers import ExternalMemoryBuffer
    api = target.api
    raw_layout = point_layout_from_las_point_format(Format(point_format), True, api=api)
    rs = raw_layout.size_of_point_entry()
    assert records.dtype == torch.uint8 and records.dim() == 1 and records.numel() % rs == 0
    n = records.numel() // rs
    if target.len() < n:
        raise ValueError("point_buffer.len() must be >= count")  # raw_readers.rs:375-377
    converter = get_default_las_converter(raw_layout, target.point_layout(), scale, offset)
    compute = torch.cuda.current_stream()
    copier = torch.cuda.Stream()
    chunk = max(1, min(chunk_points, n))
    staging = [torch.empty(chunk * rs, dtype=torch.uint8, device="cuda") for _ in range(2)]
    views = [ExternalMemoryBuffer(t, raw_layout) for t in staging]
    bounce = None if records.is_pinned() else [torch.empty(chunk * rs, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
    copied = [torch.cuda.Event() for _ in range(2)]
    decoded = [torch.cuda.Event() for _ in range(2)]
    # the staging tensors were allocated on the compute stream: torch's allocator may hand out blocks that queued compute-stream work
    # is still using, so the copy stream starts behind the compute stream and the blocks are marked as used by both
    copier.wait_stream(compute)
    for t in staging:
        t.record_stream(copier)
    prev_stream = C.c_void_p()
    api.get_stream(C.byref(prev_stream))
    api.set_stream(C.c_void_p(compute.cuda_stream))
    try:
        return _read_chunks(api, records, rs, n, chunk, staging, views, bounce, copied, decoded, compute, copier, converter, target)
    finally:
        api.set_stream(prev_stream)  # the caller's pst stream is left as it was


def _read_chunks(api, records, rs, n, chunk, staging, views, bounce, copied, decoded, compute, copier, converter, target):
    import torch
    for c, first in enumerate(range(0, n, chunk)):
        b = c & 1
        cnt = min(chunk, n - first)
        src = records[first * rs:(first + cnt) * rs]
        if c >= 2:
            decoded[b].synchronize() if bounce is not None else copier.wait_event(decoded[b])  # staging[b] is free again
        if bounce is not None:
            bounce[b][:cnt * rs].copy_(src)
            src = bounce[b][:cnt * rs]
        with torch.cuda.stream(copier):
            staging[b][:cnt * rs].copy_(src, non_blocking=True)
            copied[b].record(copier)
        compute.wait_event(copied[b])
        converter.convert_into_range_async(views[b], range(0, cnt), target, range(first, first + cnt))
        decoded[b].record(compute)
    compute.synchronize()
    return n


def write_records_from(points, point_format: int, scale, offset, records_out, header_bounds=None, max_return: int = 15,
                       chunk_points: int = 4 << 20):
    """Host side of a device LAS writer (RawLASWriter::write_points_default_layout, raw_writers.rs:203-363, whose own loop works in
    chunks of 50 000 points): typed points on the device (the format's default layout) -> raw records in HOST memory
    (`records_out`: 1-D uint8 CPU tensor, pinned for the full link rate, n * record size bytes).  Chunk i is encoded on the
    compute stream while chunk i-1 crosses PCIe on a second stream.  Returns (bounds, points_by_return) like `encode_points`;
    raises the writer's panic if any position does not fit an i32."""
    import ctypes as C

    import torch
    from ._capi import ERR_RANGE, PasturePanic
    from .buffers import ExternalMemoryBuffer
    api = points.api
    raw_layout = point_layout_from_las_point_format(Format(point_format), True, api=api)
    rs = raw_layout.size_of_point_entry()
    n = points.len()
    assert records_out.dtype == torch.uint8 and records_out.dim() == 1 and records_out.numel() >= n * rs
    compute = torch.cuda.current_stream()
    copier = torch.cuda.Stream()
    chunk = max(1, min(chunk_points, max(n, 1)))
    n_chunks = (n + chunk - 1) // chunk
    staging = [torch.empty(chunk * rs, dtype=torch.uint8, device="cuda") for _ in range(2)]
    views = [ExternalMemoryBuffer(t, raw_layout) for t in staging]
    bounce = None if records_out.is_pinned() else [torch.empty(chunk * rs, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
    dev_bounds = torch.empty(max(n_chunks, 1), 6, dtype=torch.float64, device="cuda")
    dev_counts = torch.zeros(max(n_chunks, 1), 16, dtype=torch.int64, device="cuda")
    encoded = [torch.cuda.Event() for _ in range(2)]
    copied = [torch.cuda.Event() for _ in range(2)]
    sc, of = (C.c_double * 3)(*scale), (C.c_double * 3)(*offset)
    copier.wait_stream(compute)  # staging was allocated on the compute stream (see read_records_into)
    for t in staging:
        t.record_stream(copier)
    prev_stream = C.c_void_p()
    api.get_stream(C.byref(prev_stream))
    api.set_stream(C.c_void_p(compute.cuda_stream))
    pending = [None, None]  # (bounce buffer, destination slice) whose D2H copy is in flight
    try:
        _write_chunks(api, points, point_format, sc, of, max_return, records_out, rs, n, chunk, n_chunks, staging, views, bounce, dev_bounds, dev_counts,
                      encoded, copied, compute, copier, pending)
    finally:
        api.set_stream(prev_stream)  # the caller's pst stream is left as it was
    copier.synchronize()
    compute.synchronize()
    for b in range(2):
        if pending[b] is not None:
            pending[b][1].copy_(pending[b][0][:pending[b][1].numel()])
    return _fold_header(header_bounds, max_return, n_chunks, dev_bounds, dev_counts)


def _write_chunks(api, points, point_format, sc, of, max_return, records_out, rs, n, chunk, n_chunks, staging, views, bounce, dev_bounds, dev_counts, encoded,
                  copied, compute, copier, pending):
    import ctypes as C

    import torch
    for c in range(n_chunks):
        b = c & 1
        first = c * chunk
        cnt = min(chunk, n - first)
        if c >= 2:
            compute.wait_event(copied[b])  # staging[b] has left the device
        api.las_encode_range_async(points._h, first, cnt, point_format, sc, of, views[b]._h, 0, C.c_void_p(dev_bounds[c].data_ptr()),
                                   C.c_void_p(dev_counts[c].data_ptr()), max_return)
        encoded[b].record(compute)
        if pending[b] is not None:  # pageable destination: finish the previous use of this bounce buffer on the host
            copied[b].synchronize()
            pending[b][1].copy_(pending[b][0][:pending[b][1].numel()])
            pending[b] = None
        with torch.cuda.stream(copier):
            copier.wait_event(encoded[b])
            dst = records_out[first * rs:(first + cnt) * rs]
            if bounce is None:
                dst.copy_(staging[b][:cnt * rs], non_blocking=True)
            else:
                bounce[b][:cnt * rs].copy_(staging[b][:cnt * rs], non_blocking=True)
                pending[b] = (bounce[b], dst)
            copied[b].record(copier)


def _fold_header(header_bounds, max_return, n_chunks, dev_bounds, dev_counts):
    from ._capi import ERR_RANGE, PasturePanic
    hb = list(header_bounds) if header_bounds is not None else [1.7976931348623157e308] * 3 + [-1.7976931348623157e308] * 3
    counts = [0] * max_return
    if n_chunks:
        cb, cc = dev_bounds[:n_chunks].cpu(), dev_counts[:n_chunks].cpu()
        bad = int(cc[:, 0].sum())
        if bad:
            raise PasturePanic(ERR_RANGE, f"write_position_as_las_position: Position is out of bounds given the current LAS offset and scale! ({bad} positions)")
        mn, mx = cb[:, :3].min(dim=0).values.tolist(), cb[:, 3:].max(dim=0).values.tolist()
        hb = [min(hb[i], mn[i]) for i in range(3)] + [max(hb[3 + i], mx[i]) for i in range(3)]
        counts = [int(x) for x in cc[:, 1:max_return + 1].sum(dim=0).tolist()]
    return (tuple(hb[:3]), tuple(hb[3:])), counts


@dataclass
class LasFile:
    """Just enough of an uncompressed .las file to use the reference's fixtures as golden vectors."""
    point_format: int
    record_length: int
    num_points: int
    scale: Tuple[float, float, float]
    offset: Tuple[float, float, float]
    records: np.ndarray  # (num_points, record_length) uint8
    extra_bytes_attributes: tuple = ()  # attribute definitions of the Extra Bytes VLR


# ExtraBytesDataType -> PointAttributeDataType, las_metadata.rs:289-315 (LAS 1.4 Extra Bytes VLR data_type codes 1..10)
_EXTRA_BYTES_TYPES = {1: T.U8, 2: T.I8, 3: T.U16, 4: T.I16, 5: T.U32, 6: T.I32, 7: T.U64, 8: T.I64, 9: T.F32, 10: T.F64}


def parse_extra_bytes_vlr(raw: bytes, header_size: int, n_vlrs: int):
    """Attribute definitions of the Extra Bytes VLR (user id "LASF_Spec", record id 4; 192-byte entries: data_type at byte 2,
    name at bytes 4..36) — ExtraBytesEntry::get_point_attribute, las_metadata.rs:508-515."""
    pos = header_size
    attrs = []
    for _ in range(n_vlrs):
        user_id = raw[pos + 2:pos + 18].split(b"\0")[0]
        record_id, length = struct.unpack_from("<HH", raw, pos + 18)
        body = raw[pos + 54:pos + 54 + length]
        if user_id == b"LASF_Spec" and record_id == 4:
            for off in range(0, len(body) - 191, 192):
                code = body[off + 2]
                if code not in _EXTRA_BYTES_TYPES:
                    raise ValueError("Extra bytes of type 'undocumented' / 'deprecated' / 'reserved' are currently unsupported in pasture")
                name = body[off + 4:off + 36].split(b"\0")[0].decode("utf-8")
                attrs.append(PointAttributeDefinition.custom(name, _EXTRA_BYTES_TYPES[code]))
        pos += 54 + length
    return attrs


def point_layout_from_las_metadata(fmt: Format, num_extra_bytes: int, extra_byte_attributes, exact_binary_representation: bool,
                                   api=None) -> PointLayout:
    """las_layout.rs:134-185: the format's layout plus the Extra Bytes VLR attributes (Packed(1)); bytes the VLR does not
    describe become one `UndescribedExtraBytes` byte array whose length is — as in the reference (:171-181, sic) — the number
    of DESCRIBED bytes."""
    layout = point_layout_from_las_point_format(fmt, exact_binary_representation, api=api)
    if num_extra_bytes == 0:
        return layout
    described = 0
    for a in extra_byte_attributes:
        layout.add_attribute(a, FieldAlignment.Packed(1))
        described += a.datatype().size()
    if num_extra_bytes - described > 0:
        layout.add_attribute(PointAttributeDefinition.custom("UndescribedExtraBytes", T.ByteArray(described)), FieldAlignment.Packed(1))
    return layout


def read_las_records(path: str) -> LasFile:
    """LAS 1.2-1.4 public header block (ASPRS LAS specification): offsets 94/96/100/104/105/107/131/155."""
    raw = open(path, "rb").read()
    assert raw[:4] == b"LASF"
    minor = raw[25]
    header_size = struct.unpack_from("<H", raw, 94)[0]
    n_vlrs = struct.unpack_from("<I", raw, 100)[0]
    offset_to_points = struct.unpack_from("<I", raw, 96)[0]
    fmt = raw[104] & 0x3F
    rec_len = struct.unpack_from("<H", raw, 105)[0]
    n_legacy = struct.unpack_from("<I", raw, 107)[0]
    n = n_legacy
    if minor >= 4 and n_legacy == 0:
        n = struct.unpack_from("<Q", raw, 247)[0]
    sx, sy, sz, ox, oy, oz = struct.unpack_from("<6d", raw, 131)
    recs = np.frombuffer(raw, dtype=np.uint8, count=n * rec_len, offset=offset_to_points).reshape(n, rec_len).copy()
    f = LasFile(fmt, rec_len, n, (sx, sy, sz), (ox, oy, oz), recs)
    f.extra_bytes_attributes = parse_extra_bytes_vlr(raw, header_size, n_vlrs)
    return f
