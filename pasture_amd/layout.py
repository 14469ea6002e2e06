"""PointLayout and friends — Python mirror of pasture-core/src/layout/point_layout.rs over the C ABI.

Same names, argument meaning and error behaviour as the reference (panics surface as `PasturePanic`):
  PointAttributeDataType (:23-127), PointAttributeDefinition (:261-341), PointAttributeMember (:353-431),
  attributes::* (:454-598), FieldAlignment (:600-610), PointLayout (:648-997).
All offset / size / alignment arithmetic happens in the C++ library; this file only marshals.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Iterable, List, Optional

import numpy as np

from . import _capi
from ._capi import CApi, DataTypeStruct, MemberStruct


@dataclass(frozen=True)
class PointAttributeDataType:
    kind: int
    size_param: int = 0
    align_param: int = 0
    uuid: bytes = b"\0" * 16

    _NAMES = ("U8", "I8", "U16", "I16", "U32", "I32", "U64", "I64", "F32", "F64", "Vec3u8", "Vec3u16", "Vec3f32",
              "Vec3i32", "Vec3f64", "Vec4u8", "ByteArray", "Custom")
    _SIZES = (1, 1, 2, 2, 4, 4, 8, 8, 4, 8, 3, 6, 12, 12, 24, 4)
    _ALIGNS = (1, 1, 2, 2, 4, 4, 8, 8, 4, 8, 1, 2, 4, 4, 8, 1)
    # numpy component dtype + component count for the typed host views
    _NP = ("u1", "i1", "<u2", "<i2", "<u4", "<i4", "<u8", "<i8", "<f4", "<f8", "u1", "<u2", "<f4", "<i4", "<f8", "u1")
    _NCOMP = (1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 3, 3, 3, 3, 3, 4)

    @staticmethod
    def ByteArray(length: int) -> "PointAttributeDataType":
        return PointAttributeDataType(16, size_param=length)

    @staticmethod
    def Custom(size: int, min_alignment: int, name: bytes) -> "PointAttributeDataType":
        assert len(name) == 16
        return PointAttributeDataType(17, size_param=size, align_param=min_alignment, uuid=bytes(name))

    def size(self) -> int:  # :72-95
        return self._SIZES[self.kind] if self.kind < 16 else self.size_param

    def min_alignment(self) -> int:  # :98-126
        if self.kind < 16:
            return self._ALIGNS[self.kind]
        return 1 if self.kind == 16 else self.align_param

    def numpy_dtype(self) -> np.dtype:
        return np.dtype(self._NP[self.kind]) if self.kind < 16 else np.dtype("u1")

    def num_components(self) -> int:
        return self._NCOMP[self.kind] if self.kind < 16 else self.size()

    def is_float(self) -> bool:
        return self.kind in (8, 9, 12, 14)

    def to_c(self) -> DataTypeStruct:
        s = DataTypeStruct()
        s.kind = self.kind
        s.size_param = self.size_param
        s.align_param = self.align_param
        s.uuid[:] = list(self.uuid)
        return s

    @staticmethod
    def from_c(s: DataTypeStruct) -> "PointAttributeDataType":
        if s.kind < 16:
            return PointAttributeDataType(s.kind)
        if s.kind == 16:
            return PointAttributeDataType(16, size_param=s.size_param)
        return PointAttributeDataType(17, s.size_param, s.align_param, bytes(s.uuid))

    def __str__(self) -> str:
        return self._NAMES[self.kind]


for _i, _n in enumerate(PointAttributeDataType._NAMES[:16]):
    setattr(PointAttributeDataType, _n, PointAttributeDataType(_i))


@dataclass(frozen=True)
class PointAttributeDefinition:
    _name: str
    _datatype: PointAttributeDataType

    @staticmethod
    def custom(name: str, datatype: PointAttributeDataType) -> "PointAttributeDefinition":  # :279-281
        return PointAttributeDefinition(name, datatype)

    def name(self) -> str:
        return self._name

    def datatype(self) -> PointAttributeDataType:
        return self._datatype

    def size(self) -> int:
        return self._datatype.size()

    def with_custom_datatype(self, new_datatype: PointAttributeDataType) -> "PointAttributeDefinition":  # :317-322
        return PointAttributeDefinition(self._name, new_datatype)

    def at_offset_in_type(self, offset: int) -> "PointAttributeMember":  # :327-333
        return PointAttributeMember(self, offset, self.size())

    def __str__(self) -> str:
        return f"[{self._name};{self._datatype}]"


@dataclass(frozen=True)
class PointAttributeMember:
    _definition: PointAttributeDefinition
    _offset: int
    _size: int

    @staticmethod
    def custom(name: str, datatype: PointAttributeDataType, offset: int) -> "PointAttributeMember":  # :374-384
        return PointAttributeMember(PointAttributeDefinition(name, datatype), offset, datatype.size())

    def name(self) -> str:
        return self._definition.name()

    def datatype(self) -> PointAttributeDataType:
        return self._definition.datatype()

    def offset(self) -> int:
        return self._offset

    def size(self) -> int:
        return self._size

    def attribute_definition(self) -> PointAttributeDefinition:
        return self._definition

    def byte_range_within_point(self) -> range:
        return range(self._offset, self._offset + self._size)


class attributes:
    """Built-in attribute definitions, point_layout.rs:454-598."""
    _T = PointAttributeDataType
    POSITION_3D = PointAttributeDefinition("Position3D", _T.Vec3f64)
    INTENSITY = PointAttributeDefinition("Intensity", _T.U16)
    RETURN_NUMBER = PointAttributeDefinition("ReturnNumber", _T.U8)
    NUMBER_OF_RETURNS = PointAttributeDefinition("NumberOfReturns", _T.U8)
    CLASSIFICATION_FLAGS = PointAttributeDefinition("ClassificationFlags", _T.U8)
    SCANNER_CHANNEL = PointAttributeDefinition("ScannerChannel", _T.U8)
    SCAN_DIRECTION_FLAG = PointAttributeDefinition("ScanDirectionFlag", _T.U8)
    EDGE_OF_FLIGHT_LINE = PointAttributeDefinition("EdgeOfFlightLine", _T.U8)
    CLASSIFICATION = PointAttributeDefinition("Classification", _T.U8)
    SCAN_ANGLE_RANK = PointAttributeDefinition("ScanAngleRank", _T.I8)
    SCAN_ANGLE = PointAttributeDefinition("ScanAngle", _T.I16)
    USER_DATA = PointAttributeDefinition("UserData", _T.U8)
    POINT_SOURCE_ID = PointAttributeDefinition("PointSourceID", _T.U16)
    COLOR_RGB = PointAttributeDefinition("ColorRGB", _T.Vec3u16)
    GPS_TIME = PointAttributeDefinition("GpsTime", _T.F64)
    NIR = PointAttributeDefinition("NIR", _T.U16)
    WAVE_PACKET_DESCRIPTOR_INDEX = PointAttributeDefinition("WavePacketDescriptorIndex", _T.U8)
    WAVEFORM_DATA_OFFSET = PointAttributeDefinition("WaveformDataOffset", _T.U64)
    WAVEFORM_PACKET_SIZE = PointAttributeDefinition("WaveformPacketSize", _T.U32)
    RETURN_POINT_WAVEFORM_LOCATION = PointAttributeDefinition("ReturnPointWaveformLocation", _T.F32)
    WAVEFORM_PARAMETERS = PointAttributeDefinition("WaveformParameters", _T.Vec3f32)
    POINT_ID = PointAttributeDefinition("PointID", _T.U64)
    NORMAL = PointAttributeDefinition("Normal", _T.Vec3f32)


@dataclass(frozen=True)
class FieldAlignment:
    """FieldAlignment::{Default, Packed(max_alignment)}, point_layout.rs:600-610."""
    packed: bool = False
    max_alignment: int = 0

    @staticmethod
    def Packed(max_alignment: int) -> "FieldAlignment":
        return FieldAlignment(True, max_alignment)


FieldAlignment.Default = FieldAlignment(False, 0)


class PointLayout:
    """PointLayout, point_layout.rs:648-997.  Wraps a `pst_layout*` (or an oracle layout when `api` is given)."""

    def __init__(self, api: Optional[CApi] = None, _handle: Optional[int] = None):
        self.api = api or _capi.product_api()
        if _handle is None:
            h = C.c_void_p()
            self.api.layout_create(C.byref(h))  # PointLayout::default() :1011-1023
            _handle = h.value
        self._h = C.c_void_p(_handle)
        self._members: Optional[List[PointAttributeMember]] = None

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                self.api.layout_destroy(self._h)
        except Exception:
            pass

    # -- constructors -----------------------------------------------------------------------------------
    @classmethod
    def default(cls, api: Optional[CApi] = None) -> "PointLayout":
        return cls(api)

    @classmethod
    def from_attributes(cls, attrs: Iterable[PointAttributeDefinition], api: Optional[CApi] = None) -> "PointLayout":  # :667-669
        layout = cls(api)
        for a in attrs:
            layout.add_attribute(a, FieldAlignment.Default)
        return layout

    @classmethod
    def from_attributes_packed(cls, attrs: Iterable[PointAttributeDefinition], max_alignment: int,
                               api: Optional[CApi] = None) -> "PointLayout":  # :693-702
        layout = cls(api)
        for a in attrs:
            layout.add_attribute(a, FieldAlignment.Packed(max_alignment))
        return layout

    @classmethod
    def from_members_and_alignment(cls, members: Iterable[PointAttributeMember], type_alignment: int,
                                   api: Optional[CApi] = None) -> "PointLayout":  # :719-759
        api = api or _capi.product_api()
        members = list(members)
        arr = (MemberStruct * max(1, len(members)))()
        keep = []
        for i, m in enumerate(members):
            nm = m.name().encode()
            keep.append(nm)
            arr[i].name = nm
            arr[i].datatype = m.datatype().to_c()
            arr[i].offset = m.offset()
            arr[i].size = m.size()
        h = C.c_void_p()
        api.layout_from_members(arr, len(members), type_alignment, C.byref(h))
        return cls(api, h.value)

    def clone(self) -> "PointLayout":
        h = C.c_void_p()
        self.api.layout_clone(self._h, C.byref(h))
        return PointLayout(self.api, h.value)

    # -- mutation ---------------------------------------------------------------------------------------
    def add_attribute(self, attribute: PointAttributeDefinition, field_alignment: FieldAlignment = FieldAlignment.Default) -> None:  # :778-822
        dt = attribute.datatype().to_c()
        self.api.layout_add_attribute(self._h, attribute.name().encode(), C.byref(dt), 1 if field_alignment.packed else 0,
                                      field_alignment.max_alignment)
        self._members = None

    # -- queries ----------------------------------------------------------------------------------------
    def attributes(self) -> List[PointAttributeMember]:  # :902-904
        if self._members is None:
            n = C.c_size_t()
            self.api.layout_num_attributes(self._h, C.byref(n))
            out = []
            for i in range(n.value):
                m = MemberStruct()
                self.api.layout_get_member(self._h, i, C.byref(m))
                d = PointAttributeDefinition(m.name.decode(), PointAttributeDataType.from_c(m.datatype))
                out.append(PointAttributeMember(d, m.offset, m.size))
            self._members = out
        return self._members

    def at(self, index: int) -> PointAttributeMember:  # :898-900
        return self.attributes()[index]

    def has_attribute_with_name(self, name: str) -> bool:  # :834-838
        return any(a.name() == name for a in self.attributes())

    def has_attribute(self, attribute: PointAttributeDefinition) -> bool:  # :861-866
        return self.get_attribute(attribute) is not None

    def get_attribute(self, attribute: PointAttributeDefinition) -> Optional[PointAttributeMember]:  # :882-890
        for a in self.attributes():
            if a.name() == attribute.name() and a.datatype() == attribute.datatype():
                return a
        return None

    def get_attribute_by_name(self, name: str) -> Optional[PointAttributeMember]:  # :892-896
        for a in self.attributes():
            if a.name() == name:
                return a
        return None

    def size_of_point_entry(self) -> int:  # :928-931
        v = C.c_uint64()
        self.api.layout_size_of_point_entry(self._h, C.byref(v))
        return v.value

    def alignment(self) -> int:
        v = C.c_uint64()
        self.api.layout_alignment(self._h, C.byref(v))
        return v.value

    def index_of(self, attribute: PointAttributeDefinition) -> Optional[int]:  # :950-955
        for i, a in enumerate(self.attributes()):
            if a.name() == attribute.name() and a.datatype() == attribute.datatype():
                return i
        return None

    def offset_of(self, attribute: PointAttributeDefinition) -> Optional[int]:  # :974-983
        m = self.get_attribute(attribute)
        return None if m is None else m.offset()

    def compare_without_offsets(self, other: "PointLayout") -> bool:  # :957-972
        if len(self.attributes()) != len(other.attributes()):
            return False
        for a in self.attributes():
            o = other.get_attribute_by_name(a.name())
            if o is None or o.datatype() != a.datatype():
                return False
        return True

    def __eq__(self, other) -> bool:  # derive(PartialEq) :646
        if not isinstance(other, PointLayout):
            return NotImplemented
        if other.api is not self.api:
            return (self.attributes() == other.attributes() and self.size_of_point_entry() == other.size_of_point_entry()
                    and self.alignment() == other.alignment())
        r = C.c_int()
        self.api.layout_equals(self._h, other._h, C.byref(r))
        return bool(r.value)

    def __repr__(self) -> str:
        body = ", ".join(f"{a.name()}:{a.datatype()}@{a.offset()}" for a in self.attributes())
        return f"PointLayout{{{body}; size={self.size_of_point_entry()} align={self.alignment()}}}"

    def numpy_record_dtype(self) -> np.dtype:
        """Structured dtype with the exact offsets of this layout (itemsize = size_of_point_entry)."""
        names, formats, offsets = [], [], []
        for a in self.attributes():
            dt = a.datatype()
            names.append(a.name())
            nc = dt.num_components()
            formats.append((dt.numpy_dtype(), (nc,)) if nc > 1 else dt.numpy_dtype())
            offsets.append(a.offset())
        return np.dtype({"names": names, "formats": formats, "offsets": offsets, "itemsize": self.size_of_point_entry()})
