"""BufferLayoutConverter — Python mirror of pasture-core/src/layout/conversion/buffer_conversion.rs:98-663.

The reference accepts arbitrary `Fn(T) -> T` closures for `set_custom_mapping_with_transformation`; a device path
can only evaluate a CLOSED set, described by `Transform` objects (AFFINE, BITFIELD — every closure the reference's
own callers use: pasture-io/src/las/raw_readers.rs:42-164, buffer_conversion.rs:780-782).  Anything else must stay
on the CPU path of the reference.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple, Type

from . import _capi
from ._capi import MappingInfoStruct, TransformStruct
from .buffers import _Buffer, HashMapBuffer, VectorBuffer
from .layout import PointAttributeDataType, PointAttributeDefinition, PointLayout

XF_NONE, XF_AFFINE, XF_BITFIELD = 0, 1, 2


@dataclass(frozen=True)
class Transform:
    """Closed-set stand-in for the reference's `transform_fn: Fn(T) -> T`; `datatype` is the closure's `T`."""
    kind: int
    datatype: PointAttributeDataType
    scale: Tuple[float, float, float] = (1.0, 1.0, 1.0)
    offset: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    shift: int = 0
    mask: int = 0xFFFFFFFFFFFFFFFF

    @staticmethod
    def affine(datatype: PointAttributeDataType, scale: Sequence[float], offset: Sequence[float]) -> "Transform":
        """|p| (p * scale) + offset per component, two roundings (raw_readers.rs:42-48); f32 types compute in f64 (:49-55)."""
        s = tuple(float(x) for x in scale)
        o = tuple(float(x) for x in offset)
        if len(s) == 1:
            s, o = s * 3, o * 3
        return Transform(XF_AFFINE, datatype, s, o)

    @staticmethod
    def add_scalar(datatype: PointAttributeDataType, value: float) -> "Transform":
        """|p| p.add_scalar(value) (buffer_conversion.rs:780-782): p*1.0 is exact, so this is affine with scale 1."""
        return Transform.affine(datatype, (1.0, 1.0, 1.0), (value, value, value))

    @staticmethod
    def bitfield(datatype: PointAttributeDataType, shift: int, mask: int) -> "Transform":
        """|v| (v >> shift) & mask on unsigned integers (raw_readers.rs:61-164)."""
        return Transform(XF_BITFIELD, datatype, shift=shift, mask=mask)

    def to_c(self) -> TransformStruct:
        t = TransformStruct()
        t.kind = self.kind
        t.shift = self.shift
        t.datatype = self.datatype.to_c()
        t.scale[:] = list(self.scale)
        t.offset[:] = list(self.offset)
        t.mask = self.mask
        return t


@dataclass(frozen=True)
class MappingInfo:
    source: PointAttributeDefinition
    target: PointAttributeDefinition
    source_offset: int
    target_offset: int
    has_converter: bool
    transform_kind: int
    apply_to_source: bool


# kernel families a conversion call can take (include/pasture_amd.h PST_PLAN_*)
PLAN_NONE, PLAN_INTERPRETED, PLAN_JIT, PLAN_STATIC, PLAN_LAS, PLAN_STREAM, PLAN_COLUMN, PLAN_COPY, PLAN_DIRECT = range(9)
PLAN_NAMES = ("none", "interpreted", "jit", "static", "las-specialised", "stream", "column", "copy", "direct", "expression")


def last_plan_kinds(api=None) -> List[str]:
    """Names of the kernel families the calling thread's last conversion call launched."""
    api = api or _capi.product_api()
    mask = C.c_uint32()
    api.last_plan_kinds(C.byref(mask))
    return [PLAN_NAMES[k] for k in range(len(PLAN_NAMES)) if mask.value >> k & 1]


def jit_set_mode(mode: str, api=None) -> None:
    """'off' | 'async' | 'sync' | 'env' (back to PST_JIT): when plan-specialised kernels are compiled (tests, A/B harnesses)."""
    (api or _capi.product_api()).jit_set_mode({"env": -1, "off": 0, "async": 1, "sync": 2}[mode])


def jit_stats(api=None) -> dict:
    st = _capi.JitStatsStruct()
    (api or _capi.product_api()).jit_get_stats(C.byref(st))
    return {k: getattr(st, k) for k, _ in st._fields_}


def jit_compile_source(source: str, api=None) -> bytes:
    """hipRTC-compile a translation unit against the embedded device headers for gfx950 (no device needed); returns the code object."""
    n = C.c_size_t()
    log = C.create_string_buffer(1 << 16)
    code = C.create_string_buffer(1 << 20)
    (api or _capi.product_api()).jit_compile_source(source.encode(), code, len(code), C.byref(n), log, len(log))
    return code.raw[:n.value]


def expr_source(kind: str, expression: str, layout: PointLayout = None, src_datatype=None, dst_datatype=None, apply_to_source: bool = False, api=None) -> str:
    """The translation unit a device expression becomes: kind "transform" between the two datatypes; over `layout`: "predicate" (-> byte mask),
    "predicate-count" (-> matches per 2048-point tile), "predicate-filter-columns" / "predicate-filter-records" (the streaming compaction kernel
    with the predicate inside; '' when the layout does not take that kernel)."""
    api = api or (layout.api if layout is not None else _capi.product_api())
    n = C.c_size_t()
    sd = src_datatype.to_c() if src_datatype is not None else None
    dd = dst_datatype.to_c() if dst_datatype is not None else None
    kinds = {"transform": 0, "predicate": 1, "predicate-count": 2, "predicate-filter-columns": 3, "predicate-filter-records": 4}
    args = (kinds[kind], layout._h if layout is not None else None, C.byref(sd) if sd is not None else None,
            C.byref(dd) if dd is not None else None, 1 if apply_to_source else 0, expression.encode())
    api.expr_source(*args, None, 0, C.byref(n))
    buf = C.create_string_buffer(n.value)
    api.expr_source(*args, buf, n.value, C.byref(n))
    return buf.value.decode()


class RawPointConverter:
    """attribute_conversion.rs:62-109 — the point-major converter: `from_to` collects one `as` converter per attribute present in both
    layouts whose datatypes differ (equal datatypes: no converter, the attribute is SKIPPED, not copied); `convert` runs them on
    interleaved points.  The reference converts one point slice per call; `convert` here takes point indices and a count."""

    def __init__(self, from_layout: PointLayout, to_layout: PointLayout):
        assert from_layout.api is to_layout.api
        self.api = from_layout.api
        h = C.c_void_p()
        self.api.point_converter_create(from_layout._h, to_layout._h, C.byref(h))
        self._h = h

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                self.api.point_converter_destroy(self._h)
        except Exception:
            pass

    @classmethod
    def from_to(cls, from_layout: PointLayout, to_layout: PointLayout) -> "RawPointConverter":  # :69-96
        return cls(from_layout, to_layout)

    def num_converters(self) -> int:
        n = C.c_size_t()
        self.api.point_converter_num_converters(self._h, C.byref(n))
        return n.value

    def convert(self, source, source_point: int, target, target_point: int, count: int = 1) -> None:  # :104-108
        self.api.point_converter_convert(self._h, source._h, source_point, target._h, target_point, count)


class BufferLayoutConverter:
    def __init__(self, from_layout: PointLayout, to_layout: PointLayout, with_default: bool):
        assert from_layout.api is to_layout.api
        self.api = from_layout.api
        self.from_layout = from_layout
        self.to_layout = to_layout
        h = C.c_void_p()
        self.api.converter_create(from_layout._h, to_layout._h, 1 if with_default else 0, C.byref(h))
        self._h = h

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                self.api.converter_destroy(self._h)
        except Exception:
            pass

    @classmethod
    def for_layouts(cls, from_layout: PointLayout, to_layout: PointLayout) -> "BufferLayoutConverter":  # :112-123
        return cls(from_layout, to_layout, False)

    @classmethod
    def for_layouts_with_default(cls, from_layout: PointLayout, to_layout: PointLayout) -> "BufferLayoutConverter":  # :126-143
        return cls(from_layout, to_layout, True)

    def set_custom_mapping(self, from_attribute: PointAttributeDefinition, to_attribute: PointAttributeDefinition) -> None:  # :156-183
        f, t = from_attribute.datatype().to_c(), to_attribute.datatype().to_c()
        self.api.converter_set_custom_mapping(self._h, from_attribute.name().encode(), C.byref(f), to_attribute.name().encode(), C.byref(t))

    def set_custom_mapping_with_transformation(self, from_attribute: PointAttributeDefinition, to_attribute: PointAttributeDefinition,
                                               transform_fn: Transform, apply_to_source_attribute: bool) -> None:  # :194-234
        if not isinstance(transform_fn, Transform):
            raise _capi.PastureError(_capi.ERR_UNSUPPORTED_TRANSFORM,
                                     "arbitrary closures cannot run on the device; describe the transformation with pasture_amd.Transform")
        f, t = from_attribute.datatype().to_c(), to_attribute.datatype().to_c()
        x = transform_fn.to_c()
        self.api.converter_set_custom_mapping_with_transformation(self._h, from_attribute.name().encode(), C.byref(f),
                                                                  to_attribute.name().encode(), C.byref(t), C.byref(x),
                                                                  1 if apply_to_source_attribute else 0)

    def set_custom_mapping_with_expression(self, from_attribute: PointAttributeDefinition, to_attribute: PointAttributeDefinition,
                                           expression: str, apply_to_source_attribute: bool) -> None:
        """set_custom_mapping_with_transformation (:194-234) with the closure as a DEVICE EXPRESSION (include/pasture_amd.h, "device
        expressions"): C++ expression text over v (this component), x y z (the Vec3's components), c (component index), i (point index),
        compiled at run time; T is the source datatype when apply_to_source_attribute, the target's otherwise."""
        f, t = from_attribute.datatype().to_c(), to_attribute.datatype().to_c()
        self.api.converter_set_custom_mapping_with_expression(self._h, from_attribute.name().encode(), C.byref(f), to_attribute.name().encode(), C.byref(t),
                                                              expression.encode(), 1 if apply_to_source_attribute else 0)

    def mappings(self) -> List[MappingInfo]:
        n = C.c_size_t()
        self.api.converter_num_mappings(self._h, C.byref(n))
        out = []
        for i in range(n.value):
            m = MappingInfoStruct()
            self.api.converter_get_mapping(self._h, i, C.byref(m))
            out.append(MappingInfo(PointAttributeDefinition(m.source_name.decode(), PointAttributeDataType.from_c(m.source_datatype)),
                                   PointAttributeDefinition(m.target_name.decode(), PointAttributeDataType.from_c(m.target_datatype)),
                                   m.source_offset, m.target_offset, bool(m.has_converter), m.transform_kind, bool(m.apply_to_source)))
        return out

    def convert(self, source_buffer: _Buffer, out_buffer_type: Type[_Buffer] = HashMapBuffer) -> _Buffer:  # :242-259
        h = C.c_void_p()
        self.api.converter_convert(self._h, source_buffer._h, out_buffer_type._storage, C.byref(h))
        return out_buffer_type(h.value, self.api)

    def convert_into(self, source_buffer: _Buffer, target_buffer: _Buffer) -> None:  # :268-283
        n = source_buffer.len()
        self.convert_into_range(source_buffer, range(0, n), target_buffer, range(0, n))

    def convert_into_range(self, source_buffer: _Buffer, source_range: range, target_buffer: _Buffer, target_range: range) -> None:  # :292-359
        self.api.converter_convert_into_range(self._h, source_buffer._h, source_range.start, source_range.stop, target_buffer._h,
                                              target_range.start, target_range.stop)

    # ---- device-only extras ---------------------------------------------------------------------------------
    def prepare(self, source_type: Type[_Buffer], target_type: Type[_Buffer], with_bounds: bool = False) -> int:
        """Compile (hipRTC, cached) the plan-specialised kernel for conversions between buffers of these storage kinds NOW instead of on a
        background thread after the first large call.  Returns the PLAN_* family such a conversion will take."""
        kind = C.c_uint32()
        self.api.converter_prepare(self._h, 1 if source_type._storage == HashMapBuffer._storage else 0,
                                   1 if target_type._storage == HashMapBuffer._storage else 0, 1 if with_bounds else 0, C.byref(kind))
        return kind.value

    def family_choice(self, target_type: Type[_Buffer], with_bounds: bool = False, source_type: Optional[Type[_Buffer]] = None):
        """Which of the two kernel families that can serve an interleaved LAS-shaped plan this converter measured to be the faster one on this
        device (first SYNCHRONOUS conversion of at least 2^22 points, or measure_families): (choice, (ms LAS family, ms plan-specialised));
        choice -1 = not measured yet, 0 = LAS family, 1 = plan-specialised, 2 = the plan has one family only.  source_type (default: records)
        selects the pairing: records from COLUMNS has its own slot."""
        choice, ms = C.c_int(), (C.c_float * 2)()
        dst_col = target_type._storage == HashMapBuffer._storage
        src_col = source_type is not None and source_type._storage == HashMapBuffer._storage
        self.api.converter_family_choice(self._h, 1 if dst_col else (2 if src_col else 0), 1 if with_bounds else 0, C.byref(choice), ms)
        return choice.value, (ms[0], ms[1])

    def measure_families(self, source_buffer: _Buffer, target_buffer: _Buffer, with_bounds: bool = False, source_range: Optional[range] = None,
                         target_range: Optional[range] = None) -> None:
        """The measurement behind family_choice, run NOW on these buffers (converts the range several times -- same bytes -- and waits): what a
        caller of the stream-ordered `_async` conversions does once before its loop, since those never measure."""
        n = source_buffer.len()
        sr = range(0, n) if source_range is None else source_range
        tr = range(0, n) if target_range is None else target_range
        self.api.converter_measure_families(self._h, source_buffer._h, sr.start, sr.stop, target_buffer._h, tr.start, tr.stop, 1 if with_bounds else 0)

    def jit_source(self, source_type: Type[_Buffer], target_type: Type[_Buffer], with_bounds: bool = False) -> str:
        """The translation unit the run-time compiler is handed for this converter and storage pairing ('' if another family serves it)."""
        need = C.c_size_t()
        sc, dc = (1 if t._storage == HashMapBuffer._storage else 0 for t in (source_type, target_type))
        self.api.converter_jit_source(self._h, sc, dc, 1 if with_bounds else 0, None, 0, C.byref(need))
        buf = C.create_string_buffer(need.value)
        self.api.converter_jit_source(self._h, sc, dc, 1 if with_bounds else 0, buf, need.value, None)
        return buf.value.decode()

    def convert_into_range_async(self, source_buffer, source_range, target_buffer, target_range) -> None:
        self.api.converter_convert_into_range_async(self._h, source_buffer._h, source_range.start, source_range.stop, target_buffer._h,
                                                    target_range.start, target_range.stop)

    def convert_into_with_bounds(self, source_buffer, target_buffer, source_range: Optional[range] = None,
                                 target_range: Optional[range] = None):
        """convert_into_range + calculate_bounds(target range) in one pass over HBM.  Returns AABB or None."""
        from .algorithms import AABB
        n = source_buffer.len()
        sr = range(0, n) if source_range is None else source_range  # NB: an empty range is falsy
        tr = range(0, n) if target_range is None else target_range
        mn, mx, has = (C.c_double * 3)(), (C.c_double * 3)(), C.c_int()
        self.api.converter_convert_into_range_with_bounds(self._h, source_buffer._h, sr.start, sr.stop, target_buffer._h, tr.start, tr.stop,
                                                          mn, mx, C.byref(has))
        return AABB(tuple(mn), tuple(mx)) if has.value else None

    def convert_into_with_bounds_async(self, source_buffer, target_buffer, device_out6_ptr: int, source_range=None, target_range=None) -> None:
        n = source_buffer.len()
        sr = range(0, n) if source_range is None else source_range  # NB: an empty range is falsy
        tr = range(0, n) if target_range is None else target_range
        self.api.converter_convert_into_range_with_bounds_async(self._h, source_buffer._h, sr.start, sr.stop, target_buffer._h, tr.start,
                                                                tr.stop, C.c_void_p(device_out6_ptr))
