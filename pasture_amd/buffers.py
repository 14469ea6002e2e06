"""Point buffers backed by HBM — Python mirror of pasture-core/src/containers/point_buffer.rs over the C ABI.

  VectorBuffer          interleaved (AoS), library-owned device memory          point_buffer.rs:659-945
  HashMapBuffer         columnar (SoA), one device column per attribute          point_buffer.rs:1031-1474
  ExternalMemoryBuffer  interleaved view over caller-owned device memory        point_buffer.rs:1479-1708
                        (here: a torch CUDA tensor / raw device pointer)

Host access goes through numpy arrays (`set_point_range`, `get_attribute_range`, ...) and is a real H2D/D2H
transfer; the bulk entry points (conversion, bounds, ...) never leave the device.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _capi
from ._capi import CApi
from .layout import PointAttributeDefinition, PointLayout

STORAGE_INTERLEAVED = 0
STORAGE_COLUMNAR = 1
MEM_DEVICE = 0
MEM_PINNED_HOST = 1


class _Buffer:
    _storage = STORAGE_INTERLEAVED

    def __init__(self, handle: int, api: CApi, keepalive=None):
        self.api = api
        self._h = C.c_void_p(handle)
        self._keepalive = keepalive
        self._layout: Optional[PointLayout] = None

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                self.api.buffer_destroy(self._h)
        except Exception:
            pass

    # MakeBufferFromLayout::new_from_layout :497-500
    @classmethod
    def new_from_layout(cls, point_layout: PointLayout, memkind: int = MEM_DEVICE):
        h = C.c_void_p()
        point_layout.api.buffer_create(point_layout._h, cls._storage, memkind, C.byref(h))
        return cls(h.value, point_layout.api)

    @classmethod
    def with_capacity(cls, capacity: int, point_layout: PointLayout):  # :666-673 / :1040-1057 (capacity is a hint)
        return cls.new_from_layout(point_layout)

    @classmethod
    def from_numpy(cls, points: np.ndarray, point_layout: PointLayout):
        """FromIterator<T: PointType>: collect a structured / raw byte array of points (one record per point)."""
        buf = cls.new_from_layout(point_layout)
        raw = _as_record_bytes(points, point_layout)
        buf.resize(raw.shape[0])
        buf.set_point_range(range(0, raw.shape[0]), raw)
        return buf

    # BorrowedBuffer ------------------------------------------------------------------------------------
    def len(self) -> int:  # :29
        n = C.c_size_t()
        self.api.buffer_len(self._h, C.byref(n))
        return n.value

    __len__ = len

    def is_empty(self) -> bool:
        return self.len() == 0

    def point_layout(self) -> PointLayout:  # :33
        if self._layout is None:
            h = C.c_void_p()
            self.api.buffer_layout(self._h, C.byref(h))
            self._layout = PointLayout(self.api, h.value)
        return self._layout

    def as_columnar(self):  # :149-151
        r = C.c_int()
        self.api.buffer_is_columnar(self._h, C.byref(r))
        return self if r.value else None

    def as_interleaved(self):  # :143-145
        return None if self.as_columnar() is not None else self

    def get_point_range(self, point_range: range) -> np.ndarray:  # :45
        """Returns a (count, size_of_point_entry) uint8 array."""
        first, count = _range(point_range)
        stride = self.point_layout().size_of_point_entry()
        out = np.zeros((count, stride), dtype=np.uint8)
        self.api.buffer_read_points(self._h, first, count, out.ctypes.data_as(C.c_void_p))
        return out

    def get_point(self, index: int) -> np.ndarray:  # :41
        return self.get_point_range(range(index, index + 1))[0]

    def get_attribute_range(self, attribute: PointAttributeDefinition, point_range: range) -> np.ndarray:  # :71-93
        """Typed array: shape (count,) for scalars, (count, 3) for Vec3, (count, size) uint8 for opaque types."""
        first, count = _range(point_range)
        dt = attribute.datatype()
        nc = dt.num_components()
        out = np.zeros((count, nc) if nc > 1 else (count,), dtype=dt.numpy_dtype())
        cdt = dt.to_c()
        self.api.buffer_read_attribute(self._h, attribute.name().encode(), C.byref(cdt), first, count, out.ctypes.data_as(C.c_void_p))
        return out

    def get_attribute(self, attribute: PointAttributeDefinition, index: int):  # :57-69
        return self.get_attribute_range(attribute, range(index, index + 1))[0]

    def view_attribute(self, attribute: PointAttributeDefinition) -> np.ndarray:
        """view_attribute::<T>(attr).into_iter().collect() — the whole attribute as a host array (buffer_views.rs:291-369)."""
        return self.get_attribute_range(attribute, range(0, self.len()))

    def view_attribute_with_conversion(self, attribute: PointAttributeDefinition, point_range: Optional[range] = None) -> np.ndarray:
        """view_attribute_with_conversion::<T>(attr).into_iter().collect() (point_buffer.rs:322-330, buffer_views.rs:533-650): the
        attribute NAMED like `attribute`, converted from its stored datatype to attribute.datatype() with the Rust-`as` table."""
        first, count = _range(point_range if point_range is not None else range(0, self.len()))
        dt = attribute.datatype()
        nc = dt.num_components()
        out = np.zeros((count, nc) if nc > 1 else (count,), dtype=dt.numpy_dtype())
        cdt = dt.to_c()
        self.api.buffer_read_attribute_converted(self._h, attribute.name().encode(), C.byref(cdt), first, count, out.ctypes.data_as(C.c_void_p))
        return out

    # SliceBuffer / SliceBufferMut, slice.rs:16-43 -----------------------------------------------------------
    def slice(self, point_range: range):
        """A view of `point_range` with the same layout and storage kind; every algorithm / conversion takes it like a buffer.  It borrows
        this buffer's memory (kept alive through the view); resizing the parent while a view exists is what Rust's borrow checker forbids."""
        first, count = _range(point_range)
        h = C.c_void_p()
        self.api.buffer_slice(self._h, first, count, C.byref(h))
        view = _Buffer.__new__(HashMapBuffer if self.as_columnar() is not None else VectorBuffer)
        _Buffer.__init__(view, h.value, self.api, keepalive=self)
        return view

    slice_mut = slice

    # BorrowedMutBuffer ---------------------------------------------------------------------------------
    def set_point_range(self, point_range: range, point_data: np.ndarray) -> None:  # :90
        first, count = _range(point_range)
        raw = _as_record_bytes(point_data, self.point_layout())
        if raw.shape[0] != count:
            raise _capi.PasturePanic(_capi.ERR_RANGE, "source slice length does not match destination slice length")
        self.api.buffer_write_points(self._h, first, count, raw.ctypes.data_as(C.c_void_p))

    def set_attribute_range(self, attribute: PointAttributeDefinition, point_range: range, attribute_data: np.ndarray) -> None:  # :110
        first, count = _range(point_range)
        dt = attribute.datatype()
        arr = np.ascontiguousarray(attribute_data, dtype=dt.numpy_dtype())
        if arr.size != count * dt.num_components():
            raise _capi.PasturePanic(_capi.ERR_RANGE, "source slice length does not match destination slice length")
        cdt = dt.to_c()
        self.api.buffer_write_attribute(self._h, attribute.name().encode(), C.byref(cdt), first, count, arr.ctypes.data_as(C.c_void_p))

    def swap(self, from_index: int, to_index: int) -> None:  # BorrowedMutBuffer::swap :229 (panics when an index is out of bounds)
        self.api.buffer_swap(self._h, from_index, to_index)

    # OwningBuffer --------------------------------------------------------------------------------------
    def resize(self, count: int) -> None:  # :263 (new points zero-filled)
        self.api.buffer_resize(self._h, count)

    def clear(self) -> None:  # :266
        self.resize(0)

    def append(self, other: "_Buffer") -> None:  # OwningBufferExt::append :419-489
        self.api.buffer_append(self._h, other._h)
        self._keepalive = getattr(self, "_keepalive", None)

    @staticmethod
    def _mask_arg(predicate, n: int):
        """The reference's predicate is `Fn(usize) -> bool`; here: a bool/uint8 numpy array, a callable evaluated on the
        host into such an array, or a (device_pointer, 'device') pair for a mask that already lives in HBM."""
        if isinstance(predicate, tuple) and len(predicate) == 2 and predicate[1] == "device":
            return C.c_void_p(int(predicate[0])), MEM_DEVICE, None
        if callable(predicate):
            predicate = np.fromiter((bool(predicate(i)) for i in range(n)), dtype=np.bool_, count=n)
        m = np.ascontiguousarray(np.asarray(predicate)).astype(np.uint8, copy=False)
        if m.shape != (n,):
            raise ValueError(f"mask must have {n} entries")
        return m.ctypes.data_as(C.c_void_p), 1, m

    def filter(self, out_buffer_type, predicate):  # HashMapBuffer::filter :1064-1076
        ptr, kind, keep = self._mask_arg(predicate, self.len())
        h = C.c_void_p()
        self.api.buffer_filter(self._h, ptr, kind, out_buffer_type._storage, C.byref(h))
        return out_buffer_type(h.value, self.api)

    def filter_expr(self, out_buffer_type, expression: str, device_params=()):
        """HashMapBuffer::filter(|index| expression) with the predicate as a device expression over the layout's attribute names (scalars by
        value, Vec3 as .x .y .z), i and p0 .. p3: "Classification == 2 && Position3D.z < 120.0"."""
        h = C.c_void_p()
        arr = (C.c_void_p * max(1, len(device_params)))(*[C.c_void_p(int(p)) for p in device_params])
        self.api.buffer_filter_expr(self._h, expression.encode(), C.cast(arr, C.POINTER(C.c_void_p)) if device_params else None, len(device_params),
                                    out_buffer_type._storage, C.byref(h))
        return out_buffer_type(h.value, self.api)

    def filter_into(self, buffer: "_Buffer", predicate, num_matches_hint: Optional[int] = None) -> int:  # :1082-1136
        ptr, kind, keep = self._mask_arg(predicate, self.len())
        n = C.c_size_t()
        self.api.buffer_filter_into(self._h, buffer._h, ptr, kind, -1 if num_matches_hint is None else num_matches_hint, C.byref(n))
        return n.value

    def filter_into_async(self, buffer: "_Buffer", device_mask_ptr: int, num_matches: int, device_count_ptr: int = 0) -> None:
        """Stream-ordered filter_into for a device mask and a known count (`Some(num_matches)`): nothing waits on the host; the number
        of mask hits lands in the 8 bytes at device_count_ptr (optional) in stream order."""
        self.api.buffer_filter_into_async(self._h, buffer._h, C.c_void_p(int(device_mask_ptr)), num_matches, C.c_void_p(int(device_count_ptr) or None))

    # device-side helpers ---------------------------------------------------------------------------------
    def synth_fill(self, seed: int, first_index: int = 0) -> None:
        self.api.buffer_synth_fill(self._h, seed, first_index)


class VectorBuffer(_Buffer):
    _storage = STORAGE_INTERLEAVED

    def points_ptr(self) -> int:
        p = C.c_void_p()
        self.api.buffer_points_ptr(self._h, C.byref(p))
        return p.value or 0


class HashMapBuffer(_Buffer):
    _storage = STORAGE_COLUMNAR

    def column_ptr(self, attribute: PointAttributeDefinition) -> int:
        p = C.c_void_p()
        cdt = attribute.datatype().to_c()
        self.api.buffer_column_ptr(self._h, attribute.name().encode(), C.byref(cdt), C.byref(p))
        return p.value or 0


class ExternalMemoryBuffer(_Buffer):
    """Interleaved view over caller-owned DEVICE memory (a torch CUDA uint8 tensor or a raw device pointer)."""
    _storage = STORAGE_INTERLEAVED

    def __init__(self, external_memory, point_layout: PointLayout, nbytes: Optional[int] = None):
        api = point_layout.api
        if hasattr(external_memory, "data_ptr"):  # torch tensor
            ptr = external_memory.data_ptr()
            nbytes = external_memory.numel() * external_memory.element_size()
        else:
            ptr = int(external_memory)
            assert nbytes is not None
        h = C.c_void_p()
        api.buffer_wrap_external(point_layout._h, C.c_void_p(ptr), nbytes, C.byref(h))
        super().__init__(h.value, api, keepalive=external_memory)

    def points_ptr(self) -> int:
        p = C.c_void_p()
        self.api.buffer_points_ptr(self._h, C.byref(p))
        return p.value or 0


class ExternalColumnsBuffer(HashMapBuffer):
    """Columnar view over caller-owned device columns (e.g. one torch tensor per attribute, layout order)."""

    def __init__(self, columns, point_layout: PointLayout, length: int):
        api = point_layout.api
        ptrs = (C.c_void_p * max(1, len(columns)))()
        for i, c in enumerate(columns):
            ptrs[i] = c.data_ptr() if hasattr(c, "data_ptr") else int(c)
        h = C.c_void_p()
        api.buffer_wrap_external_columns(point_layout._h, ptrs, length, C.byref(h))
        _Buffer.__init__(self, h.value, api, keepalive=list(columns))


def _range(r: range):
    if r.step != 1:
        raise ValueError("point ranges must be contiguous")
    return r.start, max(0, r.stop - r.start)


def _as_record_bytes(points: np.ndarray, layout: PointLayout) -> np.ndarray:
    stride = layout.size_of_point_entry()
    arr = np.ascontiguousarray(points)
    if arr.dtype.itemsize == stride and arr.dtype.names is not None:
        return arr.view(np.uint8).reshape(arr.shape[0], stride)
    if arr.dtype == np.uint8:
        return arr.reshape(-1, stride) if stride else arr.reshape(0, 0)
    raise TypeError("points must be a structured array with the layout's record dtype or raw uint8 bytes")
