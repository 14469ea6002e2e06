"""ctypes binding of the C ABI declared in include/pasture_amd.h.

`CApi(lib, prefix)` binds one shared library whose entry points are `<prefix>_*`.  The product library is
`pasture_amd/libpasture_amd.so` (prefix ``pst``).  The same class can bind any library exposing the same ABI shape
— the parity tests use that to drive the CPU oracle (prefix ``orc``) through the identical Python surface; the
product package itself never loads anything but its own HIP library and fails loudly when it is missing.
"""
from __future__ import annotations

import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PASTURE_AMD_LIB") or os.path.join(PKG_DIR, "libpasture_amd.so")  # override: A/B runs of two builds

# status codes (include/pasture_amd.h)
OK = 0
ERR_INVALID_ARGUMENT = 1
ERR_LAYOUT_MISMATCH = 2
ERR_RANGE = 3
ERR_MISSING_ATTRIBUTE = 4
ERR_INVALID_CONVERSION = 5
ERR_TRANSFORM_TYPE_MISMATCH = 6
ERR_UNSUPPORTED_TRANSFORM = 7
ERR_DUPLICATE_ATTRIBUTE = 8
ERR_INVALID_LAYOUT = 9
ERR_BOUNDS_INVALID = 10
ERR_TOO_FEW_POINTS = 11
ERR_K_TOO_SMALL = 12
ERR_NOT_ENOUGH_NEIGHBOURS = 13
ERR_UNSUPPORTED_ATTRIBUTE = 14
ERR_HIP = 20
ERR_NO_DEVICE = 21
ERR_OUT_OF_MEMORY = 22
ERR_UNSUPPORTED = 23


class PastureError(RuntimeError):
    """Any non-zero status from the C ABI."""

    def __init__(self, code: int, message: str):
        super().__init__(f"[status {code}] {message}")
        self.code = code
        self.message = message


class PasturePanic(PastureError):
    """Status codes 2..14: conditions on which the Rust reference panics (assert!/expect/panic!)."""


class DataTypeStruct(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("reserved", C.c_uint32), ("size_param", C.c_uint64), ("align_param", C.c_uint64),
                ("uuid", C.c_uint8 * 16)]


class MemberStruct(C.Structure):
    _fields_ = [("name", C.c_char_p), ("datatype", DataTypeStruct), ("offset", C.c_uint64), ("size", C.c_uint64)]


class TransformStruct(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("shift", C.c_uint32), ("datatype", DataTypeStruct), ("scale", C.c_double * 3),
                ("offset", C.c_double * 3), ("mask", C.c_uint64)]


class MappingInfoStruct(C.Structure):
    _fields_ = [("source_name", C.c_char_p), ("target_name", C.c_char_p), ("source_datatype", DataTypeStruct),
                ("target_datatype", DataTypeStruct), ("source_offset", C.c_uint64), ("target_offset", C.c_uint64),
                ("has_converter", C.c_int32), ("transform_kind", C.c_uint32), ("apply_to_source", C.c_int32),
                ("reserved", C.c_int32)]


class JitStatsStruct(C.Structure):
    _fields_ = [("compiled", C.c_uint64), ("disk_hits", C.c_uint64), ("memory_hits", C.c_uint64), ("failures", C.c_uint64),
                ("launches", C.c_uint64), ("compile_seconds", C.c_double)]


_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
_DT = C.POINTER(DataTypeStruct)
_SZ = C.c_size_t
_D3 = C.POINTER(C.c_double)

# name -> argtypes; every function returns int except last_error
_SHARED_SIGNATURES = {
    "layout_create": [_PP],
    "layout_destroy": [_P],
    "layout_clone": [_P, _PP],
    "layout_add_attribute": [_P, C.c_char_p, _DT, C.c_uint32, C.c_uint64],
    "layout_from_members": [C.POINTER(MemberStruct), _SZ, C.c_uint64, _PP],
    "layout_num_attributes": [_P, C.POINTER(_SZ)],
    "layout_get_member": [_P, _SZ, C.POINTER(MemberStruct)],
    "layout_size_of_point_entry": [_P, C.POINTER(C.c_uint64)],
    "layout_alignment": [_P, C.POINTER(C.c_uint64)],
    "layout_equals": [_P, _P, C.POINTER(C.c_int)],
    "buffer_create": [_P, C.c_uint32, C.c_uint32, _PP],
    "buffer_destroy": [_P],
    "buffer_len": [_P, C.POINTER(_SZ)],
    "buffer_resize": [_P, _SZ],
    "buffer_swap": [_P, _SZ, _SZ],
    "buffer_is_columnar": [_P, C.POINTER(C.c_int)],
    "buffer_layout": [_P, _PP],
    "buffer_write_points": [_P, _SZ, _SZ, _P],
    "buffer_read_points": [_P, _SZ, _SZ, _P],
    "buffer_write_attribute": [_P, C.c_char_p, _DT, _SZ, _SZ, _P],
    "buffer_read_attribute": [_P, C.c_char_p, _DT, _SZ, _SZ, _P],
    "buffer_synth_fill": [_P, C.c_uint64, C.c_uint64],
    "converter_create": [_P, _P, C.c_int, _PP],
    "converter_destroy": [_P],
    "converter_set_custom_mapping": [_P, C.c_char_p, _DT, C.c_char_p, _DT],
    "converter_set_custom_mapping_with_transformation": [_P, C.c_char_p, _DT, C.c_char_p, _DT, C.POINTER(TransformStruct), C.c_int],
    "converter_num_mappings": [_P, C.POINTER(_SZ)],
    "converter_get_mapping": [_P, _SZ, C.POINTER(MappingInfoStruct)],
    "converter_convert_into_range": [_P, _P, _SZ, _SZ, _P, _SZ, _SZ],
    "converter_convert": [_P, _P, C.c_uint32, _PP],
    "point_converter_create": [_P, _P, _PP],
    "point_converter_destroy": [_P],
    "point_converter_num_converters": [_P, C.POINTER(_SZ)],
    "point_converter_convert": [_P, _P, _SZ, _P, _SZ, _SZ],
    "calculate_bounds": [_P, _D3, _D3, C.POINTER(C.c_int)],
    "minmax_attribute": [_P, C.c_char_p, _DT, _P, _P, C.POINTER(C.c_int)],
    "transform_attribute": [_P, C.c_char_p, _DT, C.POINTER(TransformStruct)],
    "compute_normals": [_P, _SZ, _D3, _D3, C.POINTER(C.c_int64)],
    "buffer_append": [_P, _P],
    "buffer_filter_into": [_P, _P, _P, C.c_uint32, C.c_int64, C.POINTER(C.c_size_t)],
    "buffer_filter": [_P, _P, C.c_uint32, C.c_uint32, C.POINTER(_P)],
    "voxelgrid_filter": [_P, C.c_double, C.c_double, C.c_double, _P],
    "las_encode_points": [_P, C.c_uint32, _D3, _D3, _P, _SZ, _D3, C.POINTER(C.c_uint64), C.c_uint32],
    "buffer_slice": [_P, _SZ, _SZ, _PP],
    "buffer_read_attribute_converted": [_P, C.c_char_p, _DT, _SZ, _SZ, _P],
    "compute_centroid": [_P, _D3],
}

# entry points only the HIP library has
_PRODUCT_SIGNATURES = {
    "device_count": [C.POINTER(C.c_int)],
    "set_device": [C.c_int],
    "set_stream": [_P],
    "get_stream": [_PP],
    "stream_synchronize": [],
    "buffer_wrap_external": [_P, _P, _SZ, _PP],
    "buffer_wrap_external_columns": [_P, _PP, _SZ, _PP],
    "buffer_points_ptr": [_P, _PP],
    "buffer_column_ptr": [_P, C.c_char_p, _DT, _PP],
    "converter_convert_into_range_async": [_P, _P, _SZ, _SZ, _P, _SZ, _SZ],
    "converter_convert_into_range_with_bounds": [_P, _P, _SZ, _SZ, _P, _SZ, _SZ, _D3, _D3, C.POINTER(C.c_int)],
    "converter_convert_into_range_with_bounds_async": [_P, _P, _SZ, _SZ, _P, _SZ, _SZ, _P],
    "calculate_bounds_async": [_P, _P],
    "las_encode_range_async": [_P, _SZ, _SZ, C.c_uint32, _D3, _D3, _P, _SZ, _P, _P, C.c_uint32],
    "compute_normals_into": [_P, _SZ, _P],
    "compute_normals_device": [_P, _SZ, _P, _P, _P],
    "buffer_filter_into_async": [_P, _P, _P, C.c_size_t, _P],
    "buffer_read_attribute_converted_device": [_P, C.c_char_p, _DT, _SZ, _SZ, _P],
    "voxelgrid_plan_create": [_P, C.c_double, C.c_double, C.c_double, _PP, C.POINTER(_SZ)],
    "voxelgrid_plan_destroy": [_P],
    "voxelgrid_filter_async": [_P, _P, _P, _SZ, _P],
    "compute_normals_plan_create": [_P, _SZ, _P, _PP],
    "normals_plan_destroy": [_P],
    "compute_normals_into_async": [_P, _P, _P, _P],
    "release_scratch": [],
    "reload_tuning": [],
    "comm_unique_id": [_P],
    "comm_init_rank": [C.c_int, C.c_int, _P, _PP],
    "comm_init": [C.c_int, _PP],
    "comm_size": [_P, C.POINTER(C.c_int)],
    "comm_destroy": [_P],
    "bounds_allreduce": [_P, _P],
    "bounds_allreduce_multi": [_P, _PP, _PP],
    "bounds_record_set_form": [_P, C.c_int],
    "last_plan_kinds": [C.POINTER(C.c_uint32)],
    "converter_prepare": [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32)],
    "converter_family_choice": [_P, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float)],
    "converter_measure_families": [_P, _P, C.c_size_t, C.c_size_t, _P, C.c_size_t, C.c_size_t, C.c_int],
    "converter_jit_source": [_P, C.c_int, C.c_int, C.c_int, C.c_char_p, _SZ, C.POINTER(_SZ)],
    "jit_compile_source": [C.c_char_p, _P, _SZ, C.POINTER(_SZ), C.c_char_p, _SZ],
    "jit_get_stats": [C.POINTER(JitStatsStruct)],
    "jit_set_mode": [C.c_int],
    "converter_set_custom_mapping_with_expression": [_P, C.c_char_p, _DT, C.c_char_p, _DT, C.c_char_p, C.c_int],
    "transform_attribute_expr": [_P, C.c_char_p, _DT, C.c_char_p, _PP, _SZ],
    "buffer_filter_expr": [_P, C.c_char_p, _PP, _SZ, C.c_uint32, _PP],
    "expr_source": [C.c_int, _P, _DT, _DT, C.c_int, C.c_char_p, C.c_char_p, _SZ, C.POINTER(_SZ)],
}

PRODUCT_SYMBOLS = ["last_error"] + list(_SHARED_SIGNATURES) + list(_PRODUCT_SIGNATURES)


class CApi:
    def __init__(self, lib: C.CDLL, prefix: str, product: bool):
        self.lib = lib
        self.prefix = prefix
        self.is_product = product
        self._last_error = getattr(lib, f"{prefix}_last_error")
        self._last_error.restype = C.c_char_p
        self._last_error.argtypes = []
        sigs = dict(_SHARED_SIGNATURES)
        if product:
            sigs.update(_PRODUCT_SIGNATURES)
        for name, argtypes in sigs.items():
            fn = getattr(lib, f"{prefix}_{name}")
            fn.restype = C.c_int
            fn.argtypes = argtypes
            setattr(self, name, self._checked(fn))

    def _checked(self, fn):
        def call(*args):
            rc = fn(*args)
            if rc != OK:
                msg = (self._last_error() or b"").decode("utf-8", "replace")
                cls = PasturePanic if 2 <= rc <= 14 else PastureError
                raise cls(rc, msg)
            return rc

        return call


_product_api: CApi | None = None


def product_api() -> CApi:
    """The HIP library.  There is no fallback: a missing or unloadable extension is an error."""
    global _product_api
    if _product_api is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C pasture_amd/csrc`).  pasture_amd has no CPU fallback.")
        # One process must hold ONE HIP runtime: torch bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's), and the
        # first one loaded serves both.  Importing torch first makes torch's runtime the shared instance, so torch streams,
        # events and allocations are valid handles inside this library (and torch keeps seeing the GPU).
        try:
            import torch  # noqa: F401
        except ImportError:  # the C ABI itself does not need torch
            pass
        _product_api = CApi(C.CDLL(LIB_PATH), "pst", product=True)
    return _product_api
