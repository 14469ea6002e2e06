"""pasture-algorithms loops — Python mirror over the C ABI.

  calculate_bounds   pasture-algorithms/src/bounds.rs:11-85
  minmax_attribute   pasture-algorithms/src/minmax.rs:13-51
  transform_attribute  pasture-core/src/containers/point_buffer.rs:391-404 (closed-set transformations)
  compute_normals    pasture-algorithms/src/normal_estimation.rs:79-130
  compute_centroid   pasture-algorithms/src/normal_estimation.rs:198-237
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from .buffers import _Buffer
from .conversion import Transform
from .layout import PointAttributeDefinition


@dataclass(frozen=True)
class AABB:
    """math::AABB<f64>, pasture-core/src/math/bounds.rs:9-26."""
    _min: Tuple[float, float, float]
    _max: Tuple[float, float, float]

    def min(self):
        return self._min

    def max(self):
        return self._max

    def extent(self):
        return tuple(b - a for a, b in zip(self._min, self._max))

    @staticmethod
    def union(a: "AABB", b: "AABB") -> "AABB":  # bounds.rs:109-122
        return AABB(tuple(x if x < y else y for x, y in zip(a._min, b._min)), tuple(x if x > y else y for x, y in zip(a._max, b._max)))


def calculate_bounds(buffer: _Buffer) -> Optional[AABB]:
    mn, mx, has = (C.c_double * 3)(), (C.c_double * 3)(), C.c_int()
    buffer.api.calculate_bounds(buffer._h, mn, mx, C.byref(has))
    return AABB(tuple(mn), tuple(mx)) if has.value else None


def calculate_bounds_async(buffer: _Buffer, device_out6_ptr: int) -> None:
    """Stream-ordered: writes {min xyz, max xyz} (seeds +/-f64::MAX) to device memory, no host synchronisation."""
    buffer.api.calculate_bounds_async(buffer._h, C.c_void_p(device_out6_ptr))


def minmax_attribute(buffer: _Buffer, attribute: PointAttributeDefinition):
    """Returns (min, max) as numpy scalars / length-3 arrays of the attribute's datatype, or None for an empty buffer."""
    dt = attribute.datatype()
    nc = dt.num_components()
    mn = np.zeros(nc, dtype=dt.numpy_dtype())
    mx = np.zeros(nc, dtype=dt.numpy_dtype())
    has = C.c_int()
    cdt = dt.to_c()
    buffer.api.minmax_attribute(buffer._h, attribute.name().encode(), C.byref(cdt), mn.ctypes.data_as(C.c_void_p),
                                mx.ctypes.data_as(C.c_void_p), C.byref(has))
    if not has.value:
        return None
    return (mn[0], mx[0]) if nc == 1 else (mn, mx)


def transform_attribute(buffer: _Buffer, attribute: PointAttributeDefinition, transform: Transform) -> None:
    cdt = attribute.datatype().to_c()
    x = transform.to_c()
    buffer.api.transform_attribute(buffer._h, attribute.name().encode(), C.byref(cdt), C.byref(x))


def transform_attribute_expr(buffer: _Buffer, attribute: PointAttributeDefinition, expression: str, device_params=()) -> None:
    """transform_attribute(attribute, |index, value| expression), point_buffer.rs:391-404: a device expression over v, x y z, c, i (the
    closure's index) and p0 .. p3 = `device_params` (addresses of device arrays of double: what the closure would capture)."""
    cdt = attribute.datatype().to_c()
    arr = (C.c_void_p * max(1, len(device_params)))(*[C.c_void_p(int(p)) for p in device_params])
    buffer.api.transform_attribute_expr(buffer._h, attribute.name().encode(), C.byref(cdt), expression.encode(), C.cast(arr, C.POINTER(C.c_void_p)) if device_params else None,
                                        len(device_params))


def compute_normals(point_cloud: _Buffer, k_nn: int, return_knn: bool = False):
    """Vec<(Vector3<f64>, f64)> as (normals (n,3) f64, curvature (n,) f64[, knn indices (n,k) int64])."""
    n = point_cloud.len()
    normals = np.zeros((n, 3), dtype=np.float64)
    curv = np.zeros(n, dtype=np.float64)
    knn = np.full((n, max(k_nn, 1)), -1, dtype=np.int64) if return_knn else None
    point_cloud.api.compute_normals(point_cloud._h, k_nn, normals.ctypes.data_as(C.POINTER(C.c_double)),
                                    curv.ctypes.data_as(C.POINTER(C.c_double)),
                                    knn.ctypes.data_as(C.POINTER(C.c_int64)) if knn is not None else None)
    return (normals, curv, knn) if return_knn else (normals, curv)


def compute_centroid(point_cloud: _Buffer) -> Tuple[float, float, float]:
    """normal_estimation.rs:198-237: mean Position3D (Vec3f64) over all points, over the finite ones when some coordinate is NaN;
    panics (status 11) on an empty cloud."""
    c = (C.c_double * 3)()
    point_cloud.api.compute_centroid(point_cloud._h, c)
    return tuple(c)


def compute_normals_into(point_cloud: _Buffer, k_nn: int, target: _Buffer) -> None:
    """Device-resident: writes NORMAL (Vec3f32) and "Curvature" (F64) attributes of `target`."""
    point_cloud.api.compute_normals_into(point_cloud._h, k_nn, target._h)


def compute_normals_device(point_cloud: _Buffer, k_nn: int, normals_ptr: int = 0, curvature_ptr: int = 0, knn_ptr: int = 0) -> None:
    """The result of compute_normals in caller-owned DEVICE memory (addresses; 0 = not wanted): normals f64 [n][3], curvature f64 [n],
    neighbour lists uint32 [n][k] in ascending distance."""
    point_cloud.api.compute_normals_device(point_cloud._h, k_nn, C.c_void_p(normals_ptr or None), C.c_void_p(curvature_ptr or None),
                                           C.c_void_p(knn_ptr or None))


def reload_tuning(api=None) -> None:
    """The PST_KNN_* switches are read from the environment once per process; this reads them again (tests and A/B harnesses that change
    them inside one process).  The oracle has no such switches: a no-op there."""
    from ._capi import product_api
    api = api or product_api()
    if hasattr(api, "reload_tuning"):
        api.reload_tuning()


def release_scratch(api=None) -> None:
    """Frees the device scratch the calling thread's compute_normals* calls keep between calls (about 55 bytes per point of the largest
    recent cloud, never more than PST_SCRATCH_MAX_BYTES = 16 GiB by default).  Never needed for correctness."""
    from ._capi import product_api
    (api or product_api()).release_scratch()


def voxelgrid_filter(buffer: _Buffer, leafsize_x: float, leafsize_y: float, leafsize_z: float, filtered_buffer: _Buffer) -> None:
    """voxel_grid.rs:109-166: down-samples `buffer` to one centroid per occupied voxel (cells centred on the axis markers),
    appended to `filtered_buffer` in (x, y, z) voxel order; per-attribute reductions of set_all_attributes (:459-689)."""
    buffer.api.voxelgrid_filter(buffer._h, leafsize_x, leafsize_y, leafsize_z, filtered_buffer._h)


class VoxelGridPlan:
    """Stream-ordered voxelgrid_filter (pst_voxelgrid_plan_create / pst_voxelgrid_filter_async): ONE synchronous pass over `buffer` sizes every
    scratch buffer; `filter_async` then runs bounds -> markers -> keys -> sort -> run heads -> reductions on the current stream without a
    host round trip or an allocation (hipGraph-capturable).  `filtered_buffer` must already hold dst_first + max_voxels points; the voxel
    count and a status word (0 = ok) land in the two uint64 at `count_and_status_ptr` (device-accessible memory) in stream order."""

    def __init__(self, buffer: _Buffer, leafsize_x: float, leafsize_y: float, leafsize_z: float):
        self.api = buffer.api
        h, mv = C.c_void_p(), C.c_size_t()
        self.api.voxelgrid_plan_create(buffer._h, leafsize_x, leafsize_y, leafsize_z, C.byref(h), C.byref(mv))
        self._h, self.max_voxels = h, mv.value

    def filter_async(self, buffer: _Buffer, filtered_buffer: _Buffer, dst_first: int, count_and_status_ptr: int) -> None:
        self.api.voxelgrid_filter_async(self._h, buffer._h, filtered_buffer._h, dst_first, C.c_void_p(int(count_and_status_ptr)))

    def destroy(self) -> None:
        if self._h is not None and self._h.value:
            self.api.voxelgrid_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class NormalsPlan:
    """Stream-ordered compute_normals_into (pst_compute_normals_plan_create / pst_compute_normals_into_async): the constructor runs the
    synchronous call once (`target` holds its result) and keeps its decisions; `compute_into_async` replays the pipeline on the current stream
    without a host round trip or an allocation.  Two uint64 at `status_ptr` (device-accessible): [0] = 0 when the result is complete."""

    def __init__(self, point_cloud: _Buffer, k_nn: int, target: _Buffer):
        self.api = point_cloud.api
        h = C.c_void_p()
        self.api.compute_normals_plan_create(point_cloud._h, k_nn, target._h, C.byref(h))
        self._h = h

    def compute_into_async(self, point_cloud: _Buffer, target: _Buffer, status_ptr: int) -> None:
        self.api.compute_normals_into_async(self._h, point_cloud._h, target._h, C.c_void_p(int(status_ptr)))

    def destroy(self) -> None:
        if self._h is not None and self._h.value:
            self.api.normals_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
