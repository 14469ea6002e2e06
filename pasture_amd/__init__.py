"""pasture_amd — MI355X-native implementation of pasture's per-point attribute-transform hot path.

Host-side mirror of the reference's interface (pasture-core layout / containers / layout::conversion and the
pasture-algorithms loops) on top of the C ABI in include/pasture_amd.h, which is implemented by hand-written HIP
kernels for gfx950 (pasture_amd/csrc).  No CPU fallback: importing the package without the built extension fails.
"""
from ._capi import PastureError, PasturePanic, product_api, LIB_PATH  # noqa: F401
from .layout import (FieldAlignment, PointAttributeDataType, PointAttributeDefinition, PointAttributeMember, PointLayout,  # noqa: F401
                     attributes)
from .buffers import ExternalColumnsBuffer, ExternalMemoryBuffer, HashMapBuffer, VectorBuffer  # noqa: F401
from .conversion import BufferLayoutConverter, RawPointConverter, Transform  # noqa: F401
from .algorithms import (AABB, calculate_bounds, calculate_bounds_async, compute_centroid, compute_normals, compute_normals_into, minmax_attribute, voxelgrid_filter,  # noqa: F401
                         transform_attribute)

product_api()  # load libpasture_amd.so now: a missing HIP extension must fail loudly, not at first use
