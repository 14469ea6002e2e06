"""Shared helpers for the parity tests."""
import numpy as np

from pasture_amd.buffers import HashMapBuffer, VectorBuffer
from pasture_amd.layout import FieldAlignment, PointAttributeDataType as T, PointAttributeDefinition, PointLayout, attributes as A

BUFFER_KINDS = {"V": VectorBuffer, "H": HashMapBuffer}
PAIRINGS = [("V", "V"), ("V", "H"), ("H", "V"), ("H", "H")]  # the reference's 4 buffer pairings (buffer_conversion.rs:875-910)


def custom_point_type_small(api):
    """CustomPointTypeSmall, pasture-core/src/test_utils.rs:8-14 (repr(C, packed))."""
    return PointLayout.from_attributes_packed([A.POSITION_3D, A.CLASSIFICATION], 1, api=api)


def custom_point_type_big(api):
    """CustomPointTypeBig, test_utils.rs:19-31: gps_time f64, color Vec3u16, position Vec3f64, classification u8, intensity I16."""
    return PointLayout.from_attributes_packed(
        [A.GPS_TIME, A.COLOR_RGB, A.POSITION_3D, A.CLASSIFICATION, A.INTENSITY.with_custom_datatype(T.I16)], 1, api=api)


def random_records(layout: PointLayout, n: int, seed: int) -> np.ndarray:
    """DefaultPointDistribution (test_utils.rs:33-54) with a seed: every field uniformly random, floats in [0,1)."""
    rng = np.random.default_rng(seed)
    rec = np.zeros(n, dtype=layout.numpy_record_dtype())
    for a in layout.attributes():
        dt = a.datatype()
        nc = dt.num_components()
        shape = (n, nc) if nc > 1 else (n,)
        npdt = dt.numpy_dtype()
        if npdt.kind == "f":
            vals = rng.random(shape).astype(npdt)
        else:
            info = np.iinfo(npdt)
            vals = rng.integers(info.min, info.max, size=shape, dtype=npdt, endpoint=True)
        rec[a.name()] = vals
    return rec


def make_buffer(kind: str, layout: PointLayout, records: np.ndarray):
    return BUFFER_KINDS[kind].from_numpy(records, layout)


def column(buffer, attribute):
    return buffer.view_attribute(attribute)
