"""End-to-end example on the device: LAS records -> typed points -> transform -> bounds -> down-sample -> normals -> LAS records.

Mirrors what a pasture user writes with LASReader / transform_attribute / calculate_bounds / voxelgrid_filter /
compute_normals / LASWriter, with every per-point loop running on the MI355X.  Usage:

    python examples/las_pipeline.py [path/to/file.las]        (default: a fixture of the reference's test suite)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import pasture_amd as pa
from pasture_amd import las
from pasture_amd.layout import PointAttributeDataType as T, PointAttributeDefinition, PointLayout, attributes as A


def main(path):
    f = las.read_las_records(path)  # header + raw records (file parsing stays on the host)
    fmt = las.Format(f.point_format)
    typed = las.point_layout_from_las_point_format(fmt, False)
    print(f"{os.path.basename(path)}: format {f.point_format}, {f.num_points} points, record length {f.record_length}")

    # 1. read: raw records (host) -> typed columns (device), PCIe copy overlapped with the decode
    points = pa.HashMapBuffer.new_from_layout(typed)
    points.resize(f.num_points)
    host = torch.from_numpy(np.ascontiguousarray(f.records[:, :las.point_layout_from_las_point_format(fmt, True).size_of_point_entry()]).reshape(-1))
    las.read_records_into(host, f.point_format, f.scale, f.offset, points)

    # 2. process: shift the cloud, bounds, 2 x 2 x 2 voxel grid, normals of the down-sampled cloud
    pa.transform_attribute(points, A.POSITION_3D, pa.Transform.affine(T.Vec3f64, (1.0, 1.0, 1.0), (100.0, 200.0, 0.0)))
    bounds = pa.calculate_bounds(points)
    print("bounds after the shift:", bounds.min(), bounds.max())
    # (round 4: slices are views every algorithm takes -- the chunked min-max of pasture-tools' `info`; the centroid; a converting view)
    half = points.len() // 2
    lo, hi = pa.calculate_bounds(points.slice(range(0, max(1, half)))), pa.calculate_bounds(points.slice(range(half, points.len())))
    assert pa.AABB.union(lo, hi) == bounds
    print("centroid:", pa.compute_centroid(points), " intensity as f32:", points.view_attribute_with_conversion(A.INTENSITY.with_custom_datatype(T.F32))[:3])
    thinned = pa.HashMapBuffer.new_from_layout(typed)
    pa.voxelgrid_filter(points, 2.0, 2.0, 2.0, thinned)
    print(f"voxel grid 2.0: {points.len()} -> {thinned.len()} points")
    if thinned.len() >= 3:
        normals = pa.HashMapBuffer.new_from_layout(PointLayout.from_attributes([A.NORMAL, PointAttributeDefinition("Curvature", T.F64)]))
        normals.resize(thinned.len())
        pa.compute_normals_into(thinned, 3, normals)
        print("first normal:", normals.view_attribute(A.NORMAL)[0])

    # 3. write: typed points (device) -> raw records (host), header bounds and return counts as the LAS writer keeps them
    out = torch.empty(thinned.len() * las.point_layout_from_las_point_format(fmt, True).size_of_point_entry(), dtype=torch.uint8, pin_memory=True)
    header_bounds, by_return = las.write_records_from(thinned, f.point_format, f.scale, f.offset, out)
    print("written", thinned.len(), "records; header bounds", header_bounds, "points by return", by_return[:5])
    return thinned.len(), header_bounds


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "las", "10_points_format_3.las"))
