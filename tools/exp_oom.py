#!/usr/bin/env python3
"""Probe (round 6): what the boundary does when device memory runs out, and what the stream-ordered pool keeps from the rest of the process.
  1. a buffer larger than the device: status + message, and the library works afterwards;
  2. memory the library has freed (pool, release threshold = never) as seen by hipMemGetInfo, by a plain hipMalloc of another library (torch), and after
     pst_release_scratch();
  3. the device filled by someone else: resize / convert / filter / compute_normals / voxelgrid_filter report a status (no crash, no hang) and work again
     once the memory is back.
Prints one line per observation; exit code 0 when every call behaved (a status where one is due, correct results afterwards)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pasture_amd as pa  # noqa: E402
from pasture_amd import las  # noqa: E402
from pasture_amd._capi import PastureError  # noqa: E402
from pasture_amd.algorithms import calculate_bounds, compute_normals, voxelgrid_filter  # noqa: E402
from pasture_amd.buffers import HashMapBuffer, VectorBuffer  # noqa: E402
from pasture_amd.conversion import BufferLayoutConverter  # noqa: E402
from pasture_amd.layout import PointLayout, attributes as A  # noqa: E402

GIB = 1 << 30
bad = []


def free_gib():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / GIB


def expect_status(what, fn):
    try:
        fn()
    except PastureError as e:
        print(f"  {what}: status -> {str(e)[:160]}")
        return True
    except Exception as e:  # noqa: BLE001
        print(f"  {what}: OTHER EXCEPTION {type(e).__name__}: {str(e)[:160]}")
        bad.append(what)
        return False
    print(f"  {what}: NO ERROR REPORTED")
    bad.append(what)
    return False


def small_round_trip(tag):
    layout = PointLayout.from_attributes([A.POSITION_3D])
    src = HashMapBuffer.new_from_layout(layout)
    src.resize(100_000)
    src.synth_fill(42, 0)
    out = BufferLayoutConverter.for_layouts(layout, layout).convert(src, VectorBuffer)
    ok = out.get_attribute_range(A.POSITION_3D, range(0, 100_000)).tobytes() == src.get_attribute_range(A.POSITION_3D, range(0, 100_000)).tobytes()
    b = calculate_bounds(out)
    print(f"  [{tag}] small conversion + bounds afterwards: {'ok' if ok and b is not None else 'WRONG'}")
    if not ok:
        bad.append("after " + tag)


def main():
    api = pa.product_api()
    layout = PointLayout.from_attributes([A.POSITION_3D])
    print(f"free at start: {free_gib():.1f} GiB")
    print("1. a buffer larger than the device")
    b = HashMapBuffer.new_from_layout(layout)
    expect_status("resize(2e11 points = 4.8 TB)", lambda: b.resize(200_000_000_000))
    print(f"  len after the failed resize: {b.len()}")
    small_round_trip("oversized resize")

    print("2. what the pool keeps")
    f0 = free_gib()
    big = HashMapBuffer.new_from_layout(layout)
    big.resize(2_000_000_000)  # 48 GB
    f1 = free_gib()
    del big
    f2 = free_gib()
    print(f"  free before / with a 44.7-GiB buffer / after its destruction: {f0:.1f} / {f1:.1f} / {f2:.1f} GiB")
    want = int((f0 - 4) * GIB)
    try:
        t = torch.empty(want, dtype=torch.uint8, device="cuda")
        print(f"  torch.empty({want / GIB:.0f} GiB) while the pool holds the freed block: ok")
        del t
        torch.cuda.empty_cache()
    except Exception as e:  # noqa: BLE001
        print(f"  torch.empty({want / GIB:.0f} GiB) while the pool holds the freed block: FAILS ({type(e).__name__})")
    api.release_scratch()
    f3 = free_gib()
    print(f"  free after pst_release_scratch(): {f3:.1f} GiB")
    try:
        t = torch.empty(want, dtype=torch.uint8, device="cuda")
        print(f"  torch.empty({want / GIB:.0f} GiB) after pst_release_scratch(): ok")
        del t
        torch.cuda.empty_cache()
    except Exception as e:  # noqa: BLE001
        print(f"  torch.empty({want / GIB:.0f} GiB) after pst_release_scratch(): FAILS ({type(e).__name__})")

    print("3. the device filled by another allocator")
    n = 20_000_000
    cloud = HashMapBuffer.new_from_layout(layout)
    cloud.resize(n)
    cloud.synth_fill(42, 0)
    las_l = las.point_layout_from_las_point_format(las.Format(0), False)
    pts = HashMapBuffer.new_from_layout(las_l)
    pts.resize(n)
    pts.synth_fill(7, 0)
    mask = (torch.arange(n, device="cuda") % 2 == 0).to(torch.uint8)
    api.release_scratch()
    torch.cuda.empty_cache()
    leave = 0.25  # GiB
    hog = torch.empty(int((free_gib() - leave) * GIB), dtype=torch.uint8, device="cuda")
    print(f"  free with the hog: {free_gib():.2f} GiB")
    grow = HashMapBuffer.new_from_layout(layout)
    expect_status("resize(1e8 points) on the full device", lambda: grow.resize(100_000_000))
    expect_status("convert (allocates 480 MB)", lambda: BufferLayoutConverter.for_layouts(layout, layout).convert(cloud, VectorBuffer))
    expect_status("filter (allocates the target)", lambda: pts.filter(HashMapBuffer, (mask.data_ptr(), "device")))
    expect_status("compute_normals (scratch ~ 1.1 GB)", lambda: compute_normals(cloud, 16))
    expect_status("voxelgrid_filter (keys + sort scratch)", lambda: voxelgrid_filter(cloud, 2.5, 2.5, 2.5, HashMapBuffer.new_from_layout(layout)))
    del hog
    torch.cuda.empty_cache()
    print(f"  free without the hog: {free_gib():.1f} GiB")
    small_round_trip("full device")
    normals = compute_normals(cloud, 16)
    print(f"  compute_normals afterwards: {len(normals[0])} normals, finite: {bool(np.isfinite(normals[0]).all())}")
    out = HashMapBuffer.new_from_layout(layout)
    voxelgrid_filter(cloud, 2.5, 2.5, 2.5, out)
    kept = pts.filter(HashMapBuffer, (mask.data_ptr(), "device"))
    print(f"  voxelgrid_filter afterwards: {out.len()} voxels; filter afterwards: {kept.len()} of {n}")
    if kept.len() != n // 2 or out.len() == 0:
        bad.append("results after the hog")
    print("BAD: " + ", ".join(bad) if bad else "all calls behaved")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
