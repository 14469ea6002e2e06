"""Experiment: cost of the fused AABB in the tile kernel (LAS-0 interleaved -> columns)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd import las
from pasture_amd.layout import attributes as A, PointLayout
api = pa.product_api(); s = torch.cuda.current_stream(); api.set_stream(ctypes.c_void_p(s.cuda_stream))
n = 100_000_000
src_l = las.point_layout_from_las_point_format(las.Format(0), False)
src = pa.VectorBuffer.new_from_layout(src_l); src.resize(n); src.synth_fill(42, 0)
rec = torch.empty(6, dtype=torch.float64, device="cuda")
def run(name, attrs, bounds):
    dst_l = PointLayout.from_attributes_packed(attrs, 1)
    dst = pa.HashMapBuffer.new_from_layout(dst_l); dst.resize(n)
    conv = pa.BufferLayoutConverter.for_layouts_with_default(src_l, dst_l)
    r = range(0, n)
    f = (lambda: conv.convert_into_with_bounds_async(src, dst, rec.data_ptr())) if bounds else (lambda: conv.convert_into_range_async(src, r, dst, r))
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(10): f()
    e1.record(s); torch.cuda.synchronize()
    print(f"{name:40s} bounds={bounds}: {e0.elapsed_time(e1)/10:8.3f} ms", flush=True)
allattrs = [m.attribute_definition() for m in src_l.attributes()]
for b in (False, True):
    run("position only", [A.POSITION_3D], b)
    run("position + intensity", [A.POSITION_3D, A.INTENSITY], b)
    run("all 10", allattrs, b)
    run("all 10, position last", allattrs[1:] + allattrs[:1], b)
