"""Experiment: cost of interpreted plan entries in the LDS tile kernel (LAS-0 35 B records -> columns)."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd import las
from pasture_amd.layout import attributes as A, PointLayout

api = pa.product_api()
s = torch.cuda.current_stream()
api.set_stream(ctypes.c_void_p(s.cuda_stream))
n = 100_000_000
src_l = las.point_layout_from_las_point_format(las.Format(0), False)
src = pa.VectorBuffer.new_from_layout(src_l); src.resize(n); src.synth_fill(42, 0)

def run(name, attrs, bytes_pp):
    dst_l = PointLayout.from_attributes_packed(attrs, 1)
    dst = pa.HashMapBuffer.new_from_layout(dst_l); dst.resize(n)
    conv = pa.BufferLayoutConverter.for_layouts_with_default(src_l, dst_l)
    for _ in range(3): conv.convert_into_range_async(src, range(0, n), dst, range(0, n))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(10): conv.convert_into_range_async(src, range(0, n), dst, range(0, n))
    e1.record(s); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name:28s} {ms:8.3f} ms  {bytes_pp * n / ms / 1e6:8.1f} GB/s (35 B read + {bytes_pp - 35} B written per point)", flush=True)

allattrs = [m.attribute_definition() for m in src_l.attributes()]
run("all 10 attributes", allattrs, 70)
run("position only", [A.POSITION_3D], 35 + 24)
run("9 small attributes", allattrs[1:], 35 + 11)
run("intensity only (u16)", [A.INTENSITY], 35 + 2)
run("classification only (u8)", [A.CLASSIFICATION], 35 + 1)
run("position + intensity", [A.POSITION_3D, A.INTENSITY], 35 + 26)
