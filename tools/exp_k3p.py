"""Experiment: K3' (columnar -> interleaved LAS-0) alone, for rocprof counter collection."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd import las
api = pa.product_api(); s = torch.cuda.current_stream(); api.set_stream(ctypes.c_void_p(s.cuda_stream))
n = 100_000_000
mode = sys.argv[1] if len(sys.argv) > 1 else "h2v"
las0 = las.point_layout_from_las_point_format(las.Format(int(os.environ.get("FMT", "0"))), False)
src = (pa.HashMapBuffer if mode[0] == "h" else pa.VectorBuffer).new_from_layout(las0); src.resize(n); src.synth_fill(42, 0)
dst = (pa.HashMapBuffer if mode[2] == "h" else pa.VectorBuffer).new_from_layout(las0); dst.resize(n)
conv = pa.BufferLayoutConverter.for_layouts(las0, las0)
r = range(0, n)
for _ in range(4): conv.convert_into_range_async(src, r, dst, r)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(s)
for _ in range(5): conv.convert_into_range_async(src, r, dst, r)
e1.record(s); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(mode, "fmt", os.environ.get("FMT", "0"), ms, "ms", 2 * las0.size_of_point_entry() * n / ms / 1e9, "TB/s")
