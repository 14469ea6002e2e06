"""Experiment: LAS record encoder per format / source storage (kernel time from HIP events)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd import las
api = pa.product_api(); s = torch.cuda.current_stream(); api.set_stream(ctypes.c_void_p(s.cuda_stream))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
for fmt in (0, 1, 3, 6, 7, 10):
    typed = las.point_layout_from_las_point_format(las.Format(fmt), False)
    raw = las.point_layout_from_las_point_format(las.Format(fmt), True)
    for kind in ("H", "V"):
        src = (pa.HashMapBuffer if kind == "H" else pa.VectorBuffer).new_from_layout(typed); src.resize(n); src.synth_fill(42, 0)
        dst = pa.VectorBuffer.new_from_layout(raw); dst.resize(n)
        for _ in range(2): las.encode_points(src, fmt, (0.001,) * 3, (0.0,) * 3, dst)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5): las.encode_points(src, fmt, (0.001,) * 3, (0.0,) * 3, dst)
        e1.record(s); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        b = typed.size_of_point_entry() + raw.size_of_point_entry()
        print(f"format {fmt:2d} {kind}: {ms:7.3f} ms  {n / ms / 1e6:7.1f} Gpts/s  {n * b / ms / 1e9:6.2f} TB/s ({b} B/pt)", flush=True)
        del src, dst
