"""Experiment: PCIe-inclusive LAS write: 10^8 typed LAS-0 points on the device -> raw records in pinned host memory."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd import las
api = pa.product_api(); s = torch.cuda.current_stream(); api.set_stream(ctypes.c_void_p(s.cuda_stream))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
typed = las.point_layout_from_las_point_format(las.Format(0), False)
src = pa.HashMapBuffer.new_from_layout(typed); src.resize(n); src.synth_fill(42, 0)
host = torch.empty(n * 20, dtype=torch.uint8, pin_memory=True)
for chunk in (1 << 20, 4 << 20, 16 << 20):
    las.write_records_from(src, 0, (0.001,) * 3, (0.0,) * 3, host, chunk_points=chunk)
    t0 = time.perf_counter(); las.write_records_from(src, 0, (0.001,) * 3, (0.0,) * 3, host, chunk_points=chunk); dt = time.perf_counter() - t0
    print(f"chunk {chunk:9d} points: {dt * 1e3:8.2f} ms  {n / dt / 1e9:5.2f} Gpts/s  link {n * 20 / dt / 1e9:5.1f} GB/s", flush=True)
