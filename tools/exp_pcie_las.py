"""Experiment: PCIe-inclusive LAS read: 10^8 raw LAS-0 records in pinned host memory -> typed columns on the device."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd import las
api = pa.product_api(); s = torch.cuda.current_stream(); api.set_stream(ctypes.c_void_p(s.cuda_stream))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
typed = las.point_layout_from_las_point_format(las.Format(0), False)
host = torch.empty(n * 20, dtype=torch.uint8, pin_memory=True)
host.random_(0, 256)
dst = pa.HashMapBuffer.new_from_layout(typed); dst.resize(n)
for chunk in (1 << 20, 4 << 20, 16 << 20):
    las.read_records_into(host, 0, (0.001,) * 3, (0.0,) * 3, dst, chunk_points=chunk)
    t0 = time.perf_counter(); las.read_records_into(host, 0, (0.001,) * 3, (0.0,) * 3, dst, chunk_points=chunk); dt = time.perf_counter() - t0
    print(f"chunk {chunk:9d} points: {dt * 1e3:8.2f} ms  {n / dt / 1e9:5.2f} Gpts/s  link {n * 20 / dt / 1e9:5.1f} GB/s", flush=True)
