"""Experiment: raw LAS-0 decode throughput per call as a function of the chunk size (the reference reads 1 MiB chunks)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd import las
api = pa.product_api(); s = torch.cuda.current_stream(); api.set_stream(ctypes.c_void_p(s.cuda_stream))
raw = las.point_layout_from_las_point_format(las.Format(0), True); typed = las.point_layout_from_las_point_format(las.Format(0), False)
N = 64_000_000
src = pa.VectorBuffer.new_from_layout(raw); src.resize(N); src.synth_fill(1, 0)
dst = pa.HashMapBuffer.new_from_layout(typed); dst.resize(N)
conv = las.get_default_las_converter(raw, typed, (0.001,) * 3, (0.0,) * 3)
for chunk in (52_428, 262_144, 1_048_576, 4_194_304, 16_777_216, 64_000_000):
    k = max(1, N // chunk)
    def run():
        for c in range(k):
            conv.convert_into_range_async(src, range(c * chunk, (c + 1) * chunk), dst, range(c * chunk, (c + 1) * chunk))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s); run(); e1.record(s); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    pts = k * chunk
    print(f"chunk {chunk:9d} points ({chunk * 20 / 2**20:7.1f} MiB of records): {pts / ms / 1e6:6.1f} Gpts/s  {pts * 55 / ms / 1e9:5.2f} TB/s  ({ms / k * 1e3:7.1f} us per call)", flush=True)
