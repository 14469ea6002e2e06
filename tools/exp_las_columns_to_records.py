"""Experiment: columns -> typed LAS records (HashMapBuffer -> VectorBuffer of LasPointFormatN), per point format: the format-specialised
transposition kernel (PST_LAS_PREFER_SPECIALISED=0) against the plan-specialised kernel (default; in-tree for formats 0-3, 6, 7, run-time
compiled otherwise -- the converter is prepared before the timed region).  Run once per setting; prints ms, fraction of 8 TB/s and the plan family."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd import las
from pasture_amd import conversion as cv
api = pa.product_api(); s = torch.cuda.current_stream(); api.set_stream(ctypes.c_void_p(s.cuda_stream))
n = 100_000_000
for f in [int(a) for a in sys.argv[1:]] or [0, 1, 3, 6, 7, 8]:
    lay = las.point_layout_from_las_point_format(las.Format(f), False)
    src = pa.HashMapBuffer.new_from_layout(lay); src.resize(n); src.synth_fill(42 + f, 0)
    dst = pa.VectorBuffer.new_from_layout(lay); dst.resize(n)
    conv = pa.BufferLayoutConverter.for_layouts(lay, lay)
    conv.prepare(pa.HashMapBuffer, pa.VectorBuffer)
    r = range(0, n)
    for _ in range(3): conv.convert_into_range_async(src, r, dst, r)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(10): conv.convert_into_range_async(src, r, dst, r)
    e1.record(s); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    size = lay.size_of_point_entry()
    print(f"format {f} ({size} B) columns -> records  {ms:.4f} ms  {2 * size * n / ms / 1e9 / 8:.4f} of peak  plan={cv.last_plan_kinds()}  "
          f"PST_LAS_PREFER_SPECIALISED={os.environ.get('PST_LAS_PREFER_SPECIALISED', '1')}", flush=True)
    del src, dst
