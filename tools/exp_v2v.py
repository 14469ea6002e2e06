"""Experiment: interleaved -> interleaved through the generic tile kernel (typed LAS-1 records, 43 B, into custom record layouts).
env PST_TILE_QUAD=0/1 selects the lane mapping."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd import las
from pasture_amd.layout import attributes as A, PointLayout
api = pa.product_api(); s = torch.cuda.current_stream(); api.set_stream(ctypes.c_void_p(s.cuda_stream))
n = 50_000_000
src_lay = las.point_layout_from_las_point_format(las.Format(1), False)
src = pa.VectorBuffer.new_from_layout(src_lay); src.resize(n); src.synth_fill(42, 0)
TARGETS = {
    24: [A.POSITION_3D],
    26: [A.POSITION_3D, A.INTENSITY],
    27: [A.POSITION_3D, A.INTENSITY, A.CLASSIFICATION],
    28: [A.POSITION_3D, A.INTENSITY, A.POINT_SOURCE_ID],
    32: [A.POSITION_3D, A.GPS_TIME],
    33: [A.GPS_TIME, A.CLASSIFICATION, A.POSITION_3D],
    35: [A.CLASSIFICATION, A.POSITION_3D, A.INTENSITY, A.GPS_TIME],
    43: list(reversed([m.attribute_definition() for m in src_lay.attributes()])),
}
for stride, attrs in TARGETS.items():
    lay = PointLayout.from_attributes_packed(attrs, 1)
    assert lay.size_of_point_entry() == stride, (stride, lay.size_of_point_entry())
    dst = pa.VectorBuffer.new_from_layout(lay); dst.resize(n)
    conv = pa.BufferLayoutConverter.for_layouts(src_lay, lay)
    r = range(0, n)
    for _ in range(3): conv.convert_into_range_async(src, r, dst, r)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(5): conv.convert_into_range_async(src, r, dst, r)
    e1.record(s); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("v2v 43 ->", stride, f"{ms:.3f} ms", f"{(43 + stride) * n / ms / 1e9:.2f} TB/s", flush=True)
    del dst
